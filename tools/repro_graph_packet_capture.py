"""Debug aid (round 2): many multi-block torch reductions inside ONE captured graph (allocator reuse of the semaphore
blocks), replayed with changing inputs."""
import sys
import torch
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "relaxed"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
EVT = len(sys.argv) > 3
if EVT:
    import os, ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from easevoice_trainer_amd.hip import lib as L


def body(x, outs):
    # x [4, 172, 192]; the engine reduces transposed views
    xt = x.transpose(1, 2)
    for i in range(N):
        km = (xt * (1.0 + i)) * 1.0
        outs.append(torch.sum(km))
        tmp = torch.empty(1000 + 37 * i, device=dev).fill_(float(i))      # churn the small pool
        outs.append(km.sum(-1).sum())
        outs.append(tmp.sum())
        w = torch.randn(576, 192, device=dev)
        y = torch.nn.functional.linear(x.reshape(-1, 192), w)
        outs.append(y.sum(0).sum())
        if EVT:
            o = torch.empty_like(x)
            L.check(L.lib().evt_leaky_relu(L.dt_of(x), L.ptr(x), C.c_float(0.1), L.ptr(o), C.c_int64(x.numel()), L.stream_ptr()), 'lrelu')
            tab = L.struct_to_device([L.Seg(o.data_ptr(), x.data_ptr(), None, x.numel(), 1.0, 0)], dev)
            lo = torch.zeros(1, device=dev)
            L.check(L.lib().evt_l1_multi_fwd(L.dt_of(x), L.ptr(tab), 1, L.ptr(lo), L.stream_ptr()), 'l1')
            outs.append(lo)
    return outs


x = torch.randn(4, 172, 192, device=dev)
for _ in range(2):
    body(x, [])
torch.cuda.synchronize()
pool = torch.cuda.graph_pool_handle()
g = torch.cuda.CUDAGraph()
outs = []
with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
    body(x, outs)
K = 5 if EVT else 4
for r in range(4):
    x.copy_(torch.randn(4, 172, 192, device=dev) * (r + 1))
    g.replay()
    torch.cuda.synchronize()
    xt = x.transpose(1, 2)
    bad = []
    for i in range(N):
        ref = float(((xt * (1.0 + i)) * 1.0).sum())
        a, b = float(outs[K * i]), float(outs[K * i + 1])
        if abs(a - ref) > 1e-3 * abs(ref) + 0.5 or abs(b - ref) > 1e-3 * abs(ref) + 0.5:
            bad.append((i, round(a, 2), round(b, 2), round(ref, 2)))
        t = float(outs[K * i + 2]); tr = float(i) * (1000 + 37 * i)
        if abs(t - tr) > 1e-3 * abs(tr) + 0.5:
            bad.append(("tmp", i, t, tr))
    print(f"mode={mode} replay {r}: {len(bad)} wrong: {bad[:6]}", flush=True)
