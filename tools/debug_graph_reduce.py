"""Debug aid (round 2): do torch's multi-block reductions (semaphore + memset) survive HIP-graph replay?"""
import torch
dev = torch.device("cuda:0")


def trial(name, shape, fn, dtype=torch.float32, n=4):
    x = torch.randn(*shape, device=dev, dtype=dtype)
    fn(x); fn(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fn(x)
    errs = []
    for i in range(n):
        x.copy_(torch.randn(*shape, device=dev, dtype=dtype) + i)
        g.replay()
        torch.cuda.synchronize()
        ref = fn(x)
        errs.append(float((y.float() - ref.float()).abs().max() / (ref.float().abs().max() + 1e-9)))
    print(f"{name:40s} rel err per replay: {['%.2e' % e for e in errs]}", flush=True)


trial("sum all 4x172x192 f32", (4, 172, 192), lambda x: x.sum())
trial("sum all 16x200x192 f32", (16, 200, 192), lambda x: x.sum())
trial("sum all 4x100x192 f32", (4, 100, 192), lambda x: x.sum())
trial("sum dim0 688x576 bf16", (688, 576), lambda x: x.sum(0), torch.bfloat16)
trial("sum dim0 3200x576 bf16", (3200, 576), lambda x: x.sum(0), torch.bfloat16)
trial("sum dim0 400x576 bf16", (400, 576), lambda x: x.sum(0), torch.bfloat16)
trial("mean 2M f32", (2 * 1024 * 1024,), lambda x: x.mean())
trial("max 2M f32", (2 * 1024 * 1024,), lambda x: x.max())
trial("l1 (a-b).abs().mean 4x128x32", (4, 128, 32), lambda x: (x - 1).abs().mean())
trial("zero_+add (memset?)", (1024 * 1024,), lambda x: torch.zeros_like(x).add_(x))
