"""one-off diagnostic (GPU): conv_ring case 0 with an output activation in f16 -- replicate tests/test_conv_gpu.py::_run_case
step by step and say where the deviation sits"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests")]
import torch
import test_conv_gpu as TC
from easevoice_trainer_amd.hip import conv as HC, lib as L
from oracle import ops as O

gpu = torch.device("cuda:0")
case = TC.RING_CASES[0]
dtype = torch.float16
L.set_half(dtype)
for impl in (0, 1):
    rep = []
    TC._run_case(gpu, case, TC.FUSIONS[2], dtype, impl, report=rep)
    print("impl", impl, [(n, f"{e:.2e}") for n, e in rep])
fusion = TC.FUSIONS[2]
cin, cout, k, stride, pad, dil, groups, transposed, wn, Lin, nseq = case
torch.manual_seed(hash(case) % 100000)
m = HC.EvtConv1d(cin, cout, k, stride, pad, dil, groups, bias=True, transposed=transposed, weight_norm=wn)
with torch.no_grad():
    m.weight_g.mul_(torch.rand_like(m.weight_g) + 0.5)
x = torch.randn(nseq, cin, Lin)
lout = m.lout(Lin)
dy = torch.randn(nseq, cout, lout)
x, dy = x.to(dtype).float(), dy.to(dtype).float()
xo = x.clone().requires_grad_(True)
po = {n_: p.detach().clone().requires_grad_(True) for n_, p in m.named_parameters()}
w = O.weight_norm_fold(po["weight_v"], po["weight_g"])
w = w + (w.detach().to(dtype).float() - w.detach())
yo = O.conv_block(xo, w, po.get("bias"), None, stride=stride, pad=pad, dil=dil, groups=groups, transposed=transposed,
                  in_slope=1.0, out_act=1, out_slope=0.1)
yo.backward(dy)
m = m.to(gpu)
bank = HC.WeightBank(m, dtype, gpu, impl=0)
bank.build_tables(); bank.fold()
xg = x.transpose(1, 2).contiguous().to(gpu, dtype).requires_grad_(True)
yg = m(xg, None, 1.0, 1, 0.1)
yg.backward(dy.transpose(1, 2).contiguous().to(gpu, dtype))
bank.grads(); torch.cuda.synchronize()
yg_c = yg.detach().float().cpu().transpose(1, 2)
print("y: ref absmax", float(yo.abs().max()), "gpu absmax", float(yg_c.abs().max()), "max err", float((yg_c - yo.detach()).abs().max()))
sg = (yg_c > 0) != (yo.detach() > 0)
print("sign disagreements:", int(sg.sum()), "of", sg.numel(), "| y == 0 on gpu:", int((yg_c == 0).sum()), "| ref |y| < 1e-4:", int((yo.detach().abs() < 1e-4).sum()))
neg = yo.detach() < 0
print("negative-branch values: ref min", float(yo.detach()[neg].abs().min()), "gpu min abs over the same", float(yg_c[neg].abs().min()))
e = (xg.grad.float().cpu().transpose(1, 2) - xo.grad).abs()
mx = float(xo.grad.abs().max())
bad = e > 1e-2 * mx
print("dx: bad elements", int(bad.sum()), "of", bad.numel(), "max rel", float(e.max()) / mx)
idx = bad.nonzero()
print("bad head (seq, channel, pos):", idx[:12].tolist())
print("bad per seq:", torch.bincount(idx[:, 0], minlength=nseq).tolist())
print("bad per position block of 16:", torch.bincount(idx[:, 2] // 16, minlength=(Lin + 15) // 16).tolist())
db = (m.bias.grad.float().cpu() - po["bias"].grad).abs()
print("dbias: worst channels", torch.topk(db, 6).indices.tolist(), "err", [round(float(v), 4) for v in torch.topk(db, 6).values], "ref there",
      [round(float(po["bias"].grad[i]), 4) for i in torch.topk(db, 6).indices])
# dy_eff as the library computes it, against dy * lrelu'(y_ref)
import ctypes as C
dyg = dy.transpose(1, 2).contiguous().to(gpu, dtype)
dye = torch.empty_like(dyg)
L.check(L.lib().evt_dact_mul(L.dt_of(dyg), L.ptr(dyg), L.ptr(yg.detach().contiguous()), 1, C.c_float(0.1), L.ptr(dye), C.c_int64(dyg.numel()), L.stream_ptr()), "dact")
torch.cuda.synchronize()
ref_eff = dy * torch.where(yo.detach() > 0, torch.ones(()), torch.full((), 0.1))
ee = (dye.float().cpu().transpose(1, 2) - ref_eff).abs()
print("dact_mul vs reference: max abs err", float(ee.max()), "elements off by > 1e-2:", int((ee > 1e-2).sum()))
print("wants_plain_dy:", L.lib().evt_conv1d_wants_plain_dy(C.byref(m._slot.params(nseq, Lin, 1.0, 1, 0.1))))
