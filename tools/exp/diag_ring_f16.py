"""one-off diagnostic (GPU): conv_ring backward-data behind an output activation, bf16 vs f16, where the error sits"""
import sys, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")]
import torch
import test_conv_gpu as TC
from easevoice_trainer_amd.hip import conv as HC, lib as L
from oracle import ops as O

gpu = torch.device("cuda:0")
case = TC.RING_CASES[0]
for dtype in (torch.bfloat16, torch.float16):
    L.set_half(dtype)
    for fi in (0, 2, 3):
        fusion = TC.FUSIONS[fi]
        rep = []
        TC._run_case(gpu, case, fusion, dtype, 0, report=rep)
        print(dtype, "fusion", fi, [(n, f"{e:.2e}") for n, e in rep])
# where: redo f16 fusion 2 by hand
dtype = torch.float16
L.set_half(dtype)
cin, cout, k, stride, pad, dil, groups, transposed, wn, Lin, nseq = case
torch.manual_seed(hash(case) % 100000)
m = HC.EvtConv1d(cin, cout, k, stride, pad, dil, groups, bias=True, transposed=False, weight_norm=wn)
x = torch.randn(nseq, cin, Lin).to(dtype).float()
dy = torch.randn(nseq, cout, Lin).to(dtype).float()
xo = x.clone().requires_grad_(True)
po = {n_: p.detach().clone().requires_grad_(True) for n_, p in m.named_parameters()}
w = O.weight_norm_fold(po["weight_v"], po["weight_g"])
w = w + (w.detach().to(dtype).float() - w.detach())
yo = O.conv_block(xo, w, po.get("bias"), None, stride=stride, pad=pad, dil=dil, groups=groups, transposed=False, in_slope=1.0, out_act=1, out_slope=0.1)
yo.backward(dy)
m = m.to(gpu)
bank = HC.WeightBank(m, dtype, gpu)
bank.build_tables(); bank.fold()
xg = x.transpose(1, 2).contiguous().to(gpu, dtype).requires_grad_(True)
yg = m(xg, None, 1.0, 1, 0.1)
yg.backward(dy.transpose(1, 2).contiguous().to(gpu, dtype))
bank.grads(); torch.cuda.synchronize()
ey = (yg.detach().float().cpu().transpose(1, 2) - yo.detach()).abs()
print("y err max", float(ey.max()), "y absmax", float(yo.abs().max()), "y_gpu absmax", float(yg.abs().max()), "nonfinite", int((~torch.isfinite(yg)).sum()))
sg = (yg.detach().float().cpu().transpose(1, 2) > 0) != (yo.detach() > 0)
print("sign disagreements between y_gpu and y_ref:", int(sg.sum()), "of", sg.numel())
e = (xg.grad.float().cpu().transpose(1, 2) - xo.grad).abs()
mx = float(xo.grad.abs().max())
bad = e > 1e-2 * mx
print("dx: bad elements", int(bad.sum()), "of", bad.numel(), "max", float(e.max()) / mx)
idx = bad.nonzero()
print("bad (seq, channel, pos) head:", idx[:20].tolist())
print("bad per seq:", torch.bincount(idx[:, 0], minlength=nseq).tolist())
print("bad positions hist (pos // 16):", torch.bincount(idx[:, 2] // 16, minlength=(Lin + 15) // 16).tolist())
print("bad channels hist (c // 32):", torch.bincount(idx[:, 1] // 32, minlength=cin // 32).tolist())
