"""Debug aid (round 2): capture the s2 step with gradient taps (clones recorded inside the graph) and report which
intermediate gradients are non-finite after the second replay."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
from debug_graph_mix import batch, fresh, A  # noqa  (runs nothing at import: guarded below)

keep = {}


def tap(name, t):
    if torch.is_tensor(t) and t.requires_grad:
        def h(g, name=name):
            keep[name] = g.detach().float().clone()
        t.register_hook(h)


eng = fresh()
G = eng.net_g
dec = G.dec
orig_dec = dec.forward


def dec_fwd(x, g=None):
    tap("d.z_slice", x); tap("d.ge_into_dec", g)
    y = orig_dec(x, g=g)
    tap("d.y_hat(dec out)", y)
    return y


dec.forward = dec_fwd
for i, u in enumerate(dec.ups):
    u.register_forward_hook(lambda m, a, o, i=i: tap(f"d.ups{i}.out", o))
for i, r in enumerate(dec.resblocks):
    r.register_forward_hook(lambda m, a, o, i=i: tap(f"d.res{i}.out", o))
dec.conv_pre.register_forward_hook(lambda m, a, o: tap("d.conv_pre.out", o))
G.enc_q.register_forward_hook(lambda m, a, o: ([tap(f"d.enc_q.out{j}", t) for j, t in enumerate(o)], None)[1])
G.flow.register_forward_hook(lambda m, a, o: tap("d.flow.out", o))
G.ref_enc.register_forward_hook(lambda m, a, o: tap("d.ref_enc.out", o))
import easevoice_trainer_amd.module.mel_processing as MP
import easevoice_trainer_amd.train.s2_engine as SE
_mel = SE.mel_spectrogram_torch


def mel(y, *a, **k):
    tap("d.y_hat(into mel)", y)
    o = _mel(y, *a, **k)
    tap("d.y_hat_mel", o)
    return o


SE.mel_spectrogram_torch = mel
_fs = eng.net_d.forward_single


def fs(y):
    tap("d.y_hat(into D)", y)
    outs, fm = _fs(y)
    for i, o in enumerate(outs):
        tap(f"d.D{i}.logit", o)
    return outs, fm


eng.net_d.forward_single = fs

for i in range(3):
    out = eng.step(*A(i))
torch.cuda.synchronize()
print("after capture+replay#1:", {k: bool(torch.isfinite(v).all()) for k, v in keep.items() if not torch.isfinite(v).all()} or "all finite", flush=True)
out = eng.step(*A(10, (170, 102, 102, 40), (30, 18, 18, 7)))
torch.cuda.synchronize()
for k, v in keep.items():
    fin = torch.isfinite(v)
    print(f"  {k:24s} shape={tuple(v.shape)} finite={bool(fin.all())} bad={int((~fin).sum())}", end="")
    if not fin.all():
        idx = (~fin).nonzero()
        print(" first_bad=", idx[0].tolist(), "last_bad=", idx[-1].tolist(), end="")
    print(flush=True)
badg = sorted({".".join(n.split(".")[:3]) for n, p in eng.net_g.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()})
badp = sorted({".".join(n.split(".")[:3]) for n, p in eng.net_g.named_parameters() if not torch.isfinite(p).all()})
print("NaN grads:", badg)
print("NaN params:", badp)
print("dw_arena finite:", bool(torch.isfinite(eng.rt_g.bank.dw_arena).all()), "reg:", bool(torch.isfinite(eng.rt_g.bank.reg_arena.float()).all()),
      "alt:", bool(torch.isfinite(eng.rt_g.bank.alt_arena.float()).all()))
for s in eng.rt_g.bank.slots[:0]:
    pass
names = {id(m): n for n, m in eng.net_g.named_modules()}
bad_dw = [names[id(s.module)] for s in eng.rt_g.bank.slots if not torch.isfinite(s.dw).all()]
print("slots with NaN dW image:", bad_dw[:80])
bad_bias = [n for n, p in eng.net_g.named_parameters() if n.endswith("bias") and p.grad is not None and not torch.isfinite(p.grad).all()]
print("NaN bias grads:", bad_bias[:80])
