"""Debug aid (round 2): run the s2 trainer over the test feature directory with HIP-graph replay and report, per step,
how the step ran (eager / capture+replay / replay) and which loss terms are non-finite."""
import io, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import data_fixture as F
from easevoice_trainer_amd.train import s2_engine as E
from easevoice_trainer_amd.train.sovits import SovitsTrain, SovitsTrainParams

gold = json.load(open(os.path.join(ROOT, "tests", "golden", "data_readers.json")))
root = tempfile.mkdtemp()
F.build_feature_dir(root, gold["symbols"])
os.remove(os.path.join(root, "5-wav32k", "a_007.wav"))
json.dump(gold["symbols"], open(os.path.join(root, "symbols.json"), "w"))
tmp = tempfile.mkdtemp()
g = torch.Generator().manual_seed(3)
pre = "quantizer.vq.layers.0._codebook."
torch.save({"weight": {pre + "inited": torch.ones(1), pre + "embed": torch.randn(1024, 768, generator=g),
                       pre + "embed_avg": torch.randn(1024, 768, generator=g), pre + "cluster_size": torch.ones(1024)}},
           os.path.join(tmp, "s2G.pth"))

orig = E.S2Engine.step
n = [0]


def step(self, *a, **k):
    inputs = a
    key = tuple((tuple(t.shape), t.dtype) if t is not None else None for t in (list(a) + [None, None])[:8])
    ent = getattr(self, "_graph_cache", {}).get(key)
    had = ent is not None and ent["graphs"] is not None
    out = orig(self, *a, **k)
    if getattr(self, "_in_dbg", False):
        return out
    ent = getattr(self, "_graph_cache", {}).get(key)
    has = ent is not None and ent["graphs"] is not None
    torch.cuda.synchronize()
    vals = {f: float(getattr(out, f)) for f in ("disc", "gen", "fm", "mel", "kl")}
    pn = sum(int(not torch.isfinite(p).all()) for p in self.net_g.parameters())
    pd = sum(int(not torch.isfinite(p).all()) for p in self.net_d.parameters())
    gn = int(not torch.isfinite(self.rt_g.arena.grad).all()), int(not torch.isfinite(self.rt_d.arena.grad).all())
    mode = "replay" if had else ("capture+replay" if has else "eager")
    if pn and not getattr(self, "_reported", False):
        self._reported = True
        bad = [nm for nm, q in self.net_g.named_parameters() if not torch.isfinite(q).all()]
        badg = [nm for nm, q in self.net_g.named_parameters() if q.grad is not None and not torch.isfinite(q.grad).all()]
        print("   NaN param prefixes:", sorted({".".join(x.split(".")[:3]) for x in bad}), file=sys.stderr)
        print("   NaN grad  prefixes:", sorted({".".join(x.split(".")[:3]) for x in badg}), file=sys.stderr)
        ok = [nm for nm, q in self.net_g.named_parameters() if torch.isfinite(q).all()]
        print("   finite param prefixes:", sorted({".".join(x.split(".")[:2]) for x in ok}), file=sys.stderr)
        if ent is not None and ent.get("st") is not None:
            st = ent["st"]
            for nm in ("y_hat", "y_hat_mel", "y_mel", "y_seg", "loss_disc", "loss_mel", "loss_kl", "loss_fm", "loss_gen"):
                v = getattr(st, nm, None)
                if torch.is_tensor(v):
                    print(f"   st.{nm} finite={bool(torch.isfinite(v).all())}", file=sys.stderr)
            for i, t in enumerate(st.lat):
                print(f"   st.lat[{i}] finite={bool(torch.isfinite(t).all())}", file=sys.stderr)
    print(f"step {n[0]:2d} {mode:15s} T={a[1].shape[2]} lens={a[2].tolist()} tl={a[5].tolist()} "
          + " ".join(f"{k}={v:.4g}" for k, v in vals.items()) + f" nan_params G={pn} D={pd} nan_grad={gn}", file=sys.stderr, flush=True)
    n[0] += 1
    return out


def step_outer(self, *a, **k):
    if getattr(self, "_in_dbg", False):
        return orig(self, *a, **k)
    self._in_dbg = False
    return step(self, *a, **k)


_sg = E.S2Engine._step_graphed


def sg(self, inputs):
    self._in_dbg = True
    try:
        return _sg(self, inputs)
    finally:
        self._in_dbg = False


E.S2Engine._step_graphed = sg
E.S2Engine.step = step_outer
if "--nopin" in sys.argv:
    from easevoice_trainer_amd.train import dataset as DS
    _c = DS.collate_s2
    DS.collate_s2 = lambda items, bins, pin=False, **k: _c(items, bins, pin=False, **k)
p = SovitsTrainParams(batch_size=4, total_epochs=1, save_every_epoch=1, output_model_name="fd", project_dir=tmp,
                      train_input_dir=root, pretrained_s2G=os.path.join(tmp, "s2G.pth"))
buf = io.StringIO()
from contextlib import redirect_stdout
with redirect_stdout(buf):
    SovitsTrain(p).train()
print("\n".join(l for l in buf.getvalue().splitlines() if "easevoice" in l), file=sys.stderr)
