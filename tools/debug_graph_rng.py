"""Debug aid (round 2): does the replay NaN need ragged lengths, RNG draws inside the graph, or both?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
from debug_graph_mix import batch, dev, hps
from easevoice_trainer_amd.train.s2_engine import S2Engine


def engine(dropout):
    h = json.loads(json.dumps(hps))
    if not dropout:
        h["model"]["p_dropout"] = 0.0
    torch.manual_seed(0)
    eng = S2Engine(h, dev, torch.bfloat16)
    if not dropout:
        for m in eng.net_g.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    eng.build_optimizers()
    cb = eng.net_g.quantizer.vq.layers[0]._codebook
    cb.embed.normal_(); cb.inited.fill_(1.0)
    eng.enable_graphs(warmup_steps=2)
    return eng


def exp(tag, ragged, dropout, draws):
    print(tag, flush=True)
    eng = engine(dropout)
    T, Tt, B = 172, 30, 4
    for i in range(7):
        lens = [(170, 100, 100, 47), (170, 102, 102, 40), (170, 65, 47, 40), (150, 170, 99, 64)][i % 4] if ragged else (T,) * 4
        tl = [(30, 1, 1, 8), (30, 18, 18, 7), (30, 11, 8, 7), (29, 30, 5, 9)][i % 4] if ragged else (Tt,) * 4
        a = batch(T, Tt, lens, tl, 100 + i)
        kw = {}
        if not draws:
            g = torch.Generator().manual_seed(5 + i)
            kw = dict(eps=torch.randn(B, 192, T, generator=g).to(dev),
                      ids_slice=torch.tensor([min(l - 32, 3 + 7 * j) for j, l in enumerate(lens)], device=dev))
        out = eng.step(*a, **kw)
        torch.cuda.synchronize()
        badg = sorted({".".join(n.split(".")[:4]) for n, p in eng.net_g.named_parameters()
                       if p.grad is not None and not torch.isfinite(p.grad).all()})
        print(f"   step {i} gen_all={float(out.gen_all):9.3f} NaN grads: {badg[:12]} ({len(badg)})", flush=True)
        if badg:
            break
    del eng
    torch.cuda.empty_cache()


if __name__ == "__main__":
  exp("E4 ragged, no dropout, injected eps/ids", True, False, False)
  exp("E5 ragged, no dropout, drawn eps/ids", True, False, True)
  exp("E6 ragged, dropout, injected eps/ids", True, True, False)
  exp("E7 full,   dropout, drawn eps/ids", False, True, True)
  exp("E8 full,   no dropout, injected", False, False, False)
