import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
from debug_graph_mix import batch, dev
from debug_graph_rng import engine

eng = engine(False)
T, Tt, B = 172, 30, 4
for i in range(6):
    a = batch(T, Tt, (T,) * 4, (Tt,) * 4, 100 + i)
    out = eng.step(*a)
    torch.cuda.synchronize()
    e = out.extras
    print(f"step {i} ids={e['ids_slice'].tolist()} disc={float(out.disc):.3f} gen={float(out.gen):.3f} fm={float(out.fm):.3f} "
          f"mel={float(out.mel):.2f} kl={float(out.kl):.2f} |y_hat|max={float(e['y_hat'].abs().max()):.3f} "
          f"z std={float(e['z'].std()):.3f} m_q std={float(e['m_q'].std()):.3f} logs_q mean={float(e['logs_q'].mean()):.3f} "
          f"y_mel mean={float(e['y_mel'].mean()):.3f} y_hat_mel mean={float(e['y_hat_mel'].mean()):.3f} "
          f"gss_g={float(out.grad_sumsq_g):.4g} gss_d={float(out.grad_sumsq_d):.4g}", flush=True)
