import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch
from debug_graph_mix import batch, dev
from debug_graph_rng import engine
import easevoice_trainer_amd.train.s2_engine as SE

keep = {}


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    z_p, logs_q, m_p, logs_p, z_mask = z_p.float(), logs_q.float(), m_p.float(), logs_p.float(), z_mask.float()
    kl = logs_p - logs_q - 0.5
    kl = kl + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2.0 * logs_p)
    km = kl * z_mask
    keep["km"] = km.detach().clone()
    keep["sum_km"] = torch.sum(km).detach().clone()
    keep["sum_mask"] = torch.sum(z_mask).detach().clone()
    keep["sum_km_2stage"] = km.sum(-1).sum().detach().clone()
    keep["sum_km_contig"] = km.contiguous().view(-1).sum().detach().clone()
    keep["strides"] = (tuple(km.shape), km.stride(), tuple(z_mask.shape), z_mask.stride())
    return torch.sum(km) / torch.sum(z_mask)


SE.kl_loss = kl_loss
eng = engine(False)
T, Tt, B = 172, 30, 4
for i in range(5):
    a = batch(T, Tt, (T,) * 4, (Tt,) * 4, 100 + i)
    out = eng.step(*a)
    torch.cuda.synchronize()
    print(f"step {i} kl={float(out.kl):.4f} sum_km={float(keep['sum_km']):.2f} eager_sum_of_km={float(keep['km'].sum()):.2f} "
          f"2stage={float(keep['sum_km_2stage']):.2f} contig={float(keep['sum_km_contig']):.2f} sum_mask={float(keep['sum_mask']):.1f} "
          f"km finite={bool(torch.isfinite(keep['km']).all())} {keep['strides']}", flush=True)
