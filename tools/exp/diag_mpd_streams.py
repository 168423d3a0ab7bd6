"""which outputs of the generator step's discriminator node differ between one stream and branch streams (and run to run)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from easevoice_trainer_amd.hip import disc as HD
from easevoice_trainer_amd.module.models import MultiPeriodDiscriminator
from easevoice_trainer_amd.runtime import ModelRuntime

gpu = torch.device("cuda", 0)
torch.manual_seed(5)
net_d = MultiPeriodDiscriminator(False)
rt = ModelRuntime(net_d, torch.bfloat16, gpu)
rt.prepare()
rt.bank.weight_grads = False
n, T = 4, 20480
y = (torch.rand(n, 1, T, device=gpu) - 0.5)
y_hat = torch.tanh(torch.randn(n, 1, T, device=gpu) * 0.5)


def run(ns):
    HD.MPD_STREAMS = ns
    b = y_hat.clone().requires_grad_(True)
    gen, fm, logits = net_d.generator_losses(y, b)
    (gen * 0.7 + fm * 1.3).backward()
    torch.cuda.synchronize()
    return [gen.detach().clone(), fm.detach().clone(), b.grad.clone()] + [l.clone() for l in logits]


names = ["gen", "fm", "dwav"] + [f"logit{i}" for i in range(6)]
ref = run(1)
for ns in (1, 1, 2, 2, 2, 3, 3):
    got = run(ns)
    bad = [(nm, float((u.float() - v.float()).abs().max()), float(v.float().abs().max())) for nm, u, v in zip(names, got, ref) if not torch.equal(u, v)]
    print("streams", ns, "differs:", bad)
