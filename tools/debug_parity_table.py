"""Debug aid (round 2): relative errors of every checked quantity of the round-2 parity tests, to set tolerances from
measurements rather than guesses."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch, yaml
import test_s2_parity_r2_gpu as P
from util_fill import fill_module, s1_batch

gpu = torch.device("cuda:0")
G = os.path.join(ROOT, "tests", "golden")


def table(gold, out, gd, gg, tag):
    got = dict(disc=out.disc, gen=out.gen, fm=out.fm, mel=out.mel, kl=out.kl, gen_all=out.gen_all)
    print(tag, "losses:", {k: "%.2e" % (abs(float(got[k]) - v) / max(abs(v), 1e-6)) for k, v in gold["losses"].items() if k in got})
    print(tag, "stats:", {k: "%.2e" % P.rel(out.extras[k][:, :8, :16], v) for k, v in gold["stats"].items()})
    td, tg = P._sumsq(gd), P._sumsq(gg)
    print(tag, "D sumsq:", {k: "%.2e" % (abs(td[k] - v) / v) for k, v in gold["d_grad_sumsq"].items()})
    print(tag, "G sumsq:", {k: "%.2e" % (abs(tg[k] - v) / v) for k, v in gold["g_grad_sumsq"].items()})
    nz = gold.get("noise", {})
    print(tag, "D slices:", {n: "%.2e" % P.rel(gd[n].flatten()[:64], s) for n, s in gold["d_grad_slices"].items()})
    print(tag, "G slices:", {n: "%.2e (noise %.1e)" % (P.rel(gg[n].flatten()[:64], s), nz.get("g_slices", gold.get("g_grad_slice_noise", {})).get(n, -1))
                             for n, s in gold["g_grad_slices"].items()})
    y = out.extras["y_hat"].squeeze(1).float().cpu()
    if "y_hat" in gold:
        print(tag, "y_hat rel %.2e" % P.rel(y, gold["y_hat"]), "mel rel %.2e" % P.rel(out.extras["y_hat_mel"], gold["y_hat_mel"]))
    else:
        print(tag, "y_hat strided rel %.2e" % P.rel(y[:, ::997], gold["y_hat_strided"]),
              "mel %.2e" % P.rel(out.extras["y_hat_mel"][:, ::7, ::3], gold["y_hat_mel_strided"]),
              "logits", ["%.1e" % P.rel(o[:, :16], r) for o, r in zip(out.extras["d_logits"], gold["d_logits_head"])])


c1 = torch.load(os.path.join(G, "s2_c1.pt"), weights_only=False)
c2 = torch.load(os.path.join(G, "s2_c2.pt"), weights_only=False)
for gold, dt, tag in ((c1, torch.bfloat16, "C1 bf16"), (c2, torch.float32, "C2 f32 "), (c2, torch.bfloat16, "C2 bf16")):
    eng = P._engine(gpu, dt)
    out, gd, gg = P._step(eng, gpu, gold["config"])
    table(gold, out, gd, gg, tag)
    del eng
    torch.cuda.empty_cache()

from easevoice_trainer_amd.train.s1_engine import S1Engine
gold = torch.load(os.path.join(G, "s1_c3.pt"), weights_only=False)
c = gold["config"]
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
for dt in (torch.float32, torch.bfloat16):
    eng = S1Engine(cfg, gpu, dt)
    fill_module(eng.model, 3)
    eng.model.eval()
    b = s1_batch(c["B"], c["x_len"], c["y_len"], seed=c["seed"])
    loss, acc = eng.model.forward_old(b["phoneme_ids"].to(gpu), torch.tensor(c["x_lens"]).to(gpu), b["semantic_ids"].to(gpu),
                                      torch.tensor(c["y_lens"]).to(gpu), b["bert_feature"].to(gpu))
    loss.backward()
    torch.cuda.synchronize()
    params = dict(eng.model.named_parameters())
    print("s1 c3", dt, "loss rel %.2e" % (abs(float(loss) - gold["loss"]) / gold["loss"]), "acc", float(acc), gold["acc"])
    print("  slices:", {n: "%.2e |g|max %.1e" % (P.rel(params[n].grad.flatten()[:96], s), float(s.abs().max())) for n, s in gold["grad_slices"].items()})
    tot = {}
    for n, p in params.items():
        top = ".".join(n.split(".")[:3]) if n.startswith("h.layers") else n.split(".")[0]
        tot[top] = tot.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    worst = sorted(((abs(tot[k] - v) / v, k) for k, v in gold["grad_sumsq"].items()), reverse=True)[:6]
    print("  worst sumsq:", [("%.2e" % e, k) for e, k in worst])
    del eng
