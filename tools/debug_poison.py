"""Debug aid (round 2): run one eager s2 step with ragged lengths after filling the caching allocator's free blocks with
NaN bit patterns.  A kernel that leaves part of an output unwritten (and relies on a later multiply-by-zero mask) then
shows up as NaN gradients instead of passing by luck."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from easevoice_trainer_amd.module.mel_processing import spectrogram_torch
from easevoice_trainer_amd.train.s2_engine import S2Engine

dev = torch.device("cuda:0")
hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
torch.manual_seed(0)
eng = S2Engine(hps, dev, torch.bfloat16)
eng.build_optimizers()
cb = eng.net_g.quantizer.vq.layers[0]._codebook
cb.embed.normal_(); cb.inited.fill_(1.0)
B, T, Tt = 4, 172, 30
lens = torch.tensor([170, 102, 102, 40]); tl = torch.tensor([30, 18, 18, 7])
g = torch.Generator().manual_seed(7)
wav = (torch.rand(B, 1, T * 640, generator=g) - 0.5) * 0.4
ssl = torch.randn(B, 768, T, generator=g)
text = torch.randint(0, 732, (B, Tt), generator=g)
for b in range(B):
    wav[b, :, lens[b] * 640:] = 0; ssl[b, :, lens[b]:] = 0; text[b, tl[b]:] = 0
spec = torch.zeros(B, 1025, T, device=dev)
for b in range(B):
    s = spectrogram_torch(wav[b, :, :lens[b] * 640].to(dev), 2048, 32000, 640, 2048)
    spec[b, :, :s.size(2)] = s[0]
args = (ssl.to(dev), spec, lens.to(dev), wav.to(dev), text.to(dev), tl.to(dev))


def report(tag, out):
    torch.cuda.synchronize()
    vals = {f: float(getattr(out, f)) for f in ("disc", "gen", "fm", "mel", "kl")}
    bad_g = [n for n, p in eng.net_g.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    bad_d = [n for n, p in eng.net_d.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(tag, vals, "nan grads G:", len(bad_g), "D:", len(bad_d), flush=True)
    pref = sorted({".".join(n.split(".")[:3]) for n in bad_g})
    print("   G prefixes:", pref[:60], flush=True)
    ex = {k: bool(torch.isfinite(v).all()) for k, v in out.extras.items() if torch.is_tensor(v)}
    print("   finite extras:", ex, flush=True)


def poison():
    big = torch.empty(int(6e9) // 4, dtype=torch.int32, device=dev).fill_(0x7FC07FC0)
    small = [torch.empty(128 * 1024, dtype=torch.int32, device=dev).fill_(0x7FC07FC0) for _ in range(6000)]
    mid = [torch.empty(3 * 1024 * 1024, dtype=torch.int32, device=dev).fill_(0x7FC07FC0) for _ in range(300)]
    torch.cuda.synchronize()
    del big, small, mid


report("clean ", eng.step(*args, do_opt=False))
torch.cuda.empty_cache()
poison()
report("poison", eng.step(*args, do_opt=False))
if "--anomaly" in sys.argv:
    poison()
    with torch.autograd.detect_anomaly(check_nan=True):
        try:
            report("anomaly", eng.step(*args, do_opt=False))
        except Exception as e:
            print("anomaly:", str(e)[:1500], flush=True)
