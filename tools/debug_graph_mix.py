"""Debug aid (round 2): which interleavings of eager steps and HIP-graph replays of the s2 step produce non-finite
gradients?  Synthetic ragged batches, no reader."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from easevoice_trainer_amd.module.mel_processing import spectrogram_torch
from easevoice_trainer_amd.train.s2_engine import S2Engine

dev = torch.device("cuda:0")
hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))


def batch(T, Tt, lens, tl, seed):
    B = len(lens)
    lens, tl = torch.tensor(lens), torch.tensor(tl)
    g = torch.Generator().manual_seed(seed)
    wav = (torch.rand(B, 1, T * 640, generator=g) - 0.5) * 0.4
    ssl = torch.randn(B, 768, T, generator=g)
    text = torch.randint(0, 732, (B, Tt), generator=g)
    for b in range(B):
        wav[b, :, lens[b] * 640:] = 0; ssl[b, :, lens[b]:] = 0; text[b, tl[b]:] = 0
    spec = torch.zeros(B, 1025, T, device=dev)
    for b in range(B):
        s = spectrogram_torch(wav[b, :, :lens[b] * 640].to(dev), 2048, 32000, 640, 2048)
        spec[b, :, :s.size(2)] = s[0]
    return (ssl.to(dev), spec, lens.to(dev), wav.to(dev), text.to(dev), tl.to(dev))


def fresh():
    torch.manual_seed(0)
    eng = S2Engine(hps, dev, torch.bfloat16)
    eng.build_optimizers()
    cb = eng.net_g.quantizer.vq.layers[0]._codebook
    cb.embed.normal_(); cb.inited.fill_(1.0)
    eng.enable_graphs(warmup_steps=2)
    return eng


def run(eng, tag, args, eager=False):
    if eager:
        eng.graphs_enabled = False
    out = eng.step(*args)
    eng.graphs_enabled = True
    torch.cuda.synchronize()
    bad = sorted({".".join(n.split(".")[:2]) for n, p in eng.net_g.named_parameters() if not torch.isfinite(p).all()})
    badd = sum(int(not torch.isfinite(p).all()) for p in eng.net_d.parameters())
    print(f"  {tag:28s} gen_all={float(out.gen_all):9.3f} disc={float(out.disc):7.3f} nanG={bad} nanD={badd}", flush=True)
    return bool(bad or badd)


A = lambda s, lens=(170, 100, 100, 47), tl=(30, 1, 1, 8): batch(172, 30, lens, tl, s)
Bb = lambda s: batch(78, 13, (77, 77, 65, 40), (13, 13, 11, 7), s)

def main():
    print("E1: capture A, then replays only (varying lens)")
    eng = fresh()
    for i in range(3):
        run(eng, f"A warm/capture {i}", A(i))
    for i, lens in enumerate([(170, 102, 102, 40), (170, 65, 47, 40), (170, 170, 170, 170), (60, 50, 45, 40)]):
        run(eng, f"A replay lens={lens}", A(10 + i, lens, (30, 18, 18, 7)))
    del eng
    
    print("E2: capture A, eager same shape A, replay A")
    eng = fresh()
    for i in range(3):
        run(eng, f"A warm/capture {i}", A(i))
    run(eng, "A eager", A(20), eager=True)
    run(eng, "A eager", A(21), eager=True)
    run(eng, "A replay", A(22))
    run(eng, "A replay", A(23))
    del eng
    
    print("E3: capture A, eager other shape B, replay A")
    eng = fresh()
    for i in range(3):
        run(eng, f"A warm/capture {i}", A(i))
    run(eng, "B eager", Bb(30), eager=True)
    run(eng, "B eager", Bb(31), eager=True)
    run(eng, "A replay", A(32))
    run(eng, "A replay", A(33))


if __name__ == "__main__":
    main()
