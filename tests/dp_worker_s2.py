"""Worker of tests/test_zz_dp_overlap_gpu.py: one rank of a data-parallel run of the s2 engine on cuda:0.  Both ranks of
the test share the ONE GPU of the box, so the collectives run over gloo (RCCL refuses two ranks on one device); everything
else -- the per-sub-model backward pieces, the side-stream reductions, the cut backward of the generator, graph replay with
reductions between the graphs -- is the code the multi-GPU run executes.
argv: out_prefix mode steps      mode: overlap | sync | single (one process, both ranks' items as one batch)"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def batch(rank, B, T, tt):
    g = torch.Generator().manual_seed(100 + rank)
    wav = torch.rand(B, 1, T * 640, generator=g) - 0.5
    ssl = torch.randn(B, 768, T, generator=g)
    text = torch.randint(0, 732, (B, tt), generator=g)
    eps = torch.randn(B, 192, T, generator=g)
    ids = torch.randint(0, T - 32 + 1, (B,), generator=g)
    return wav, ssl, text, eps, ids


def main():
    out, mode, steps, graphs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ["EVT_DP_OVERLAP"] = "0" if mode == "sync" else "1"
    import easevoice_trainer_amd  # noqa: F401  (sets the graph-capture switch before HIP loads)
    from easevoice_trainer_amd.dist import GradReducer
    from easevoice_trainer_amd.module.mel_processing import spectrogram_torch
    from easevoice_trainer_amd.train.s2_engine import S2Engine

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    reducer = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        reducer = GradReducer(world, bucket_bytes=8 << 20)
    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    hps["model"]["p_dropout"] = 0.0
    torch.manual_seed(0)
    eng = S2Engine(hps, dev, torch.float32, reducer=reducer)
    for m in eng.net_g.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    cb = eng.net_g.quantizer.vq.layers[0]._codebook
    cb.embed.copy_(torch.randn(cb.embed.shape, generator=torch.Generator().manual_seed(3)))
    cb.inited.fill_(1.0)
    if world > 1:
        reducer.broadcast_params(eng.rt_g.arena.param)
        reducer.broadcast_params(eng.rt_d.arena.param)
        assert eng.overlap == (mode == "overlap")
    eng.build_optimizers()
    if graphs:
        eng.enable_graphs(warmup_steps=1)
    B, T, tt = 1, 64, 16
    if world > 1:
        parts = [batch(rank, B, T, tt)]
    else:
        parts = [batch(0, B, T, tt), batch(1, B, T, tt)]
    wav, ssl, text, eps, ids = [torch.cat(x).to(dev) for x in zip(*parts)]
    n = wav.size(0)
    lens, tl = torch.full((n,), T, device=dev), torch.full((n,), tt, device=dev)
    spec = spectrogram_torch(wav.squeeze(1), 2048, 32000, 640, 2048)
    losses = []
    p0 = dict(g=eng.rt_g.arena.param.detach().cpu().clone(), d=eng.rt_d.arena.param.detach().cpu().clone())
    # first a step WITHOUT the optimisers: the reduced gradient arenas are what the exchange produced
    o = eng.step(ssl, spec, lens, wav, text, tl, eps=eps, ids_slice=ids, do_opt=False)
    torch.cuda.synchronize()
    grads = dict(g=eng.rt_g.arena.grad.detach().cpu().clone(), d=eng.rt_d.arena.grad.detach().cpu().clone())
    for _ in range(steps):
        o = eng.step(ssl, spec, lens, wav, text, tl, eps=eps, ids_slice=ids)
        losses.append([float(o.disc), float(o.gen), float(o.fm), float(o.mel), float(o.kl)])
    torch.cuda.synchronize()
    torch.save(dict(g=eng.rt_g.arena.param.detach().cpu(), d=eng.rt_d.arena.param.detach().cpu(), losses=losses,
                    grads=grads, p0=p0), f"{out}_{mode}_{rank}.pt")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
