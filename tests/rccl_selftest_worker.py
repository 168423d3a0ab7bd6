"""Worker of tests/test_zz_rccl_selftest_gpu.py: the data-parallel code paths on a ONE-rank RCCL group (backend "nccl" on
ROCm).  A box with a single GPU cannot give RCCL two ranks, but a one-rank communicator runs the same library calls on
the same side stream: communicator init under HSA_ENABLE_IPC_MODE_LEGACY=0, all-reduce, reduce-scatter into the shard
buffer + all-gather, broadcast, the collectives between HIP-graph replays, the hook-driven pieces inside the s1 backward.
With one rank every sum is the identity, so the result must equal the run without a reducer.
argv: out_prefix mode      mode: prims | s2_plain | s2_rccl | s1_plain | s1_rccl"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def init_rccl():
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return dist


def prims(out):
    from easevoice_trainer_amd.dist import GradReducer

    dist = init_rccl()
    dev = torch.device("cuda", 0)
    res = {}
    for mode in ("0", "1"):
        red = GradReducer(1, bucket_bytes=1 << 20, rsag=mode, force=True)
        assert red.active
        g = torch.arange(3 * (1 << 18) + 5, device=dev, dtype=torch.float32) * 1e-3      # 3 full buckets + a ragged one
        ref = g.clone()
        # async on the side stream, behind work of the compute stream that produced the buffer
        g.mul_(2.0)
        red.all_reduce(g, async_op=True, average=True)
        red.wait()
        torch.cuda.synchronize()
        res[f"all_reduce_rsag{mode}"] = bool(torch.equal(g, ref * 2.0))
        res[f"stats_rsag{mode}"] = dict(red.stats)
        p = torch.randn(1 << 20, device=dev)
        q = p.clone()
        red.broadcast_params(p)
        res[f"broadcast_rsag{mode}"] = bool(torch.equal(p, q))
        m = torch.tensor([1.0, 2.0, 3.0], device=dev)
        red.all_reduce_scalars(m)
        res[f"scalars_rsag{mode}"] = m.tolist()
    # a collective between two replays of a captured graph that produces / consumes the buffer
    red = GradReducer(1, bucket_bytes=1 << 20, rsag="1", force=True)
    buf = torch.zeros(1 << 19, device=dev)
    acc = torch.zeros(1 << 19, device=dev)
    torch.cuda.synchronize()
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    pool = torch.cuda.graph_pool_handle()
    with torch.cuda.graph(g1, pool=pool):
        buf.add_(1.0)
    with torch.cuda.graph(g2, pool=pool):
        acc.add_(buf)
    for _ in range(5):
        g1.replay()
        red.all_reduce(buf, async_op=True)
        red.wait()
        g2.replay()
    torch.cuda.synchronize()
    res["graph_interleave"] = [float(buf[0]), float(acc[-1])]      # 5 and 1 + 2 + 3 + 4 + 5 = 15
    json.dump(res, open(out + "_prims.json", "w"))
    dist.destroy_process_group()


def s2(out, rccl):
    import easevoice_trainer_amd  # noqa: F401
    from dp_worker_s2 import batch
    from easevoice_trainer_amd.dist import GradReducer
    from easevoice_trainer_amd.module.mel_processing import spectrogram_torch
    from easevoice_trainer_amd.train.s2_engine import S2Engine

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    reducer = None
    if rccl:
        dist = init_rccl()
        reducer = GradReducer(1, bucket_bytes=8 << 20, rsag="1", force=True)
    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    hps["model"]["p_dropout"] = 0.0
    torch.manual_seed(0)
    eng = S2Engine(hps, dev, torch.float32, reducer=reducer)
    assert eng.overlap == rccl
    for m in eng.net_g.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    cb = eng.net_g.quantizer.vq.layers[0]._codebook
    cb.embed.copy_(torch.randn(cb.embed.shape, generator=torch.Generator().manual_seed(3)))
    cb.inited.fill_(1.0)
    if rccl:
        reducer.broadcast_params(eng.rt_g.arena.param)
        reducer.broadcast_params(eng.rt_d.arena.param)
    eng.build_optimizers()
    eng.enable_graphs(warmup_steps=1)
    B, T, tt = 2, 64, 16
    wav, ssl, text, eps, ids = [x.to(dev) for x in batch(0, B, T, tt)]
    lens, tl = torch.full((B,), T, device=dev), torch.full((B,), tt, device=dev)
    spec = spectrogram_torch(wav.squeeze(1), 2048, 32000, 640, 2048)
    p0 = dict(g=eng.rt_g.arena.param.detach().cpu().clone(), d=eng.rt_d.arena.param.detach().cpu().clone())
    o = eng.step(ssl, spec, lens, wav, text, tl, eps=eps, ids_slice=ids, do_opt=False)
    torch.cuda.synchronize()
    grads = dict(g=eng.rt_g.arena.grad.detach().cpu().clone(), d=eng.rt_d.arena.grad.detach().cpu().clone())
    losses = []
    for _ in range(4):                      # eager, capture, two replays
        o = eng.step(ssl, spec, lens, wav, text, tl, eps=eps, ids_slice=ids)
        losses.append([float(o.disc), float(o.gen), float(o.fm), float(o.mel), float(o.kl)])
    torch.cuda.synchronize()
    torch.save(dict(g=eng.rt_g.arena.param.detach().cpu(), d=eng.rt_d.arena.param.detach().cpu(), losses=losses, grads=grads,
                    p0=p0, replayed=eng.graph_steps["replayed"], stats=dict(reducer.stats) if rccl else None,
                    graphs=len(eng._program())), out + ("_s2_rccl.pt" if rccl else "_s2_plain.pt"))
    if rccl:
        dist.destroy_process_group()


def s1(out, rccl):
    import yaml
    from easevoice_trainer_amd.dist import GradReducer
    from easevoice_trainer_amd.train.s1_engine import S1Engine
    from util_fill import fill_module, s1_batch

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    reducer = None
    if rccl:
        dist = init_rccl()
        reducer = GradReducer(1, bucket_bytes=16 << 20, rsag="1", force=True)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    cfg["model"]["dropout"] = 0.0
    torch.manual_seed(0)
    eng = S1Engine(cfg, dev, torch.float32, reducer=reducer)
    fill_module(eng.model, 3)
    eng.bank.mark_dirty()
    eng.model.eval()
    if rccl:
        reducer.broadcast_params(eng.arena.param)
    b = {k: v.to(dev) for k, v in s1_batch(2, 64, 192).items()}
    stepped = [bool(eng.micro_step(b, i)[2]) for i in range(5)]
    torch.cuda.synchronize()
    torch.save(dict(p=eng.arena.param.detach().cpu(), stepped=stepped, stats=dict(reducer.stats) if rccl else None),
               out + ("_s1_rccl.pt" if rccl else "_s1_plain.pt"))
    if rccl:
        dist.destroy_process_group()


if __name__ == "__main__":
    out, mode = sys.argv[1], sys.argv[2]
    if mode == "prims":
        prims(out)
    elif mode.startswith("s2"):
        s2(out, mode == "s2_rccl")
    else:
        s1(out, mode == "s1_rccl")
