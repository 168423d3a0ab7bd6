"""Worker of tests/test_host_cpu.py::test_reduce_scatter_all_gather_equals_all_reduce (gloo, CPU): the same ragged buffer
summed with GradReducer's all-reduce buckets and with its reduce-scatter + all-gather variant."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from easevoice_trainer_amd.dist import GradReducer

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = torch.Generator().manual_seed(7 + rank)
    for n in (1, world - 1, world, 1000, 4099, 10007):          # below / at / above the world size, ragged tails
        x = torch.randn(n, generator=g)
        a, b = x.clone(), x.clone()
        GradReducer(world, bucket_bytes=4 * 1024, rsag=False).all_reduce(a)
        GradReducer(world, bucket_bytes=4 * 1024, rsag=True).all_reduce(b)
        # one reduction over the ranks per element either way; gloo's ring adds in a rank-rotated order per chunk, so
        # the two sums may differ in the last bit
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6), (n, float((a - b).abs().max()))
        ref = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(ref, x)
        assert torch.allclose(b, torch.stack(ref).sum(0), rtol=1e-6, atol=1e-6), n
        # "auto": reduce-scatter + all-gather only for buckets of at least rsag_min_bytes, all-reduce below -- and the
        # reducer says which it used
        c = x.clone()
        r = GradReducer(world, bucket_bytes=4 * 1024, rsag="auto", rsag_min_bytes=2 * 1024)
        r.all_reduce(c)
        assert torch.allclose(c, a, rtol=1e-6, atol=1e-6), n
        full, tail = divmod(n, 1024)
        want_rsag = full + (1 if tail >= 512 else 0)
        assert r.stats["rs_ag"] == want_rsag and r.stats["all_reduce"] == (1 if 0 < tail < 512 else 0), (n, r.stats)
        assert r.stats["bytes"] == 4 * n and "reduce-scatter" in r.describe()
        assert r.rsag_verified is (True if want_rsag else None), (n, r.rsag_verified)    # checked at the first eligible bucket
    # the start-up check of auto mode: a pair that does not reproduce all_reduce on ONE rank puts EVERY rank
    # back on all-reduce, and the sums stay right
    x = torch.randn(4099, generator=g)
    a, c = x.clone(), x.clone()
    GradReducer(world, bucket_bytes=4 * 1024, rsag=False).all_reduce(a)
    r = GradReducer(world, bucket_bytes=4 * 1024, rsag="auto", rsag_min_bytes=2 * 1024)
    good = r._rs_ag

    def broken(body, chunk):
        good(body, chunk)
        if rank == 1:
            body[0] += 1.0

    r._rs_ag = broken
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r.all_reduce(c)
    assert r.rsag_verified is False and r.stats["rs_ag"] == 0 and r.stats["all_reduce"] == 5, (r.rsag_verified, r.stats)
    assert torch.allclose(c, a, rtol=1e-6, atol=1e-6)
    assert r.plan(1 << 20) == [(4 * 1024, "all-reduce")] * 1024
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
