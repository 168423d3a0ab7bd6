"""GPU: the library's side streams are HIP streams of its own (hip/lib.py::role_stream), not torch's pooled ones.

torch.cuda.Stream() hands out one of 32 streams per device round-robin; a process that has asked for more (every engine takes
a few, this test tier takes hundreds) gets the same HIP stream under two names.  With the s2 step's lanes that ended in a
segmentation fault inside hipStreamEndCapture (tests in the order streams -> book-pipe -> graph, round 6)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_role_streams_are_not_pool_streams_and_repeat_only_within_a_role(gpu):
    from easevoice_trainer_amd.hip import lib as L

    pool = {torch.cuda.Stream(device=gpu).cuda_stream for _ in range(80)}
    assert len(pool) <= 32                                    # the pool wraps: that is the premise
    roles = {r: [L.role_stream(gpu, r, ring=n) for _ in range(2 * n)] for r, n in (("t_a", 4), ("t_b", 2), ("t_c", 1))}
    seen = {}
    for r, sts in roles.items():
        n = len(sts) // 2
        ptrs = [s.cuda_stream for s in sts]
        assert ptrs[:n] == ptrs[n:] and len(set(ptrs)) == n, (r, ptrs)      # a ring of n, then the same n again
        for p in set(ptrs):
            assert p not in pool and p != 0, (r, p)
            assert p not in seen, (r, seen[p])                # never shared between roles
            seen[p] = r
    # they are ordinary streams for torch: work, events, waits
    a, b = roles["t_a"][0], roles["t_b"][0]
    x = torch.ones(1 << 20, device=gpu)
    a.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(a):
        y = x * 2
    b.wait_stream(a)
    with torch.cuda.stream(b):
        z = y + 1
    torch.cuda.current_stream(gpu).wait_stream(b)
    y.record_stream(b)
    z.record_stream(torch.cuda.current_stream(gpu))
    assert float(z.sum()) == 3.0 * (1 << 20)


def test_engine_side_streams_have_their_roles(gpu):
    """lanes, weight-gradient side streams and the bookkeeping stream of an s2 engine are pairwise different HIP streams and
    none of them is a pooled one"""
    from easevoice_trainer_amd.hip import conv as HC
    from easevoice_trainer_amd.hip import disc as HD

    pool = {torch.cuda.Stream(device=gpu).cuda_stream for _ in range(64)}
    m1, m2 = HC.EvtConv1d(32, 32, 3, padding=1), HC.EvtConv1d(32, 32, 3, padding=1)
    b1, b2 = HC.WeightBank(m1.to(gpu), torch.bfloat16, gpu), HC.WeightBank(m2.to(gpu), torch.bfloat16, gpu)
    old = (HD.MPD_STREAMS, HD.ENC_STREAM, HD.DEC_STREAM)
    HD.MPD_STREAMS, HD.ENC_STREAM, HD.DEC_STREAM = 2, True, True
    try:
        ptrs = [b1.side_stream().cuda_stream, b2.side_stream().cuda_stream, HD.enc_lane(gpu).cuda_stream,
                HD.dec_lane(gpu).cuda_stream]
    finally:
        HD.MPD_STREAMS, HD.ENC_STREAM, HD.DEC_STREAM = old
    assert len(set(ptrs)) == 4 and not (set(ptrs) & pool), ptrs
    assert HD._branches(gpu, 6)[1][0].cuda_stream == ptrs[2]            # the discriminators' lane is the encoder's lane
