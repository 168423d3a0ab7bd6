"""seeded inputs of the s1 decoding fixtures, shared by the generator (make_golden_s1.py) and the tests"""
import torch


def infer_inputs(seed=99):
    g = torch.Generator().manual_seed(seed)
    return dict(x=torch.randint(0, 732, (1, 24), generator=g), bert=torch.randn(1, 1024, 24, generator=g),
                prompts=torch.randint(0, 1024, (1, 12), generator=g),
                q=torch.empty(64, 1025).exponential_(1, generator=g))


def batch_infer_inputs(seed=77):
    """three texts of different lengths sharing one prompt; per-row sampling noise q[step][row][v] whose EOS column makes
    row 2 stop at step 7 and row 1 at step 15 (rows leave the batch from the back, so the survivors stay a prefix)"""
    g = torch.Generator().manual_seed(seed)
    lens = [24, 17, 9]
    x = [torch.randint(0, 732, (n,), generator=g) for n in lens]
    bert = [torch.randn(1024, n, generator=g) for n in lens]
    prompt = torch.randint(0, 1024, (1, 12), generator=g)
    q = torch.empty(64, 3, 1025).exponential_(1, generator=g)
    q[7, 2, 1024] = 1e-30
    q[15, 1, 1024] = 1e-30
    return dict(x=x, bert=bert, x_lens=torch.tensor(lens), prompts=prompt.expand(3, -1).contiguous(), q=q)


def pipeline_inputs(seed=55):
    """two text fragments (prompt phones + own phones) for the s1 -> s2 chain; EOS is forced at step 14 for row 0 and at
    step 9 for row 1 through the noise table (top_k covers the whole vocabulary in these runs)"""
    g = torch.Generator().manual_seed(seed)
    lens = [24, 17]
    ids = [torch.randint(0, 732, (n,), generator=g) for n in lens]
    bert = [torch.randn(1024, n, generator=g) for n in lens]
    q = torch.empty(32, 2, 1025).exponential_(1, generator=g)
    q[14, 0, 1024] = 1e-30
    q[9, 1, 1024] = 1e-30
    return dict(all_ids=ids, bert=bert, batch_phones=[i[7:] for i in ids], prompt=torch.randint(0, 1024, (1, 12), generator=g),
                q=q)
