"""Deterministic, name-keyed parameter fill shared by the golden generator (reference modules, build
container) and the GPU parity tests (this repo's modules, GPU box).  torch's CPU generator is bit-stable for a
given torch version, and both sides run the same image, so the two state_dicts are identical tensors."""
import zlib

import torch


def fill_tensor(name: str, shape, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(name.encode()) % 1000003)
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    if leaf == "weight_g":
        return torch.rand(shape, generator=g) + 0.5
    if leaf == "gamma" or (leaf == "weight" and len(shape) == 1):      # LayerNorm scales
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf in ("bias", "beta", "in_proj_bias"):
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == "alpha":
        return torch.ones(shape) * 0.9
    if leaf == "inited":
        return torch.ones(shape)
    if leaf == "cluster_size":
        return torch.ones(shape)
    if leaf in ("embed", "embed_avg"):
        return torch.randn(shape, generator=g)
    if leaf.startswith("emb_rel"):
        return 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    scale = 1.0 / max(fan_in, 1) ** 0.5
    if ".proj." in name or ".post." in name:
        scale *= 0.05      # keeps (m, logs) statistics O(0.1): exp(+-2 logs) stays well conditioned in fp32
    return torch.randn(shape, generator=g) * scale


def fill_module(module: torch.nn.Module, seed: int) -> None:
    sd = module.state_dict()
    new = {k: fill_tensor(k, v.shape, seed).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)


def s2_batch(B, T, t_text, seed=1234):
    """synthetic s2 batch of SURVEY §8(d): wav ~ U(-0.5, 0.5), ssl ~ N(0,1), text ~ randint(732), full lengths.
    Returns CPU tensors in the reference's layouts; spec is filled in by the caller (it needs an STFT)."""
    g = torch.Generator().manual_seed(seed)
    hop = 640
    wav = torch.rand(B, 1, T * hop, generator=g) - 0.5
    ssl = torch.randn(B, 768, T, generator=g)
    text = torch.randint(0, 732, (B, t_text), generator=g)
    eps = torch.randn(B, 192, T, generator=g)
    ids_slice = torch.randint(0, T - 32 + 1, (B,), generator=g)
    return dict(wav=wav, ssl=ssl, text=text, eps=eps, ids_slice=ids_slice,
                lengths=torch.full((B,), T, dtype=torch.long), text_lengths=torch.full((B,), t_text, dtype=torch.long))


def decode_inputs(t_codes=30, t_text=20, t_ref=(50, 37), seed=4321):
    """inputs of SynthesizerTrn.decode / extract_latent for the inference fixtures: codes [1,1,Tc] (25 Hz), text ids,
    two reference spectrograms (the list form averages their style vectors), the prior noise [1,192,2*Tc+pad] and an
    ssl feature map for extract_latent"""
    g = torch.Generator().manual_seed(seed)
    return dict(codes=torch.randint(0, 1024, (1, 1, t_codes), generator=g),
                text=torch.randint(0, 732, (1, t_text), generator=g),
                refers=[torch.rand(1, 1025, t, generator=g) * 2.0 for t in t_ref],
                noise=torch.randn(1, 192, 2 * t_codes + 8, generator=g),
                ssl=torch.randn(2, 768, 46, generator=g))


def s1_batch(B, x_len, y_len, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return dict(phoneme_ids=torch.randint(0, 732, (B, x_len), generator=g),
                phoneme_ids_len=torch.full((B,), x_len, dtype=torch.long),
                semantic_ids=torch.randint(0, 1024, (B, y_len), generator=g),
                semantic_ids_len=torch.full((B,), y_len, dtype=torch.long),
                bert_feature=torch.randn(B, 1024, x_len, generator=g))
