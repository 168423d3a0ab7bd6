"""Helpers of the s2 path, channels-last ([B, T, C]) unless noted.
Mirrors src/easevoice/module/commons.py:42-58 (slice_segments / rand_slice_segments, here one gather
instead of a per-item Python loop), :115-119 (sequence_mask)."""
import torch


def sequence_mask(length, max_length=None):
    if max_length is None:
        max_length = int(length.max())
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


def slice_segments(x, ids_str, segment_size=4):
    """x [B, T, C] -> [B, segment_size, C], window starting at ids_str[b] (commons.py:42-48)"""
    idx = ids_str.view(-1, 1) + torch.arange(segment_size, device=x.device).view(1, -1)      # [B, S]
    return torch.gather(x, 1, idx.unsqueeze(-1).expand(-1, -1, x.size(2)))


def slice_segments_1d(x, ids_str, segment_size):
    """x [B, T] -> [B, segment_size]"""
    idx = ids_str.view(-1, 1) + torch.arange(segment_size, device=x.device).view(1, -1)
    return torch.gather(x, 1, idx)


def rand_slice_segments(x, x_lengths=None, segment_size=4):
    """commons.py:51-58: ids = floor(U[0,1) * (len - seg + 1))"""
    b, t, _ = x.shape
    if x_lengths is None:
        x_lengths = torch.full((b,), t, device=x.device)
    ids_str_max = x_lengths - segment_size + 1
    ids_str = (torch.rand([b], device=x.device) * ids_str_max).to(dtype=torch.long)
    return slice_segments(x, ids_str, segment_size), ids_str
