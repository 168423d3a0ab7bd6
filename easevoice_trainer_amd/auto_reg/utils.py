"""Host helpers of the s1 DPO branch (src/easevoice/soundstorm/auto_reg/models/utils.py:160-228)."""
import torch
from torch.nn import functional as F


def make_reject_y(y_o, y_lens):
    """Rejected semantic sequences: each row with a random span repeated once (models/utils.py:185-228).

    Same draws, in the same order, on torch's global CPU generator as the reference (per item `randint(0, 1, (1,))`,
    whose only value selects the repeat rule, then `randint(0, len(row), (2,))` over the PADDED row), so a seeded run
    rejects the same sequences.  The cut points stay on the host: the rows are assembled on the device by slicing, with
    no device->host read.  Returns (reject_y [B, max new length] zero-padded, reject_y_lens = whole new row lengths,
    on y_lens' device)."""
    rows, lens = [], []
    width = y_o.size(1)
    for b in range(len(y_lens)):
        torch.randint(0, 1, size=(1,))
        i0, i1 = sorted(torch.randint(0, width, size=(2,)).tolist())
        row = y_o[b]
        rows.append(torch.cat([row[:i1], row[i0:]]))       # == pre + span + span + rest
        lens.append(width + i1 - i0)
    out = torch.stack([F.pad(r, (0, max(lens) - r.numel())) for r in rows], dim=0)
    return out, torch.tensor(lens, device=y_lens.device)


def dpo_loss(chosen_logps, rejected_logps, beta=0.2):
    """reference-free DPO term, models/utils.py:160-173 with reference_free=True: mean(-logsigmoid(beta*(A-R)))"""
    return (-F.logsigmoid(beta * (chosen_logps - rejected_logps))).mean()
