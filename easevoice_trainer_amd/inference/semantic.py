"""The `token` step of the reference's normalisation (src/normalization/normalize.py:181-211): semantic codes of every
extracted CN-HuBERT feature file, written as 6-name2semantic.tsv -- the s1 stage's training targets.

The reference pins this step to the CPU (normalize.py:58-60); here `extract_latent` runs through SoVITSVoice on the GPU.
Format kept byte for byte: header `item_name\\tsemantic_audio`, one `name\\tc0 c1 ...` line per item whose
`4-cnhubert/<name>.pt` exists, items in the order given, trailing newline."""
import os

import torch


def write_semantic_tsv(extract_latent, names, hubert_dir, out_path, device="cpu"):
    """extract_latent: callable ssl [1, 768, T] -> codes [1, 1, T'] (SoVITSVoice.extract_latent or
    SynthesizerTrn.extract_latent); names: item names (the wav basenames of the refinement list)."""
    opt = ["item_name\tsemantic_audio"]
    for name in names:
        path = os.path.join(hubert_dir, name + ".pt")
        if not os.path.exists(path):
            continue
        ssl = torch.load(path, map_location="cpu").float().to(device)
        codes = extract_latent(ssl)
        opt.append("%s\t%s" % (name, " ".join(str(int(i)) for i in codes[0, 0, :].tolist())))
    with open(out_path, "w", encoding="utf8") as f:
        f.write("\n".join(opt) + "\n")
    return len(opt) - 1
