"""Seeded synthetic feature directory in the reference's on-disk layout (2-name2text.txt, 3-bert, 4-cnhubert, 5-wav32k,
6-name2semantic.tsv).  Shared by tests/golden/make_golden_data.py (which runs the REFERENCE's readers over it) and the
reader tests (which run ours over an identical copy), so only the expected outputs have to be committed."""
import os
import wave

import numpy as np
import torch

HOP, SR, NFFT = 640, 32000, 2048

# (name, seconds, hubert frames relative to the spectrogram's, listed in name2text?, phones ok?, has bert?)
ITEMS = [
    ("a_000.wav", 1.30, 0, True, True, True),
    ("a_001.wav", 2.05, -1, True, True, True),     # ssl one frame short -> replicate-padded by one
    ("a_002.wav", 0.95, 0, True, True, False),
    ("a_003.wav", 3.40, -1, True, True, True),
    ("a_004.wav", 0.30, 0, True, True, True),      # shorter than 0.6 s -> filtered by duration
    ("a_005.wav", 1.75, 0, True, False, True),     # phoneme outside the table -> skipped
    ("a_006.wav", 1.10, 0, False, True, True),     # not listed in 2-name2text.txt
    ("a_007.wav", 2.60, 1, True, True, True),      # ssl one frame long -> padded again (the reference's rule)
    ("a_008.wav", 1.55, 0, True, True, True),
    ("a_009.wav", 0.80, -1, True, True, False),
]
BROKEN = "a_010.wav"  # listed and long enough by file size, but not a RIFF file -> placeholder item


def frames_of(n_samples):
    return (n_samples + (NFFT - HOP) - NFFT) // HOP + 1


def build_feature_dir(root, symbols, seed=77):
    """Writes the directory and returns {name: n_samples}.  `symbols` is the phoneme table (list, index == id); phones
    are drawn from it."""
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    for d in ("3-bert", "4-cnhubert", "5-wav32k"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    lines, sem_lines, sizes = [], ["item_name\tsemantic_audio"], {}
    for name, sec, dframes, listed, phones_ok, has_bert in ITEMS:
        n = int(sec * SR)
        sizes[name] = n
        pcm = (rng.randn(n) * 6000).clip(-32768, 32767).astype("<i2")
        with wave.open(os.path.join(root, "5-wav32k", name), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(SR)
            w.writeframes(pcm.tobytes())
        t = frames_of(n) + dframes
        torch.save(torch.randn(1, 768, t, generator=g).half(), os.path.join(root, "4-cnhubert", name + ".pt"))
        n_ph = max(4, int(sec * 9))
        phones = [symbols[int(i)] for i in rng.randint(0, len(symbols), n_ph)]
        if not phones_ok:
            phones[2] = "<no-such-phone>"
        if listed:
            lines.append("\t".join([name, " ".join(phones), "[1, 2]", "text of " + name]))
        if has_bert and phones_ok:
            torch.save(torch.randn(1024, n_ph, generator=g), os.path.join(root, "3-bert", name + ".pt"))
        n_sem = int(sec * 25)
        sem_lines.append(name + "\t" + " ".join(str(int(v)) for v in rng.randint(0, 1024, n_sem)))
    # a file that passes the size filter but cannot be decoded, and a malformed name2text line
    with open(os.path.join(root, "5-wav32k", BROKEN), "wb") as f:
        f.write(bytes(rng.randint(0, 256, 2 * SR, dtype=np.uint8)))
    torch.save(torch.randn(1, 768, 50, generator=g).half(), os.path.join(root, "4-cnhubert", BROKEN + ".pt"))
    lines.append("\t".join([BROKEN, " ".join(symbols[int(i)] for i in rng.randint(0, len(symbols), 7)), "[1]", "broken"]))
    lines.append("\t".join(["a_011.wav", "only", "three"]))
    # semantic rows the s1 filters remove: unknown name, too many tokens, phoneme rate out of range
    sem_lines.append("zz_unknown.wav\t1 2 3")
    sem_lines.append("a_000.wav\t" + " ".join(["5"] * 2600))
    sem_lines.append("a_002.wav\t" + " ".join(["7"] * 400))
    with open(os.path.join(root, "2-name2text.txt"), "w", encoding="utf8") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(root, "6-name2semantic.tsv"), "w", encoding="utf8") as f:
        f.write("\n".join(sem_lines) + "\n")
    return sizes


def sampler_lengths(n=500, seed=5):
    """item lengths (spectrogram frames) for the s2 bucket sampler cases, some outside the boundaries"""
    return [int(v) for v in np.random.RandomState(seed).randint(20, 2100, n)]


def s1_lengths(n=333, seed=6):
    """item durations (seconds) for the s1 bucket sampler cases"""
    return [float(v) for v in np.round(np.random.RandomState(seed).uniform(0.5, 17.0, n), 2)]
