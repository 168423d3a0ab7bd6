"""Instruction mix of one kernel in a hipcc -S listing, whole function and per basic block (the persistent loop bodies are
the big blocks).  usage: python tools/isa_mix.py file.s <substring of the mangled name> [min block size]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
minblk = int(sys.argv[3]) if len(sys.argv) > 3 else 60
start = next(i for i, l in enumerate(s) if l.startswith("_Z") and pat in l and l.rstrip().split(":")[0].endswith(pat.split()[-1]) or
             (l.startswith("_Z") and pat in l and ":" in l))
end = next(i for i in range(start + 1, len(s)) if s[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in s[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", "//")):
        continue
    if t.startswith(".") and not t.endswith(":"):
        continue
    if t.endswith(":"):
        blocks.append((name, cur))
        cur, name = [], t[:-1]
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
tot = Counter(x for _, b in blocks for x in b)
print("function", s[start].split(":")[0], "instructions", sum(tot.values()))
for name, b in blocks:
    if len(b) < minblk:
        continue
    c = Counter(b)
    grp = Counter()
    for k, v in c.items():
        g = ("mfma" if "mfma" in k else "ds_read_tr" if "ds_read_b64_tr" in k else "ds_read" if k.startswith("ds_read") else
             "ds_write" if k.startswith("ds_write") else "global_load" if k.startswith(("global_load", "buffer_load")) else
             "global_store" if k.startswith(("global_store", "buffer_store")) else "s_waitcnt" if k == "s_waitcnt" else
             "s_nop" if k == "s_nop" else "v_accvgpr" if "accvgpr" in k else "scratch" if k.startswith("scratch") else
             "salu" if k.startswith("s_") else "valu" if k.startswith("v_") else "other")
        grp[g] += v
    print(f"block {name:16s} n={len(b):5d}  " + "  ".join(f"{k}={v}" for k, v in grp.most_common()))
