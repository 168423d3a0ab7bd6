"""Per-kernel register / scratch / occupancy table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage):
what the compiler made of a kernel can be read without a GPU.  usage: python tools/kres.py csrc/file.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/tmp/kres.o"]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-4000:])
    sys.exit(1)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, d in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem)
    dem = re.sub(r"\(.*\)$", "", dem)
    if flt and flt not in dem:
        continue
    print(f"{dem:60s} vgpr {d.get('VGPRs','?'):>4} agpr {d.get('AGPRs','?'):>4} scratch {d.get('ScratchSize [bytes/lane]','?'):>5} "
          f"occ {d.get('Occupancy [waves/SIMD]','?')} sgpr-spill {d.get('SGPRs Spill','?')} vgpr-spill {d.get('VGPRs Spill','?')}")
