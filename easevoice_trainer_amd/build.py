"""Builds libevt_hip.so (all csrc/*.hip, gfx950 only; bfloat16 as the 16-bit type) and libevt_hip_f16.so (the same sources
with -DEVT_HALF_F16: IEEE half, the reference's fp16_run mode) in-tree with hipcc.

Incremental by CONTENT: an object is rebuilt when the hash of its source, the headers and the flags differs from the one
recorded next to it (mtimes say nothing after a checkout).  hipcc cross-compiles without a GPU, so this runs in the build
container; the .so travels to the GPU box with the tree.  The hash of all sources is compiled into evt_version(), and
hip/lib.py compares it with the sources it finds at load time: a stale library is an error, not a silent old kernel.
Only the entry points declared in include/evt.h are exported (-fvisibility=hidden + the header's visibility pragma).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libevt_hip.so")
LIB_F16 = os.path.join(HERE, "libevt_hip_f16.so")
HALF_BUILDS = {"bf16": (OBJ, LIB, []), "f16": (os.path.join(HERE, "csrc", "_obj_f16"), LIB_F16, ["-DEVT_HALF_F16"])}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]
EXTRA_FLAGS = {}   # per-file flags (none at present)
VERSION_SRC = "elementwise.hip"   # defines evt_version(): compiled with -DEVT_SRC_HASH=<hash of all sources>


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdrs.append(os.path.join(HERE, "..", "include", "evt.h"))
    return hdrs


def source_hash() -> str:
    """12 hex digits over every csrc/*.hip, csrc/*.h and include/evt.h: what evt_version() of a fresh build reports"""
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    return _sha(srcs + _headers())[:12]


def build_lib(force: bool = False, verbose: bool = False, half: str = "bf16") -> str:
    """half: "bf16" -> libevt_hip.so, "f16" -> libevt_hip_f16.so (same sources, -DEVT_HALF_F16)"""
    OBJ, LIB, half_flags = HALF_BUILDS[half]
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = _headers()
    shash = source_hash()
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        flags = [*FLAGS, *half_flags, *EXTRA_FLAGS.get(s, [])]
        if s == VERSION_SRC:
            flags.append(f'-DEVT_SRC_HASH="{shash}"')
        stamp = _sha([src] + hdrs, " ".join(flags))
        rec = obj + ".hash"
        old = open(rec).read().strip() if os.path.exists(rec) else ""
        if force or not os.path.exists(obj) or old != stamp:
            jobs.append((src, obj, flags, stamp))

    def cc(job):
        src, obj, flags, stamp = job
        cmd = [HIPCC, *flags, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        with open(obj + ".hash", "w") as f:
            f.write(stamp)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


def build_all(force: bool = False, verbose: bool = False):
    return [build_lib(force, verbose, half) for half in ("bf16", "f16")]


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
