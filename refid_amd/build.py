"""Build librefid_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

`python -m refid_amd.build` or `refid_amd.build.build()`; incremental (mtime based).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librefid_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-pass-failed"]
# (the tiles that were measured and lost -- the persistent one-wave-per-SIMD Winograd tile, the wide Winograd x six tile, the
#  F(3x3,4x4) weight gradient -- lived in csrc/experimental/ behind REFID_EXPERIMENTAL_TILES=1 until round 6; their numbers
#  are in DESIGN.md section 7, their sources in the history)


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(verbose=False, force=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    flags = list(FLAGS)
    # The flag set the objects and the library were built with is recorded AFTER a successful link (never before: a failed
    # compile must not leave a stamp that says "already built this way"); a different or missing record rebuilds everything.
    stamp = os.path.join(CSRC, ".buildflags")
    want = " ".join(flags)
    try:
        have = open(stamp).read()
    except OSError:
        have = None if os.path.exists(LIB) else want       # a tree that was never built: nothing stale to distrust
    if have != want:
        force = True
        if os.path.exists(stamp):
            os.remove(stamp)                               # until the link below succeeds the tree counts as unknown
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "refid_hip.h")]
    objs = []
    jobs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _newer([s] + hdrs, o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or force or _newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if not os.path.exists(stamp):
        with open(stamp, "w") as f:
            f.write(want)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
