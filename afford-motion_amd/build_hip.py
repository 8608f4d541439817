"""Build libafm_hip.so (gfx950) in-tree: hipcc cross-compiles without a GPU.

    python afford-motion_amd/build_hip.py [--force]

Objects go to afford-motion_amd/build/, the library to afford-motion_amd/afm/libafm_hip.so
(git-ignored; it travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "afm", "libafm_hip.so")
OBJ = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wall", "-Wno-unused-function", "-Werror=pass-failed",          # a failed `#pragma unroll` demotes register arrays to scratch
         "-Rpass-analysis=kernel-resource-usage"]
MAX_SCRATCH_BYTES = 32       # per lane; anything larger means an accumulator array left the register file


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([hipcc] + FLAGS + ["-c", s, "-o", o], capture_output=True, text=True)
        log = r.stdout + r.stderr
        rc = r.returncode
        keep = []
        name = "?"
        skip = 0
        for line in log.splitlines():
            if skip and ("|" in line[:8] or not line.strip()):      # source excerpt printed under a remark
                continue
            skip = 0
            if "remark:" in line:
                skip = 1
                if "Function Name:" in line:
                    name = line.split("Function Name:")[1].split("[")[0].strip()
                if "ScratchSize [bytes/lane]:" in line:
                    sz = int(line.split("ScratchSize [bytes/lane]:")[1].split("[")[0])
                    if sz > MAX_SCRATCH_BYTES:
                        keep.append(f"error: kernel {name} uses {sz} B/lane of scratch (register array spilled)")
                        rc = rc or 1
                continue
            keep.append(line)
        if rc != 0 and os.path.exists(o):
            os.remove(o)           # a rejected object must not satisfy the next (incremental) build
        return s, rc, "\n".join(keep)

    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for s, rc, log in ex.map(cc, jobs):
            if verbose and log.strip():
                print(log, file=sys.stderr)
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {s}")
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(OUT, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(f"built {OUT} ({os.path.getsize(OUT) / 1024:.0f} KiB) from {len(srcs)} sources, {len(jobs)} recompiled")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
