"""Build libafm_hip.so (gfx950) in-tree: hipcc cross-compiles without a GPU.

    python afford-motion_amd/build_hip.py [--force]

Objects go to afford-motion_amd/build/, the library to afford-motion_amd/afm/libafm_hip.so
(git-ignored; it travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import re
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
# Round 4 narrowed the defect to ONE instruction property (tools/probes/pk_repro.hip, profiles/r04_packed_f32_repro.md): a v_pk_{mul,add,fma}_f32
# whose op_sel routes a source's HIGH half into the LOW result computes that low result wrong in lanes 48..63 whenever another wave on the
# chip interleaves MFMAs with v_accvgpr_read / v_accvgpr_write (what every GEMM kernel of this library does) - reproduced as a single
# instruction in a loop; op_sel_hi variants and op_sel-free packed ops are exact.  hipcc picks the operand selection, so the WHOLE class
# stays fenced:
# Packed-f32 VALU instructions (v_pk_mul / v_pk_fma / v_pk_add _f32) are NOT selected: kernels using them gave wrong results in lanes
# 48..63 of single waves whenever a second stream had kernels in flight on the chip - round 2's lat_decfold (hi half of an SGPR-pair
# operand read as 0) and round 3's statistics-carrying GEMM epilogue (low halves of VGPR pairs wrong: rows 3 mod 4 x even columns), the
# latter reproduced in isolation by tools/probes/ln_fold_streams.py; with the target feature off both are exact in every run
# (profiles/r03_packed_f32_defect.md).  AFM_PACKED_FP32=1 in the BUILD's environment re-enables them (to run the reproducer).
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
MAX_SCRATCH_BYTES = 96       # per lane; anything larger means an accumulator array left the register file (the 168-register attention
                             # variants park scalars and pointers - 44 B in round 3, 84 B with round 4's two key segments - in scratch: spill
                             # stores in the prologue, single-dword reloads per pass / key block; checked in the ISA and against the timing)
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# ---- ISA scan (profiles/r02_decfold_nondeterminism.md, profiles/r03_packed_f32_defect.md).  Two kernels have now lost single values in
# lanes 48..63 of single waves whenever a second stream had kernels in flight, both inside packed-f32 VALU instructions; the library is
# therefore compiled without them (NO_PACKED_F32) and every code object is scanned: ANY v_pk_mul / v_pk_fma / v_pk_add _f32 rejects it.
_PK = re.compile(r"^\s*(v_pk_(?:mul|fma|add)_f32)\s+(.*?)(?://.*)?$")
_INS = re.compile(r"^\s*([a-z_0-9]+)\s+(.*?)(?://.*)?$")
_MOD = re.compile(r"\s+(?:op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b.*$")
ISA_SCAN_WINDOW = 8


def _sregs(tok):
    m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_disassembly(text, window=ISA_SCAN_WINDOW):
    """-> [(function, packed instruction, note)] for every fenced instruction of an llvm-objdump -d listing: ANY v_pk_mul / v_pk_fma /
    v_pk_add _f32 (round 3; the library is built without them), the note naming round 2's narrower shape when it is present (an
    `s_mov_b32 s, s` re-pack of the SGPR pair whose high half the instruction reads, within `window` instructions)."""
    hits, fn, body = [], "?", []

    def flush():
        for i, (op, rest) in enumerate(body):
            if not _PK.match(op + " " + rest):
                continue
            note = "(packed-f32 VALU instruction)"
            sel = {k: None for k in ("op_sel", "op_sel_hi")}
            for k in sel:
                mm = re.search(k + r":\[([0-9,]+)\]", rest)
                if mm:
                    sel[k] = [int(x) for x in mm.group(1).split(",")]
            if sel["op_sel"] and any(sel["op_sel"]):
                # round 4 (profiles/r04_packed_f32_repro.md): THE failing form - a source's HIGH half feeding the LOW result - reproduced as one
                # instruction in a loop: wrong low results in lanes 48..63 while a co-resident wave interleaves MFMAs with v_accvgpr moves
                note = "(packed-f32 with op_sel: a source's high half feeds the low result - the form that fails in lanes 48..63, round 4)"
            srcs = re.split(r",\s*", _MOD.sub("", rest.strip()))[1:]
            hi_pairs = []
            for si, tok in enumerate(srcs):
                if not re.match(r"^s\[\d+:\d+\]$", tok):
                    continue
                lo_sel = sel["op_sel"][si] if sel["op_sel"] else 0            # which half feeds the low lane
                hi_sel = sel["op_sel_hi"][si] if sel["op_sel_hi"] else 1      # which half feeds the high lane (default: the high half)
                if lo_sel or hi_sel:
                    hi_pairs.append(max(_sregs(tok)))                         # the pair's high register is read
            for j in range(max(0, i - window), min(len(body), i + window + 1)):
                o2, r2 = body[j]
                if o2 != "s_mov_b32" or not hi_pairs:
                    continue
                ops2 = re.split(r",\s*", r2.strip())
                if len(ops2) == 2 and _sregs(ops2[0]) & set(hi_pairs) and _sregs(ops2[1]):     # SGPR -> SGPR copy into a read high half
                    note = f"next to `{o2} {r2.strip()}` (round 2's lat_decfold shape)"
            hits.append((fn, op + " " + rest.strip(), note))
    for line in text.splitlines():
        m0 = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m0:
            flush()
            body, fn = [], m0.group(1)
            continue
        m = _INS.match(line)
        if m:
            body.append(m.groups())
    flush()
    return hits


def scan_object(obj):
    """Disassemble the gfx950 code object embedded in a hipcc -c object and scan it; -> list of hits (empty = clean)."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="afm_isa_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", os.path.basename(local)], cwd=tmp, capture_output=True, text=True)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not cos:                                      # host-only translation unit (profile.hip)
            return []
        hits = []
        for co in cos:
            r = subprocess.run([OBJDUMP, "-d", co], cwd=tmp, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"llvm-objdump failed on {obj}: {r.stderr[:400]}")
            hits += scan_disassembly(r.stdout)
        return hits
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "hipcc")
    packed = os.environ.get("AFM_PACKED_FP32") == "1"          # build-time switch of THIS script (the library reads no environment)
    flags = FLAGS + ([] if packed else NO_PACKED_F32)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([hipcc] + flags + ["-c", s, "-o", o], capture_output=True, text=True)
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
        if rc == 0 and not packed:
            for fn, pk, mv in scan_object(o):
                keep.append(f"error: fenced instruction in {fn}: `{pk}` {mv} (see ISA scan in build_hip.py)")
                rc = 1
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
