#!/usr/bin/env python
"""Per-entry-point time of a run, by C-ABI call and argument shape: every `afm_*` function of the loaded library is wrapped with a HIP
event pair (synchronising - this serialises the run, so the numbers attribute, they do not add up to the unprofiled step time), keyed by
the function name plus the integer shape fields of its arguments (M / N / K of the GEMM structs, small integer scalars).

    python tools/abi_call_profile.py [--top 40] [--skip-calls N] -- tools/bench_train.py --scene --steps 2 --warmup 1 --cpu-steps 0
"""
import argparse
import collections
import ctypes as C
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from afm import ffi  # noqa: E402

def shape_key(args):
    out = []
    for a in args:
        obj = getattr(a, "_obj", None)              # byref(struct)
        if obj is not None and isinstance(obj, C.Structure):
            ints = [(f[0], getattr(obj, f[0])) for f in obj._fields_ if f[1] in (C.c_int32, C.c_int64, C.c_int)]
            ptrs = [f[0] for f in obj._fields_ if f[1] not in (C.c_int32, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_uint32) and getattr(obj, f[0])]
            out.append("{" + ",".join(f"{n}={v}" for n, v in ints if 0 < v < (1 << 24)) + " | " + ",".join(ptrs) + "}")
        elif isinstance(a, bool):
            out.append(str(int(a)))
        elif isinstance(a, int) and 0 <= a < (1 << 24):
            out.append(str(a))
        elif isinstance(a, float):
            out.append(f"{a:g}")
    return " ".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--by-name", action="store_true", help="aggregate over shapes")
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    rest = a.rest[1:] if a.rest and a.rest[0] == "--" else a.rest
    lib = ffi.load()
    stats = collections.defaultdict(lambda: [0, 0.0])
    state = {"on": True}

    def wrap(name, fn):
        def inner(*args):
            if not state["on"] or not torch.cuda.is_available():
                return fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(torch.cuda.current_stream())
            rc = fn(*args)
            e1.record(torch.cuda.current_stream())
            torch.cuda.synchronize()
            st = stats[(name, "" if a.by_name else shape_key(args))]
            st[0] += 1
            st[1] += e0.elapsed_time(e1)
            return rc
        return inner

    for n in ffi.EXPORTS:
        fn = getattr(lib, n)
        if "workspace_bytes" in n or "profile" in n or "version" in n or "last_error" in n or ffi.EXPORTS[n][0] is not C.c_int:
            continue
        setattr(lib, n, wrap(n, fn))
    sys.argv = rest
    try:
        runpy.run_path(rest[0], run_name="__main__")
    finally:
        state["on"] = False
        rows = sorted(stats.items(), key=lambda kv: -kv[1][1])
        total = sum(v[1] for v in stats.values())
        print(f"# {len(rows)} (entry, shape) keys, {total:.2f} ms inside profiled calls", file=sys.stderr)
        for (name, key), (cnt, ms) in rows[:a.top]:
            print(f"{ms:9.3f} ms  {cnt:5d} calls  {ms / cnt * 1e3:9.1f} us/call  {name}  {key}", file=sys.stderr)


if __name__ == "__main__":
    main()
