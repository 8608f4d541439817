#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3k; mkdir -p $O
for t in 0 6 4 0 6; do
  ( timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 --gemm-tile $t ) > $O/bench_t$t.json 2> $O/bench_t$t.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_t$t.json") if l.startswith("{")][-1])
    print("tile $t:", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("ERR", e)
PY
done
