#!/bin/bash
# weight-stationary GEMM form: bit-identity tests + timing next to the staged kernels + the loop with it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3o; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout=300 -k "slab or folded" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
( timeout 300 tools/kernel_sweep x9 32,16 ) > $O/sweep.txt 2>&1
tail -5 $O/pytest.log; grep "slab\|64x64 \|128x128" $O/sweep.txt | grep -v "motion_layer\|split-K" | cut -c1-120
for s in 2 1; do
  ( timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 --streams $s ) > $O/bench_s$s.json 2> $O/bench_s$s.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_s$s.json") if l.startswith("{")][-1])
    print("slab default, streams $s:", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("ERR", e)
PY
done
( timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 --gemm-tile 3 ) > $O/bench_t3.json 2> $O/bench_t3.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_t3.json") if l.startswith("{")][-1])
print("64x64 forced on N>=512 (no slab):", d["value"], d["ms_per_step"])
PY
