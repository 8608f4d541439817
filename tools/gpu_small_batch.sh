#!/bin/bash
# small-batch check: bit-identity of the launch forms + the CMDM tests + bench at B = 4 / 1 (/ 32)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/small_batch; mkdir -p $O
( timeout 300 tools/kernel_sweep gemm 32,8,4,1 ) > $O/sweep.txt 2>&1
echo "variants that differ: $(grep -c DIFFERS $O/sweep.txt)  bit-identical: $(grep -c 'bit-identical' $O/sweep.txt)"
grep "x9" $O/sweep.txt | grep "B=4 \|B=1 " | grep -v "motion_layer\|128x128" | cut -c1-118
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py -q -x --timeout=600 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for b in 4 1 32; do
  ( timeout 300 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > $O/bench_b$b.json 2>&1
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_b$b.json") if l.startswith("{")][-1])
    print("B=$b:", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("ERR", e)
PY
done
