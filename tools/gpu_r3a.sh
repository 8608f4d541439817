#!/bin/bash
# round 3, call A: new attention kernel (parity + groupings + sweep) and the step rate per attention group size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3a; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py -m gpu -q -x --timeout=600 2>&1 | tail -25 ) > $O/pytest.log 2>&1
( timeout 300 tools/kernel_sweep mha 32,16,4,1 ) > $O/mha_sweep.txt 2>&1
for g in 4 6 12; do
  ( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 --attn-group $g ) > $O/bench_g$g.json 2> $O/bench_g$g.err
done
tail -5 $O/pytest.log; grep "B=32" $O/mha_sweep.txt | head -20
for g in 4 6 12; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_g$g.json") if l.startswith("{")][-1])
    print("group $g:", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("group $g ERR", e)
PY
done
