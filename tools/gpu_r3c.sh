#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=600 -k "mha or encoder" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
( timeout 120 tools/mha_timeline ) > $O/mha_timeline.txt 2>&1
( timeout 300 tools/kernel_sweep mha 32,16,4,1 ) > $O/mha_sweep.txt 2>&1
for g in 0 4; do
  ( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 --attn-group $g ) > $O/bench_g$g.json 2> $O/bench_g$g.err
done
tail -3 $O/pytest.log; cat $O/mha_timeline.txt; grep -E "B=32|B=4 " $O/mha_sweep.txt | grep "no-mask"
for g in 0 4; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_g$g.json") if l.startswith("{")][-1])
    print("group $g:", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("group $g ERR", e)
PY
done
