#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py -m gpu -q -x --timeout=600 --deselect tests/test_gpu_cmdm.py::test_two_stream_loop_soak 2>&1 | tail -25 ) > $O/pytest.log 2>&1
for v in fold nofold; do
  f=""; [ $v = nofold ] && f="--no-ln-fold"
  ( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 $f ) > $O/bench_$v.json 2> $O/bench_$v.err
  ( timeout 300 python bench.py --batch 4 --steps 200 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 $f ) > $O/bench4_$v.json 2> $O/bench4_$v.err
done
tail -12 $O/pytest.log
for v in bench_fold bench_nofold bench4_fold bench4_nofold; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$v.json") if l.startswith("{")][-1])
    print("$v:", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("$v ERR", e)
PY
done
