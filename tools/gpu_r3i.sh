#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -12 ) > $O/pytest_gpu.log 2>&1
for v in fold nofold; do
  f=""; [ $v = nofold ] && f="--no-ln-fold"
  ( timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 $f ) > $O/bench_$v.json 2> $O/bench_$v.err
  ( timeout 300 python bench.py --batch 4 --steps 200 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 $f ) > $O/bench4_$v.json 2> $O/bench4_$v.err
done
( timeout 600 python tools/cdm_ab.py 100 ) > $O/cdm_ab.jsonl 2> $O/cdm_ab.err
( timeout 600 python tools/bench_configs.py --only config3 ) > $O/config3.jsonl 2> $O/config3.err
tail -6 $O/pytest_gpu.log
for v in bench_fold bench_nofold bench4_fold bench4_nofold; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$v.json") if l.startswith("{")][-1])
    print("$v:", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("$v ERR", e)
PY
done
cut -c1-260 $O/cdm_ab.jsonl | head -3; cut -c1-400 $O/config3.jsonl
