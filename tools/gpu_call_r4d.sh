#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py tests/test_gpu_c_abi.py -m gpu -q -x --timeout=600 2>&1 | tail -25 ) > $O/pytest.log 2>&1
( timeout 400 python tools/small_batch_probe.py 200 ) > $O/small_batch.jsonl 2> $O/small_batch.err
( timeout 300 python bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > $O/bench_k100.json 2> $O/bench.err
tail -12 $O/pytest.log; cat $O/small_batch.jsonl; tail -2 $O/small_batch.err
python - $O <<'PY'
import json,sys
O=sys.argv[1]
d=json.loads([l for l in open(f"{O}/bench_k100.json") if l.startswith("{")][-1])
print("bench", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"], d["roofline"]["all_kernels_tflops"])
PY
