#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py tests/test_gpu_cdm.py -m gpu -q -x --timeout=600 2>&1 | tail -12 ) > $O/pytest.log 2>&1
( timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > $O/bench_k200.json 2> $O/bench.err
( timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > $O/bench_k20.json 2>> $O/bench.err
( timeout 400 python tools/small_batch_probe.py 200 ) > $O/small_batch.jsonl 2> $O/small_batch.err
tail -8 $O/pytest.log
python - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ("bench_k200","bench_k20"):
    d=json.loads([l for l in open(f"{O}/{f}.json") if l.startswith("{")][-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"], d["roofline"]["all_kernels_tflops"])
PY
cut -c1-110 $O/small_batch.jsonl
