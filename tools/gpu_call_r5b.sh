#!/bin/bash
# last check of the round: full -m gpu suite, smoke, the default bench line and the driver's command (no profile collection)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -6 ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 900 python bench.py ) > $O/bench_b32.json 2> $O/bench_b32.err
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_k20.json 2> $O/bench_k20.err
tail -4 $O/pytest_gpu.log; tail -3 $O/smoke.log
python - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ("bench_b32","bench_k20"):
    try:
        d=json.loads([l for l in open(f"{O}/{f}.json") if l.startswith("{")][-1])
        s=d.get("secondary") or {}
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["sample_latency"]["B1"]["p50_ms"], [c["value"] for c in s.get("configs[2]", [])], s.get("strong_scaling_batch_per_gpu", {}).get("B4"))
    except Exception as e:
        print(f, "ERR", e)
PY
