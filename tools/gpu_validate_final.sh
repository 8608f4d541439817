#!/bin/bash
# trimmed end-of-round check: full -m gpu suite, smoke, the default bench line (with the informational passes) and the driver's command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -6 ) > gpurun_out/r02x_pytest_gpu.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r02x_smoke.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/r02x_bench_b32.json 2> gpurun_out/r02x_bench_b32.err
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r02x_bench_k20.json 2>&1
tail -4 gpurun_out/r02x_pytest_gpu.log; tail -2 gpurun_out/r02x_smoke.log
python - <<'PY'
import json
for f in ("r02x_bench_b32","r02x_bench_k20"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], json.dumps(d.get("alt_gemm_modes"))[:700])
    except Exception as e:
        print(f, "ERR", e)
PY
