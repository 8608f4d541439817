#!/bin/bash
# end-of-round validation and artifacts: full -m gpu suite, smoke, bench (default, driver's command, B = 4), profiles (CMDM + CDM), secondary configs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -6 ) > gpurun_out/r02z_pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r02z_smoke.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r02z_bench_b32.json 2> gpurun_out/r02z_bench_b32.err
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r02z_bench_k20.json 2>&1
( timeout 300 python bench.py --batch 4 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02z_bench_b4.json 2>&1
( timeout 600 bash tools/collect_profiles.sh r2 ) > gpurun_out/r02z_collect.log 2>&1
( timeout 500 bash tools/collect_profiles.sh r2 cdm ) > gpurun_out/r02z_collect_cdm.log 2>&1
( timeout 900 python tools/bench_configs.py ) > gpurun_out/r02z_configs.jsonl 2> gpurun_out/r02z_configs.err
tail -4 gpurun_out/r02z_pytest_gpu.log; tail -3 gpurun_out/r02z_smoke.log
python - <<'PY'
import json
for f in ("r02z_bench_b32","r02z_bench_k20","r02z_bench_b4"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], (d["roofline"]["traffic"] or {}).get("bytes_per_launch"), d.get("sample_latency"))
    except Exception as e:
        print(f, "ERR", e)
PY
cut -c1-400 gpurun_out/r02z_configs.jsonl
