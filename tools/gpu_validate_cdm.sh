cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -6 ) > gpurun_out/r02y_pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r02y_smoke.log 2>&1
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r02y_bench_k20.json 2>&1
( timeout 500 bash tools/collect_profiles.sh r2 cdm ) > gpurun_out/r02y_collect_cdm.log 2>&1
( timeout 600 python tools/bench_configs.py --only config2 ) > gpurun_out/r02y_config2.jsonl 2> gpurun_out/r02y_config2.err
( timeout 600 python tools/bench_configs.py --only config4 ) > gpurun_out/r02y_config4.jsonl 2> gpurun_out/r02y_config4.err
( timeout 200 python tools/loop_determinism_probe.py 40 ) > gpurun_out/r02y_probe.log 2>&1
tail -4 gpurun_out/r02y_pytest_gpu.log; tail -3 gpurun_out/r02y_smoke.log; tail -1 gpurun_out/r02y_probe.log
grep -h '^{' gpurun_out/r02y_bench_k20.json | cut -c1-300; cut -c1-200 gpurun_out/r02y_config2.jsonl gpurun_out/r02y_config4.jsonl
