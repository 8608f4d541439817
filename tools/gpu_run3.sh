#!/bin/bash
# round-2 GPU call: x9 BK=32 check, secondary configs (incl. configs[0] CPU timing), rocprof kernel stats + PMC passes for the CMDM headline path and
# the CDM Perceiver (BASELINE configs[2]), the full bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 150 tools/kernel_sweep gemm 4,32 ) > gpurun_out/r02_gemm_sweep9.txt 2>&1
( timeout 600 python tools/bench_configs.py ) > gpurun_out/r02_configs.jsonl 2> gpurun_out/r02_configs.err
( timeout 500 bash tools/collect_profiles.sh r2 ) > gpurun_out/r02_collect.log 2>&1
( timeout 400 bash tools/collect_profiles.sh r2 cdm ) > gpurun_out/r02_collect_cdm.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/r02c_bench_b32.json 2> gpurun_out/r02c_bench_b32.err
tail -2 gpurun_out/r02_configs.err; head -c 400 gpurun_out/r02c_bench_b32.json
