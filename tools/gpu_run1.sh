#!/bin/bash
# round-2 GPU call #1: full -m gpu suite, kernel sweep (tile shapes / attention groupings / small batches), bench at B = 32 / 16 / 8 / 4
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -60 ) > gpurun_out/r02_pytest_gpu.log 2>&1
( time timeout 300 tools/kernel_sweep all 32,16,8,4,1 ) > gpurun_out/r02_kernel_sweep.txt 2>&1
( time timeout 600 python bench.py ) > gpurun_out/r02_bench_b32.json 2> gpurun_out/r02_bench_b32.err
for b in 16 8 4 1; do
  ( timeout 200 python bench.py --batch $b --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02_bench_b$b.json 2> gpurun_out/r02_bench_b$b.err
done
( timeout 200 python bench.py --batch 4 --streams 1 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02_bench_b4_s1.json 2>&1
( timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02_bench_b32_k20.json 2>&1
tail -5 gpurun_out/r02_pytest_gpu.log
head -c 1500 gpurun_out/r02_bench_b32.json
