#!/bin/bash
# round-2 GPU call #2: GEMM per-workgroup timelines, attention sweep after the loader fix, bench with the new defaults (x9 on every eligible GEMM,
# round-aware tile choice, 4-wave attention groups), parity prints, full -m gpu suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 120 tools/gemm_timeline ) > gpurun_out/r02_gemm_timeline.txt 2>&1
( timeout 200 tools/kernel_sweep mha 32,16,4,1 ) > gpurun_out/r02_mha_sweep2.txt 2>&1
( timeout 600 python bench.py --no-cpu-baseline --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02b_bench_b32.json 2> gpurun_out/r02b_bench_b32.err
( timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02b_bench_b32_k20.json 2>&1
for b in 16 8 4 1; do
  ( timeout 200 python bench.py --batch $b --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02b_bench_b$b.json 2> gpurun_out/r02b_bench_b$b.err
done
( timeout 200 python bench.py --streams 3 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02b_bench_b32_s3.json 2>&1
( timeout 900 python -m pytest tests -m gpu -q -x -s --timeout=900 -k "drift or full_size or config4 or recycled or default_seed or small_batch" 2>&1 | grep -E "drift|parity|cache|passed|failed|Error" ) > gpurun_out/r02_parity.txt 2>&1
( time timeout 900 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -15 ) > gpurun_out/r02b_pytest_gpu.log 2>&1
tail -3 gpurun_out/r02b_pytest_gpu.log; head -c 600 gpurun_out/r02b_bench_b32.json
