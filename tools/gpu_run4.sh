#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 200 tools/kernel_sweep gemm 32,8,4,1 ) > gpurun_out/r02_gemm_sweep10.txt 2>&1
grep -c bit-identical gpurun_out/r02_gemm_sweep10.txt; grep -c DIFFERS gpurun_out/r02_gemm_sweep10.txt
( timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py -q -x --timeout=600 2>&1 | tail -4 ) > gpurun_out/r02d_pytest.log 2>&1
tail -3 gpurun_out/r02d_pytest.log
for b in 32 8 4 1; do
  ( timeout 200 python bench.py --batch $b --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02d_bench_b$b.json 2> gpurun_out/r02d_bench_b$b.err
  python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r02d_bench_b$b.json') if l.startswith('{')][-1])
print('B=$b', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
done
