#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 tools/kernel_sweep gemm 32 ) > gpurun_out/r02_gemm_sweep_bk32.txt 2>&1
echo "sweep: identical $(grep -c bit-identical gpurun_out/r02_gemm_sweep_bk32.txt) differs $(grep -c DIFFERS gpurun_out/r02_gemm_sweep_bk32.txt) rc-lines $(grep -c 'rc=' gpurun_out/r02_gemm_sweep_bk32.txt)"
grep "B=32" gpurun_out/r02_gemm_sweep_bk32.txt | grep "x9 auto\|x9 64x64\|x9 128\|BK32" | cut -c1-120
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py -q -x --timeout=600 2>&1 | grep -v "^$" | tail -15 ) > gpurun_out/r02g_pytest.log 2>&1
tail -6 gpurun_out/r02g_pytest.log
