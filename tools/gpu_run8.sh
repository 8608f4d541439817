#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 tools/kernel_sweep gemm ${1:-32} ) > gpurun_out/r02_gemm_sweep_half.txt 2>&1
echo "sweep: identical $(grep -c bit-identical gpurun_out/r02_gemm_sweep_half.txt) differs $(grep -c DIFFERS gpurun_out/r02_gemm_sweep_half.txt) rc-lines $(grep -c 'rc=' gpurun_out/r02_gemm_sweep_half.txt)"
grep "x9" gpurun_out/r02_gemm_sweep_half.txt | grep -v "split-K\|BK32" | cut -c1-108
