#!/bin/bash
# row-statistic records with 16-byte loads: the prologue of folded launches, then the loop at B = 1 / 4 / 32 (old vs new library, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
( timeout 120 tools/gemm_timeline small ) > $O/gemm_timeline_small.txt 2>&1
grep "^==\|phases" $O/gemm_timeline_small.txt | grep -A1 "lnfold=[12]" | grep "phases" | cut -c1-200
bash tools/gpu_ab_libs.sh 1 4 32
