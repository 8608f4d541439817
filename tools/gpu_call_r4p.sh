#!/bin/bash
# epilogue with hoisted column vectors + residual one trip ahead (new) against the round-3 form (old: -DAFM_PK_PROBE=256), same call
O=gpurun_out/r04p; mkdir -p $O
for bin in gemm_timeline_old gemm_timeline gemm_timeline_old gemm_timeline; do
  echo "#### $bin"
  for M in 10432 5216; do
    timeout 60 tools/$bin $M 512 512 0 9 0 1
    timeout 60 tools/$bin $M 512 1024 0 9 0 1
    timeout 60 tools/$bin $M 1024 512 0 9 0 2
    timeout 60 tools/$bin $M 1536 512 0 9 0 2
  done
done > $O/epi.txt 2>&1
grep -v "^   resident\|distinct CUs" $O/epi.txt | cut -c1-300
