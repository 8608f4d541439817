#!/bin/bash
# lockstep experiment: half of the workgroups of a launch start 6 / 12 us late (tools builds only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05c; mkdir -p $O
for bin in gemm_timeline gemm_timeline_desync6 gemm_timeline_desync12 gemm_timeline; do
  echo "#### $bin"
  timeout 60 tools/$bin 10432 512 512 0 9 0 1
  timeout 60 tools/$bin 5216 512 512 0 9 0 1
  timeout 60 tools/$bin 10432 1536 512 0 9 0 2
  timeout 60 tools/$bin 5216 1536 512 0 9 0 2
done > $O/desync.txt 2>&1
grep "^####\|^==\|phases\|started" $O/desync.txt | cut -c1-250
