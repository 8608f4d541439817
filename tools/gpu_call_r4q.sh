#!/bin/bash
# (1) A-path ablation: the floor of out_proj with its A operand already in LDS; (2) sub-batch stream count and forced 64x64 in_proj tiles in the loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04q; mkdir -p $O
for bin in gemm_timeline gemm_timeline_abl64 gemm_timeline_abl96; do
  echo "#### $bin"
  for M in 10432 5216; do timeout 60 tools/$bin $M 512 512 0 9 0 1; done
done > $O/ablate_a.txt 2>&1
grep "^==\|phases" $O/ablate_a.txt | cut -c1-260
B="python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0"
for cfg in "--streams 2" "--streams 1" "--streams 3" "--streams 4" "--streams 2 --gemm-tile 3" "--streams 3 --gemm-tile 3" "--streams 2"; do
  ( timeout 200 $B $cfg ) > $O/b.json 2> $O/b.err
  python - "$cfg" $O/b.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done | tee $O/streams_tiles.txt
