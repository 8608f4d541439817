#!/bin/bash
# HIP_FORCE_DEV_KERNARG: where the 584-byte afm_linear_args kernarg block of a launch lives (host-coherent memory vs device memory)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04v; mkdir -p $O
for v in 0 1; do
  echo "#### HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v timeout 120 tools/probes/launch_shape 2>&1 | head -12
done > $O/launch_shape_kernarg.txt 2>&1
cat $O/launch_shape_kernarg.txt | cut -c1-170
B="python bench.py --steps 300 --warmup 30 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0"
for rep in 1 2; do for v in 0 1; do for b in 4 32 1; do
  ( HIP_FORCE_DEV_KERNARG=$v timeout 200 $B --batch $b ) > $O/b.json 2> $O/b.err
  python - "kernarg=$v B=$b rep=$rep" $O/b.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done; done; done | tee $O/bench_kernarg.txt
