#!/bin/bash
# A/B of two builds of the library in ONE call: tools/ab_libs/libafm_old.so vs libafm_new.so, bench at the batch sizes given (default 1 4), alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/ab_libs; mkdir -p $O
L=afford-motion_amd/afm/libafm_hip.so
cp $L $O/orig.so
for rep in 1 2; do
  for v in old new; do
    cp tools/ab_libs/libafm_$v.so $L
    for b in ${@:-1 4}; do
      ( timeout 300 python bench.py --batch $b --steps 300 --warmup 30 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > $O/${v}_b${b}_$rep.json 2>&1
      python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/${v}_b${b}_$rep.json") if l.startswith("{")][-1])
    print("$v rep $rep B=$b:", d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"])
except Exception as e: print("ERR", e)
PY
    done
  done
done
cp $O/orig.so $L; rm $O/orig.so
