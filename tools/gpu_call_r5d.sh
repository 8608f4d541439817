#!/bin/bash
# the driver's 20-step region: does the length of the warm-up call matter?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
B="python bench.py --gpus 1 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0"
for rep in 1 2; do for kw in "20 5" "20 20" "20 40" "200 20"; do set -- $kw
  ( timeout 200 $B --steps $1 --warmup $2 ) > $O/b.json 2> $O/b.err
  python - "K=$1 W=$2 rep $rep" $O/b.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done; done | tee $O/warmup_len.txt
