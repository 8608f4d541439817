#!/bin/bash
# the driver's command, several processes in one call (process-to-process spread), and the same with a 200-step region
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
B="python bench.py --gpus 1 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0"
for i in 1 2 3 4 5; do
  for kw in "20 5" "200 20"; do set -- $kw
    ( timeout 200 $B --steps $1 --warmup $2 ) > $O/b.json 2> $O/b.err
    python - "K=$1 W=$2 run $i" $O/b.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["preflight"]["priming_loop_ms"], d["preflight"]["ms_between_priming_and_warmup"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
  done
done | tee $O/driver_cmd_spread.txt
