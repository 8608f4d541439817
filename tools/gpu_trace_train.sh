#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04g; mkdir -p $O
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -- python $GRAFT_REPO_ROOT/tools/bench_train.py --scene --cpu-steps 0 --steps 6 --warmup 2 > $O/train_full.log 2>&1 )
f=$(find $O/trace_full -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms over 11 steps:", tot/1e6)
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/1e6/11:8.3f} ms/step  calls/step {int(r["Calls"])/11:7.1f}  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:150]}')
PY
find $O/trace_full -name "*kernel_trace.csv" -delete
