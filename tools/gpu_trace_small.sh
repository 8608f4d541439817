#!/bin/bash
# rocprofv3 kernel trace of the small-batch loops (B = 4 and B = 1): per-kernel launches and durations of a step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/trace_small; mkdir -p $O
for b in 4 1; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_b$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 > $O/bench_b$b.log 2>&1 )
  f=$(find $O/trace_b$b -name "*kernel_stats.csv" | head -1)
  echo "== B=$b $f"; head -30 "$f" | cut -c1-200
  find $O/trace_b$b -name "*kernel_trace.csv" -delete
done
