#!/bin/bash
# the kernel-trace --stats pass of tools/collect_profiles.sh alone (counters already collected into gpurun_out/prof_$R by an earlier call of the round)
R=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT; rm -rf $OUT/stats
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --streams 1 --steps 100 --warmup 10 --latency-runs 0 --latency-runs-b1 0 --no-cpu-baseline --no-alt-gemm --no-secondary"
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_rocprof.json 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
tail -1 $OUT/bench_under_rocprof.json | cut -c1-400
