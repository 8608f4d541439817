#!/bin/bash
# is the sampling loop power-limited?  board power / shader clock sampled while the loop runs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/power_loop; mkdir -p $O
rocm-smi --showmaxpower 2>&1 | grep -i "power" | head -3 > $O/cap.txt
sample() {
  tag=$1; shift
  "$@" > $O/$tag.out 2> $O/$tag.err &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.4
  done > $O/$tag.smi
  wait $pid
}
sample cmdm timeout 200 python bench.py --steps 1000 --warmup 10 --no-cpu-baseline --no-alt-gemm --latency-runs 6 --latency-runs-b1 0
cat $O/cap.txt; sort -t' ' -k3 -n -r $O/cmdm.smi | head -12; echo; wc -l $O/cmdm.smi; tail -1 $O/cmdm.out | cut -c1-200
