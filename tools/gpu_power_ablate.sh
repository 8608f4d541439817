#!/bin/bash
# energy attribution of the 64x64 nine-product GEMM: the library built with one K-loop ingredient removed at a time (tools/probes/ablate/, -DAFM_ABLATE)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/power_ablate; mkdir -p $O
sample() {
  tag=$1; shift
  "$@" > $O/$tag.out 2> $O/$tag.err &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.3
  done > $O/$tag.smi
  wait $pid
  python - $O/$tag.smi $tag <<'PY'
import sys,re
rows=[]
for l in open(sys.argv[1]):
    m=re.findall(r'\((\d+)Mhz\)\s+([\d.]+)',l)
    if m: rows.append((float(m[0][1]),int(m[0][0])))
rows.sort()
top=rows[len(rows)//2:]
if top:
    mid=top[len(top)//2]
    print(f"{sys.argv[2]}: busy-half median {mid[0]:.0f} W at {mid[1]} MHz", end="   ")
PY
  tail -1 $O/$tag.out | cut -c1-120
}
sample full tools/kernel_sweep one 10432 512 512 3 0 60000
for v in 1 2 4 8 6 7 16; do
  LD_PRELOAD=$PWD/tools/probes/ablate/libafm_hip_abl$v.so sample abl$v tools/kernel_sweep one 10432 512 512 3 0 60000
done
