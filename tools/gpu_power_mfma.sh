#!/bin/bash
# board power / shader clock of the bf16 matrix pipe alone at several duty cycles, next to the GEMM kernels alone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/power_mfma; mkdir -p $O
sample() {
  tag=$1; shift
  "$@" > $O/$tag.out 2> $O/$tag.err &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.3
  done > $O/$tag.smi
  wait $pid
  # median of the upper half of the power samples (the run itself, not start-up)
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
    print(f"{sys.argv[2]}: {len(rows)} samples, busy-half median {mid[0]:.0f} W at {mid[1]} MHz, max {rows[-1][0]:.0f} W")
PY
  tail -1 $O/$tag.out | cut -c1-200
}
for cfg in "4 0" "2 0" "1 0" "2 9" "2 18" "2 36" "2 72"; do
  set -- $cfg
  sample mfma_w$1_g$2 tools/probes/mfma_power $1 $2 5
done
sample gemm64 tools/kernel_sweep one 10432 512 512 3 0 70000
sample gemm128 tools/kernel_sweep one 10432 1536 512 5 0 30000
