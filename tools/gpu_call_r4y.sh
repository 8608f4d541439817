#!/bin/bash
# attention workgroup shape inside the two-stream loop (bit-neutral knob)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04y; mkdir -p $O
B="python bench.py --gpus 1 --steps 300 --warmup 30 --no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0"
for cfg in "" "--attn-group 8" "--attn-group 6" "--attn-group 4" "--attn-group 104" "--attn-group 102" "" "--attn-group 6"; do
  ( timeout 200 $B $cfg ) > $O/b.json 2> $O/b.err
  python - "attn[$cfg]" $O/b.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["all_kernels_ms_per_step"].get("mha_fwd_split_kernel"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[2].replace('.json','.err')).read()[-300:])
PY
done | tee $O/attn_group.txt
