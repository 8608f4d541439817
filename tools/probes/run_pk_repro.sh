#!/bin/bash
# one GPU call (tools/gpu_call.sh pk_repro): the stand-alone reproducer.  Default: round 5's controls (mode opsel4: padding around the co-runner's
# v_accvgpr moves; victim / co-runner placement by CU masks) + opsel3 as the reference row.  `full`: also the library sweeps of round 4 (needs
# tools/probes/build_packed.sh first).
cd "$GRAFT_REPO_ROOT/tools/probes" || exit 1
timeout 200 ./pk_repro opsel4 3
timeout 200 ./pk_repro opsel3 2 | head -8
if [ "$1" = "full" ]; then
  for co in chain mfma valu mem none; do timeout 120 ./pk_repro lib packed/libafm_hip.so 20 $co; done
  for v in 1 2 4 8 16 32 64 128; do [ -f packed/libafm_hip_p$v.so ] && timeout 120 ./pk_repro lib packed/libafm_hip_p$v.so 30 chain; done
  timeout 120 ./pk_repro lib ../../afford-motion_amd/afm/libafm_hip.so 20 chain
fi
