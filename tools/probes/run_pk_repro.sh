#!/bin/bash
# one GPU call: the stand-alone reproducer over the library builds and the co-runners
cd "$GRAFT_REPO_ROOT/tools/probes" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O
{
for co in chain mfma valu mem none; do timeout 120 ./pk_repro lib packed/libafm_hip.so 20 $co; done
for v in 1 2 4 8 16 32 64 128; do [ -f packed/libafm_hip_p$v.so ] && timeout 120 ./pk_repro lib packed/libafm_hip_p$v.so 30 chain; done
timeout 120 ./pk_repro lib ../../afford-motion_amd/afm/libafm_hip.so 20 chain
} > $O/pk_repro_sweep.txt 2>&1
cat $O/pk_repro_sweep.txt
