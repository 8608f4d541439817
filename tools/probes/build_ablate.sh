#!/bin/bash
# builds tools/probes/ablate/libafm_hip_abl<bits>.so: the library with one ingredient of the 64x64 nine-product GEMM's K loop removed
# (-DAFM_ABLATE, gemm_split.hip; wrong results, same control flow) for tools/gpu_power_ablate.sh.  Run after afford-motion_amd/build_hip.py.
cd "$(dirname "$0")/../.." || exit 1
OBJ=afford-motion_amd/build
mkdir -p tools/probes/ablate
for v in 1 2 4 8 6 7 16; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude -Iafford-motion_amd/csrc -DAFM_ABLATE=$v \
    -c afford-motion_amd/csrc/gemm_split.hip -o /tmp/gs_$v.o 2>/dev/null || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v gemm_split.o) /tmp/gs_$v.o -o tools/probes/ablate/libafm_hip_abl$v.so && echo "built ablation $v"
done
