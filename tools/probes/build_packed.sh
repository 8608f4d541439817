#!/bin/bash
# Packed-f32 defect hunt (profiles/r03_packed_f32_defect.md, profiles/r04_packed_f32_repro.md).  Builds
#   tools/probes/packed/libafm_hip.so           the library with packed-f32 VALU instructions ALLOWED in the GEMM translation units (gemm.hip,
#                                               gemm_split.hip, gemm_slab.hip: the statistics-carrying epilogue lives there), other objects as shipped
#   tools/probes/packed/libafm_hip_p<N>.so      the same with -DAFM_PK_PROBE=<N> (gemm_epilogue.h): 1 = loads landed + idle cycles before the packed
#                                               arithmetic, 2 = every trip drained, 4 = inputs pinned in registers, 8 = idle cycles behind it,
#                                               16 / 32 / 64 / 128 = the instruction forms written out in inline asm (see gemm_epilogue.h)
#   tools/probes/pk_repro                       the stand-alone reproducer (dlopens the library it is given)
# Run after afford-motion_amd/build_hip.py.  Never part of the product.
cd "$(dirname "$0")/../.." || exit 1
OBJ=afford-motion_amd/build
mkdir -p tools/probes/packed
others=$(ls $OBJ/*.o | grep -v "/gemm.o\|/gemm_split.o\|/gemm_slab.o")
for v in ${PK_VARIANTS:-0 1 2 4 8 16 32 64 128}; do
  for f in gemm gemm_split gemm_slab; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iafford-motion_amd/csrc -DAFM_PK_PROBE=$v -c afford-motion_amd/csrc/$f.hip -o /tmp/pk_${f}_$v.o 2>/dev/null || exit 1
  done
  out=tools/probes/packed/libafm_hip$([ $v = 0 ] || echo _p$v).so
  hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/pk_gemm_$v.o /tmp/pk_gemm_split_$v.o /tmp/pk_gemm_slab_$v.o -o $out || exit 1
  ( cd /tmp && rm -rf pk_scan && mkdir pk_scan && cd pk_scan && cp /tmp/pk_gemm_split_$v.o x.o && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1 &&
    for co in $(ls | grep gfx950); do echo "$out: gemm_split has $(/opt/rocm/lib/llvm/bin/llvm-objdump -d $co | grep -c 'v_pk_\(fma\|mul\|add\)_f32') v_pk_{fma,mul,add}_f32"; done )
done
hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude tools/probes/pk_repro.hip -ldl -o tools/probes/pk_repro || exit 1
