#!/bin/bash
# A/B library variants for tools/gpu_call.sh ab: tools/build_variant.sh <name> "<extra hipcc flags>" <source.hip> [more sources]
# compiles the named csrc/ sources with the extra flags and links them with the library's other (already built) objects into
# tools/ab_libs/libafm_<name>.so.  Run `python afford-motion_amd/build_hip.py` first.
set -e
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=$2; shift 2
B=afford-motion_amd/build; T=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iafford-motion_amd/csrc -Werror=pass-failed -Xclang -target-feature -Xclang -packed-fp32-ops"
OBJS=""
for o in $B/*.o; do
  base=$(basename $o .o); hit=0
  for s in "$@"; do [ "$(basename $s .hip)" = "$base" ] && hit=1; done
  [ $hit = 0 ] && OBJS="$OBJS $o"
done
for s in "$@"; do
  hipcc $FLAGS $EXTRA -c afford-motion_amd/csrc/$(basename $s) -o $T/$(basename $s .hip).o 2>&1 | grep -E "error" || true
  OBJS="$OBJS $T/$(basename $s .hip).o"
done
mkdir -p tools/ab_libs
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/ab_libs/libafm_$NAME.so
rm -rf $T
ls -la tools/ab_libs/libafm_$NAME.so
