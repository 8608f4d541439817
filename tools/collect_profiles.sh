#!/bin/bash
# Round profile artifacts (run on the GPU box through gpurun): kernel-trace stats of the bench command and
# separate PMC passes (HBM read / write bytes, MFMA busy) as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
#   tools/collect_profiles.sh r2          CMDM headline path (bench.py) -> gpurun_out/prof_r2
#   tools/collect_profiles.sh r2 cdm      CDM Perceiver loop (BASELINE configs[2]) -> gpurun_out/prof_r2_cdm     (cdm_h: its HUMANISE variant)
#   tools/collect_profiles.sh r2 points   set abstraction (BASELINE configs[3]: FPS, kNN, fused gather-MLP-max) -> gpurun_out/prof_r2_points
set -u
R=${1:-r1}
WHICH=${2:-cmdm}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$R
[ "$WHICH" = "cdm" ] && OUT=${OUT}_cdm
[ "$WHICH" = "cdm_h" ] && OUT=${OUT}_cdm_h
[ "$WHICH" = "points" ] && OUT=${OUT}_points
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMC="python $ROOT/tools/pmc_target.py $WHICH"
# round 6: the two-stream loop runs its wide GEMMs on 128 x 128 tiles (csrc/cmdm.hip); the single-stream profile targets force the same tile
# program (bit-identical) so that the kernel profiled here IS the dominant kernel of the bench line (bench.py's roofline pass does the same)
[ "$WHICH" = "cmdm" ] && export AFM_PROFILE_TILE=5
if [ "$WHICH" = "cdm" ] || [ "$WHICH" = "cdm_h" ] || [ "$WHICH" = "points" ]; then
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PMC > /dev/null 2>&1
else
  BENCH="python $ROOT/bench.py --streams 1 --gemm-tile 5 --steps 100 --warmup 10 --latency-runs 0 --latency-runs-b1 0 --no-cpu-baseline --no-alt-gemm --no-secondary"
  if [ "${SKIP_STATS:-0}" != "1" ]; then timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_rocprof.json 2>/dev/null; fi
fi
# counter passes on a lean target (rocprofv3 --pmc segfaults around the full bench process): 12 steps, same shapes.  Every pass in its
# own timeout (a crashed --pmc pass can hang), counters only with --kernel-trace (gpurun refuses --pmc together with other trace domains)
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMC > /dev/null 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMC > /dev/null 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $PMC > /dev/null 2>&1
python $ROOT/tools/summarize_profiles.py $OUT $WHICH > $OUT/summary.md 2>&1
cat $OUT/summary.md
# the raw kernel traces are large; keep only stats + the summary
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
