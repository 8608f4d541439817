#!/bin/bash
# Round profile artifacts (run on the GPU box through gpurun): kernel-trace stats of the bench command and
# separate PMC passes (HBM read / write bytes, MFMA busy) as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
set -u
R=${1:-r1}
OUT=/root/repo/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python /root/repo/bench.py --streams 1 --steps 100 --warmup 10 --latency-runs 0 --no-cpu-baseline --no-alt-gemm"
if [ "${SKIP_STATS:-0}" != "1" ]; then timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_rocprof.json 2>/dev/null; fi
# counter passes on a lean target (rocprofv3 --pmc segfaults around the full bench process): 12 steps, same shapes
PMC="python /root/repo/tools/pmc_target.py"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMC > /dev/null 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMC > /dev/null 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- $PMC > /dev/null 2>&1
python /root/repo/tools/summarize_profiles.py $OUT > $OUT/summary.md 2>&1
cat $OUT/summary.md
# the raw kernel traces are large; keep only stats + the summary
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
