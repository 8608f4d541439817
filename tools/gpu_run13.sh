#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in fold nofold; do
  rm -rf $ROOT/gpurun_out/cdmstats_$mode
  if [ $mode = nofold ]; then export AFM_CDM_NO_FOLD=1; else unset AFM_CDM_NO_FOLD; fi
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/cdmstats_$mode -- python $ROOT/tools/pmc_target.py cdm > /dev/null 2>&1
  f=$(find $ROOT/gpurun_out/cdmstats_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}")
PY
  find $ROOT/gpurun_out/cdmstats_$mode -name "*kernel_trace.csv" -delete
done
