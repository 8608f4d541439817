#!/bin/bash
# CDM parity tests + A/B of the sampling forms (one gpurun call)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/cdm_check; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_cdm.py -q -x --timeout=600 -s 2>&1 | grep -v "^$" | tail -40 ) > $O/pytest.log 2>&1
( timeout 600 python tools/cdm_ab.py 100 ) > $O/cdm_ab.jsonl 2> $O/cdm_ab.err
grep "passed\|failed\|default\|Error" $O/pytest.log | tail -12 | cut -c1-200; cut -c1-220 $O/cdm_ab.jsonl; tail -2 $O/cdm_ab.err
# per-kernel durations of the default form (rocprofv3 --stats of 12 native steps)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/tools/pmc_target.py cdm > /dev/null 2>&1 )
find $O/stats -name "*kernel_trace.csv" -delete
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_h -- python $GRAFT_REPO_ROOT/tools/pmc_target.py cdm_h > /dev/null 2>&1 )
find $O/stats_h -name "*kernel_trace.csv" -delete
python - $O <<'PY'
import csv,glob,sys
for d in ('stats', 'stats_h'):
    f=glob.glob(sys.argv[1]+'/'+d+'/**/*kernel_stats.csv',recursive=True)
    print(d)
    for r in (list(csv.DictReader(open(f[0])))[:10] if f else []):
        print(f"{r['Name'][:56]:56s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Percentage']}%")
PY
