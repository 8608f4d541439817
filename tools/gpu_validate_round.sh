#!/bin/bash
# end-of-round validation and artifacts (one gpurun call): full -m gpu suite, smoke, bench (default, driver's command, small batches),
# profiles (CMDM + CDM), secondary configs, CDM A/B.  Outputs under gpurun_out/$R/ (R = round tag, default r03).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=${1:-r03}
O=gpurun_out/$R; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -6 ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 900 python bench.py ) > $O/bench_b32.json 2> $O/bench_b32.err
( timeout 300 python bench.py --steps 20 --warmup 5 ) > $O/bench_k20.json 2>&1
for b in 32 16 8 4 1; do
  ( timeout 300 python bench.py --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > $O/bench_small_b$b.json 2>&1
done
( timeout 600 bash tools/collect_profiles.sh ${R/r0/r} ) > $O/collect.log 2>&1
( timeout 500 bash tools/collect_profiles.sh ${R/r0/r} cdm ) > $O/collect_cdm.log 2>&1
# kernel stats of the HUMANISE variant of the CDM loop (41 input channels)
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_cdm_h -- python $GRAFT_REPO_ROOT/tools/pmc_target.py cdm_h > /dev/null 2>&1 )
find $O/stats_cdm_h -name "*kernel_trace.csv" -delete
( timeout 900 python tools/bench_configs.py ) > $O/configs.jsonl 2> $O/configs.err
( timeout 600 python tools/cdm_ab.py 100 ) > $O/cdm_ab.jsonl 2> $O/cdm_ab.err
tail -4 $O/pytest_gpu.log; tail -3 $O/smoke.log
python - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ("bench_b32","bench_k20","bench_small_b32","bench_small_b16","bench_small_b8","bench_small_b4","bench_small_b1"):
    try:
        d=json.loads([l for l in open(f"{O}/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], (d["roofline"]["traffic"] or {}).get("bytes_per_launch"), d.get("sample_latency"))
    except Exception as e:
        print(f, "ERR", e)
PY
cut -c1-300 $O/configs.jsonl; cut -c1-200 $O/cdm_ab.jsonl
