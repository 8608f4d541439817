#!/bin/bash
# end-of-round validation and artifacts (one gpurun call): full -m gpu suite, smoke, bench (default with `secondary`, the driver's command),
# rocprofv3 stats + PMC passes (CMDM + CDM), training benches.  Outputs under gpurun_out/$R/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=${1:-r04}
O=gpurun_out/$R; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -6 ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 900 python bench.py ) > $O/bench_b32.json 2> $O/bench_b32.err
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_k20.json 2> $O/bench_k20.err
( timeout 600 bash tools/collect_profiles.sh ${R/r0/r} ) > $O/collect.log 2>&1
( timeout 500 bash tools/collect_profiles.sh ${R/r0/r} cdm ) > $O/collect_cdm.log 2>&1
( timeout 300 python tools/bench_train.py --scene --cpu-steps 0 --steps 10 --warmup 3 ) > $O/train_full.json 2> $O/train.err
( timeout 300 python tools/bench_train.py --cpu-steps 0 --steps 20 --warmup 3 ) > $O/train_trunk.json 2>> $O/train.err
( timeout 300 python tools/bench_train.py --cdm --cpu-steps 0 --steps 10 --warmup 3 ) > $O/train_cdm.json 2>> $O/train.err
tail -4 $O/pytest_gpu.log; tail -3 $O/smoke.log
python - $O <<'PY'
import json,sys
O=sys.argv[1]
for f in ("bench_b32","bench_k20"):
    try:
        d=json.loads([l for l in open(f"{O}/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], (d["roofline"]["traffic"] or {}), d.get("sample_latency"), d.get("preflight"))
        if d.get("secondary"): print(json.dumps(d["secondary"])[:3000])
    except Exception as e:
        print(f, "ERR", e)
PY
for f in full trunk cdm; do tail -1 $O/train_$f.json | cut -c1-330; done
