#!/bin/bash
# round 4, first GPU call: phase-A validation (tests incl. the new multi-rank / no-eager-math / attention-rows tests, smoke, bench with the
# secondary block) + diagnostics for the small-launch regime and the CDM sub-batch streams.  Outputs under gpurun_out/r04a/.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -25 ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_k20.json 2> $O/bench_k20.err
( timeout 120 tools/gemm_timeline small ) > $O/gemm_timeline_small.txt 2>&1
( timeout 400 python tools/cdm_streams_probe.py 100 ) > $O/cdm_streams.jsonl 2> $O/cdm_streams.err
( timeout 400 python tools/small_batch_probe.py 200 ) > $O/small_batch.jsonl 2> $O/small_batch.err
tail -12 $O/pytest_gpu.log; tail -3 $O/smoke.log; tail -3 $O/bench_k20.err
python - $O <<'PY'
import json,sys
O=sys.argv[1]
try:
    d=json.loads([l for l in open(f"{O}/bench_k20.json") if l.startswith("{")][-1])
    print("bench", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("sample_latency"))
    print("secondary", json.dumps(d.get("secondary"))[:3000])
except Exception as e:
    print("bench ERR", e)
PY
cat $O/gemm_timeline_small.txt | grep -v "^   resident\|distinct CUs" | cut -c1-330
cut -c1-260 $O/cdm_streams.jsonl; tail -3 $O/cdm_streams.err
cat $O/small_batch.jsonl; tail -3 $O/small_batch.err
