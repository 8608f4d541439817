#!/bin/bash
# ONE parameterised runner for everything that goes through gpurun (replaces the one-off tools/gpu_call_r4*.sh / r5*.sh scripts of earlier rounds):
#   gpurun --timeout N -- 'bash tools/gpu_call.sh <recipe> [args...]'
# Every recipe writes under gpurun_out/<recipe>/ and prints the lines worth reading at the end of the call.
#   gemm_small               bit-identity sweep + per-phase timelines of the split-K forms (two-stage vs three-stage), small launches
#   ab <tag> <B...>          A/B of the library builds under tools/ab_libs/libafm_*.so inside ONE call (bench.py --batch B, alternating, two repetitions)
#   flags "<B...>" "<f1>" "<f2>"   A/B of bench.py flag sets with the built library (alternating, two repetitions)
#   ab_cdm                   the same for the CDM loop (tools/cdm_ab.py)
#   tests [pytest args]      pytest -m gpu (default: the whole suite)
#   small_batch              kernel_sweep bit-identity + CMDM tests + bench at B = 4 / 1 / 32
#   cdm_check                CDM parity tests + tools/cdm_ab.py + rocprofv3 --stats of the CDM loop (H3D and HUMANISE variants)
#   points                   configs[3] kernels: timings + PMC passes (FETCH / WRITE / SQ) -> points_summary.md
#   pk_repro                 the packed-f32 reproducer with its controls
#   validate <rNN>           end-of-round validation: full -m gpu suite, smoke, both bench commands, rocprofv3 stats + PMC passes (CMDM, CDM, points), training benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
RECIPE=${1:-validate}; shift
O=gpurun_out/$RECIPE; mkdir -p $O
LIB=afford-motion_amd/afm/libafm_hip.so
BENCH_LEAN="--no-secondary --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0"

bench_line() {        # <label> <json file>: value, ms/step, all-kernel ms/step
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d.get("roofline", {}).get("all_kernels_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}

case $RECIPE in
gemm_small)
  ( timeout 200 tools/kernel_sweep gemm 4,1 ) > $O/sweep.txt 2>&1
  echo "variants that differ: $(grep -c DIFFERS $O/sweep.txt)  bit-identical: $(grep -c 'bit-identical' $O/sweep.txt)"
  grep "x9" $O/sweep.txt | grep -v "motion_layer\|128x128" | cut -c1-125
  for bin in gemm_timeline_r5t2; do
    ( timeout 200 tools/$bin r5 ) > $O/$bin.txt 2>&1
    echo "---- $bin"; grep "^==\|phases" $O/$bin.txt | sed 's/arith=9 flags=0 pad=0 //; s/: [0-9]* workgroups, event time/ ev/; s/phases of thread 0, mean over workgroups (us): //; s/launch overhead.*//' | paste - - | cut -c1-330
  done
  ;;
ab)
  TAG=${1:-ab}; shift
  cp $LIB $O/orig.so
  trap 'cp $O/orig.so $LIB 2>/dev/null; rm -f $O/orig.so' EXIT          # a kill mid-recipe must not leave a variant build installed
  for rep in 1 2; do
    for so in tools/ab_libs/libafm_*.so; do
      v=$(basename $so .so); v=${v#libafm_}
      cp $so $LIB
      for b in ${@:-1 4}; do
        ( timeout 300 python bench.py --batch $b --steps 300 --warmup 30 $BENCH_LEAN ) > $O/${TAG}_${v}_b${b}_$rep.json 2>&1
        bench_line "$v rep $rep B=$b:" $O/${TAG}_${v}_b${b}_$rep.json
      done
    done
  done | tee $O/$TAG.txt
  ;;
flags)
  # A/B of bench.py FLAG sets with one library: tools/gpu_call.sh flags "<B list>" "<flags A>" "<flags B>" ... (alternating, two repetitions)
  BS=$1; shift
  for rep in 1 2; do
    i=0
    for fl in "$@"; do
      i=$((i+1))
      for b in $BS; do
        ( timeout 300 python bench.py --batch $b --steps 300 --warmup 30 $BENCH_LEAN $fl ) > $O/f${i}_b${b}_$rep.json 2>&1
        bench_line "[$fl] rep $rep B=$b:" $O/f${i}_b${b}_$rep.json
      done
    done
  done | tee $O/flags.txt
  ;;
ab_cdm)
  cp $LIB $O/orig.so
  trap 'cp $O/orig.so $LIB 2>/dev/null; rm -f $O/orig.so' EXIT
  for rep in 1 2; do
    for so in tools/ab_libs/libafm_*.so; do
      v=$(basename $so .so); v=${v#libafm_}
      cp $so $LIB
      ( timeout 300 python tools/cdm_ab.py ) > $O/cdm_${v}_$rep.jsonl 2>&1
      echo "$v rep $rep: $(grep -o '"steps_per_s": [0-9.]*' $O/cdm_${v}_$rep.jsonl | tr '\n' ' ')"
    done
  done | tee $O/ab_cdm.txt
  ;;
arith)
  # worst-case table of the GEMM arithmetics, then the WHOLE -m gpu suite with six products as the process default
  # (AFM_GEMM_SPLIT=6 is read by afm/ops.py at import), then the headline under both settings in this one call
  ( timeout 600 python -m pytest tests/test_gpu_arith.py -q -s --timeout=600 2>&1 | grep "\[arith\]\|passed\|failed\|Error\|assert" ) > $O/arith.log 2>&1
  cat $O/arith.log | cut -c1-220
  ( time AFM_GEMM_SPLIT=6 timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -40 ) > $O/pytest_x6.log 2>&1
  tail -45 $O/pytest_x6.log | cut -c1-250
  for rep in 1 2; do
    for sp in 9 6; do
      ( AFM_GEMM_SPLIT=$sp timeout 300 python bench.py --steps 200 --warmup 20 $BENCH_LEAN ) > $O/bench_x${sp}_$rep.json 2>&1
      bench_line "x$sp rep $rep:" $O/bench_x${sp}_$rep.json
    done
  done
  ;;
pair_timeline)
  # kernel-trace timeline of the paired launch against the unpaired two-stream loop (same tile program: 128 x 128 everywhere)
  for pr in 0 1; do
    ( cd /tmp && AFM_PROFILE_STREAMS=2 AFM_PROFILE_TILE=5 AFM_PROFILE_PAIR=$pr timeout 240 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_pair$pr -- python $GRAFT_REPO_ROOT/tools/pmc_target.py cmdm > /dev/null 2>&1 )
  done
  f0=$(find $O/trace_pair0 -name "*kernel_trace.csv" | head -1); f1=$(find $O/trace_pair1 -name "*kernel_trace.csv" | head -1)
  python tools/pair_timeline.py "$f0" "$f1" > $O/pair_timeline.md 2> $O/pair_timeline.err
  find $O -name "*kernel_trace.csv" -size +20M -delete
  cat $O/pair_timeline.md | cut -c1-200; tail -3 $O/pair_timeline.err
  ;;
tests)
  ( time timeout 1500 python -m pytest ${@:-tests -m gpu} -q -x --timeout=900 2>&1 | tail -8 ) 2>&1 | tee $O/pytest.log
  ;;
small_batch)
  ( timeout 300 tools/kernel_sweep gemm 32,8,4,1 ) > $O/sweep.txt 2>&1
  echo "variants that differ: $(grep -c DIFFERS $O/sweep.txt)  bit-identical: $(grep -c 'bit-identical' $O/sweep.txt)"
  ( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py -q -x --timeout=600 2>&1 | tail -5 ) > $O/pytest.log 2>&1
  tail -3 $O/pytest.log
  for b in 4 1 32; do
    ( timeout 300 python bench.py --batch $b --steps 200 --warmup 20 $BENCH_LEAN ) > $O/bench_b$b.json 2>&1
    bench_line "B=$b:" $O/bench_b$b.json
  done
  ;;
cdm_check)
  # CDM parity tests + A/B of the sampling forms + per-kernel durations of the default form (both variants)
  ( timeout 900 python -m pytest tests/test_gpu_cdm.py -q -x --timeout=600 -s 2>&1 | grep -v "^$" | tail -40 ) > $O/pytest.log 2>&1
  ( timeout 600 python tools/cdm_ab.py 100 ) > $O/cdm_ab.jsonl 2> $O/cdm_ab.err
  grep "passed\|failed\|default\|Error" $O/pytest.log | tail -12 | cut -c1-200; cut -c1-220 $O/cdm_ab.jsonl; tail -2 $O/cdm_ab.err
  for v in cdm cdm_h; do
    ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_$v -- python $GRAFT_REPO_ROOT/tools/pmc_target.py $v > /dev/null 2>&1 )
    find $O/stats_$v -name "*kernel_trace.csv" -delete
    f=$(find $O/stats_$v -name "*kernel_stats.csv" | head -1); echo "---- $v"; head -12 "$f" | cut -d, -f1-4 | cut -c1-160
  done
  ;;
points)
  ( timeout 600 bash tools/collect_profiles.sh ${1:-r5} points ) > $O/collect_points.log 2>&1
  tail -40 $O/collect_points.log
  ;;
pk_repro)
  ( timeout 600 bash tools/probes/run_pk_repro.sh ) > $O/pk_repro.txt 2>&1
  tail -60 $O/pk_repro.txt
  ;;
validate)
  R=${1:-r05}
  ( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -6 ) > $O/pytest_gpu.log 2>&1
  ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
  ( timeout 900 python bench.py ) > $O/bench_b32.json 2> $O/bench_b32.err
  ( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_k20.json 2> $O/bench_k20.err
  ( timeout 600 bash tools/collect_profiles.sh ${R/r0/r} ) > $O/collect.log 2>&1
  ( timeout 500 bash tools/collect_profiles.sh ${R/r0/r} cdm ) > $O/collect_cdm.log 2>&1
  ( timeout 500 bash tools/collect_profiles.sh ${R/r0/r} points ) > $O/collect_points.log 2>&1
  ( timeout 300 python tools/bench_train.py --scene --cpu-steps 0 --steps 10 --warmup 3 ) > $O/train_full.json 2> $O/train.err
  ( timeout 300 python tools/bench_train.py --cpu-steps 0 --steps 20 --warmup 3 ) > $O/train_trunk.json 2>> $O/train.err
  ( timeout 300 python tools/bench_train.py --cdm --cpu-steps 0 --steps 10 --warmup 3 ) > $O/train_cdm.json 2>> $O/train.err
  tail -4 $O/pytest_gpu.log; tail -3 $O/smoke.log
  python - $O <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_b32", "bench_k20"):
    try:
        d = json.loads([l for l in open(f"{O}/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], (d["roofline"]["traffic"] or {}), d.get("sample_latency"), d.get("preflight"))
        if d.get("secondary"):
            print(json.dumps(d["secondary"])[:3000])
    except Exception as e:
        print(f, "ERR", e)
PY
  for f in full trunk cdm; do tail -1 $O/train_$f.json | cut -c1-330; done
  ;;
*)
  echo "unknown recipe $RECIPE"; exit 2
  ;;
esac
