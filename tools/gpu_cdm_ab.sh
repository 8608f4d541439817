#!/bin/bash
# batched latent chain: parity tests, A/B against the serial latent_post, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
ROOT=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_cdm.py tests/test_gpu_c_abi.py -q -x --timeout=900 2>&1 | grep -v "^$" | tail -25 ) > gpurun_out/r02k_pytest.log 2>&1
tail -12 gpurun_out/r02k_pytest.log
( timeout 300 python tools/bench_configs.py --only config2 ) > gpurun_out/r02k_cdm_chain.jsonl 2> gpurun_out/r02k_cdm_chain.err
( AFM_CDM_SERIAL_LATENT=1 timeout 300 python tools/bench_configs.py --only config2 ) > gpurun_out/r02k_cdm_serial.jsonl 2> gpurun_out/r02k_cdm_serial.err
tail -1 gpurun_out/r02k_cdm_chain.jsonl | cut -c1-700; tail -3 gpurun_out/r02k_cdm_chain.err
tail -1 gpurun_out/r02k_cdm_serial.jsonl | cut -c1-700
cd /tmp
rm -rf $ROOT/gpurun_out/cdmstats_chain
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/cdmstats_chain -- python $ROOT/tools/pmc_target.py cdm > /dev/null 2>&1
f=$(find $ROOT/gpurun_out/cdmstats_chain -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']}")
PY
find $ROOT/gpurun_out/cdmstats_chain -name "*kernel_trace.csv" -delete
