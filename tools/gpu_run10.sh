#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 tools/kernel_sweep gemm ${1:-32} ) > gpurun_out/r02_gemm_sweep_swz.txt 2>&1
echo "sweep: identical $(grep -c bit-identical gpurun_out/r02_gemm_sweep_swz.txt) differs $(grep -c DIFFERS gpurun_out/r02_gemm_sweep_swz.txt) rc-lines $(grep -c 'rc=' gpurun_out/r02_gemm_sweep_swz.txt)"
grep "x9" gpurun_out/r02_gemm_sweep_swz.txt | grep -v "BK32\|pl \|half" | cut -c1-108
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -x --timeout=600 2>&1 | grep -v "^$" | tail -5 ) > gpurun_out/r02h_pytest.log 2>&1
tail -3 gpurun_out/r02h_pytest.log
( timeout 200 python bench.py --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02h_bench.json 2> gpurun_out/r02h_bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r02h_bench.json') if l.startswith('{')][-1])
    print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d['roofline']['all_kernels_tflops'])
except Exception as e:
    print('ERR', e); print(open('gpurun_out/r02h_bench.err').read()[-1500:])
PY
