#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 tools/kernel_sweep gemm 32,8,4,1 ) > gpurun_out/r02_kernel_sweep_items4.txt 2>&1
echo "sweep: identical $(grep -c bit-identical gpurun_out/r02_kernel_sweep_items4.txt) differs $(grep -c DIFFERS gpurun_out/r02_kernel_sweep_items4.txt) rc-lines $(grep -c 'rc=' gpurun_out/r02_kernel_sweep_items4.txt)"
grep "B=32\|B=4 " gpurun_out/r02_kernel_sweep_items4.txt | grep "x9" | cut -c1-100
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py tests/test_gpu_c_abi.py -q -x --timeout=900 -s 2>&1 | grep "bf16x1\|passed\|failed\|Error" | tail -12 ) > gpurun_out/r02i_pytest.log 2>&1
tail -8 gpurun_out/r02i_pytest.log
for b in 32 4; do
( timeout 300 python bench.py --batch $b --no-cpu-baseline --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02i_bench_b$b.json 2> gpurun_out/r02i_bench_b$b.err
python - $b <<'PY'
import json,sys
b=sys.argv[1]
try:
    d=json.loads([l for l in open(f'gpurun_out/r02i_bench_b{b}.json') if l.startswith('{')][-1])
    print(b, d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d['roofline']['all_kernels_tflops'], d.get('alt_gemm_modes'))
except Exception as e:
    print('ERR', e); print(open(f'gpurun_out/r02i_bench_b{b}.err').read()[-1500:])
PY
done
