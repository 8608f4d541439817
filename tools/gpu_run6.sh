#!/bin/bash
# plane-image operands (ABI v4): kernel sweep, the new parity tests, A/B of the sampling loop with the images on / weights only / off
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 tools/kernel_sweep gemm 32,4 ) > gpurun_out/r02_gemm_sweep_planes.txt 2>&1
echo "sweep: identical $(grep -c bit-identical gpurun_out/r02_gemm_sweep_planes.txt) differs $(grep -c DIFFERS gpurun_out/r02_gemm_sweep_planes.txt) rc-lines $(grep -c 'rc=' gpurun_out/r02_gemm_sweep_planes.txt)"
grep "B=32" gpurun_out/r02_gemm_sweep_planes.txt | grep -v "f32 \|x6" | awk '{print $2, $6, $7, $8, $9, $10, $11, $12, $13}' | head -80
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py -q -x --timeout=600 -s 2>&1 | grep -v "^$" | tail -25 ) > gpurun_out/r02f_pytest.log 2>&1
tail -12 gpurun_out/r02f_pytest.log
run() {  # tag, env...
  tag=$1; shift
  ( env "$@" timeout 200 python bench.py --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02f_bench_$tag.json 2> gpurun_out/r02f_bench_$tag.err
  python - "$tag" <<'PY'
import json,sys
tag=sys.argv[1]
try:
    d=json.loads([l for l in open(f'gpurun_out/r02f_bench_{tag}.json') if l.startswith('{')][-1])
    print(tag, d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], d['roofline']['all_kernels_tflops'])
except Exception as e:
    print(tag, 'ERR', e); print(open(f'gpurun_out/r02f_bench_{tag}.err').read()[-1500:])
PY
}
run planes AFM_X=1
run wonly AFM_CMDM_NO_A_PLANES=1
run inkernel AFM_CMDM_NO_A_PLANES=1 AFM_CMDM_NO_W_PLANES=1
run planes2 AFM_X=1
( timeout 200 python bench.py --batch 4 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02f_bench_b4.json 2>&1
( AFM_CMDM_NO_A_PLANES=1 AFM_CMDM_NO_W_PLANES=1 timeout 200 python bench.py --batch 4 --no-cpu-baseline --no-alt-gemm --latency-runs 0 --latency-runs-b1 0 ) > gpurun_out/r02f_bench_b4_inkernel.json 2>&1
python - <<'PY'
import json
for f in ("r02f_bench_b4","r02f_bench_b4_inkernel"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d['roofline']['all_kernels_ms_per_step'])
    except Exception as e:
        print(f, "ERR", e)
PY
