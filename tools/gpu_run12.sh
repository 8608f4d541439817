#!/bin/bash
# folded CDM sampling form: parity tests, then A/B of the configs[2] loop with the fold on / off in one call
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_cdm.py tests/test_gpu_ops.py tests/test_gpu_c_abi.py -q -x --timeout=900 2>&1 | grep -v "^$" | tail -25 ) > gpurun_out/r02j_pytest.log 2>&1
tail -15 gpurun_out/r02j_pytest.log
( timeout 300 python tools/bench_configs.py --only config2 ) > gpurun_out/r02j_cdm_fold.jsonl 2> gpurun_out/r02j_cdm_fold.err
( AFM_CDM_NO_FOLD=1 timeout 300 python tools/bench_configs.py --only config2 ) > gpurun_out/r02j_cdm_nofold.jsonl 2> gpurun_out/r02j_cdm_nofold.err
tail -2 gpurun_out/r02j_cdm_fold.jsonl | cut -c1-900; tail -3 gpurun_out/r02j_cdm_fold.err
tail -2 gpurun_out/r02j_cdm_nofold.jsonl | cut -c1-900
