#!/bin/bash
# weight-stationary row-dot GEMM (CDM linear1): tests + CDM A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cdm.py -q -x --timeout=600 2>&1 | tail -12 ) > $O/pytest.log 2>&1
( timeout 600 python tools/cdm_ab.py 100 ) > $O/cdm_ab.jsonl 2> $O/cdm_ab.err
tail -8 $O/pytest.log; cut -c1-330 $O/cdm_ab.jsonl | head -4
