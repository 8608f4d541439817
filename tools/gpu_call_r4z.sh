#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_cdm.py -m gpu -q -x --timeout=600 -k "pointtrans" -s 2>&1 ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log | cut -c1-250
