#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_cdm.py -m gpu -q -x --timeout=600 --deselect tests/test_gpu_cdm.py::test_two_stream_loop_soak 2>&1 | tail -25 ) > $O/pytest.log 2>&1
( timeout 600 python tools/cdm_ab.py 100 ) > $O/cdm_ab.jsonl 2> $O/cdm_ab.err
tail -15 $O/pytest.log; tail -3 $O/cdm_ab.err; cut -c1-700 $O/cdm_ab.jsonl
