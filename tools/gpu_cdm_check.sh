#!/bin/bash
# CDM parity tests + A/B of the sampling forms (one gpurun call)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/cdm_check; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_cdm.py -q -x --timeout=600 -s 2>&1 | grep -v "^$" | tail -40 ) > $O/pytest.log 2>&1
( timeout 600 python tools/cdm_ab.py 100 ) > $O/cdm_ab.jsonl 2> $O/cdm_ab.err
grep "passed\|failed\|default\|Error" $O/pytest.log | tail -12 | cut -c1-200; cut -c1-300 $O/cdm_ab.jsonl | head -3; tail -2 $O/cdm_ab.err
