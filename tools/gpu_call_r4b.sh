#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
( timeout 120 tools/probes/launch_shape ) > $O/launch_shape.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -40 ) > $O/pytest_gpu.log 2>&1
cat $O/launch_shape.txt; tail -30 $O/pytest_gpu.log
