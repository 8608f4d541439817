#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -6 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( timeout 300 python tools/bench_train.py --cpu-steps 0 --steps 20 --warmup 3 ) 2>/dev/null | tail -1 | cut -c1-300
