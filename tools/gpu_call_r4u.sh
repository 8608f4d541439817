#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04u; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_points.py tests/test_gpu_train_ddp.py -m gpu -q -x --timeout=600 2>&1 | tail -8 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( timeout 300 python tools/bench_train.py --scene --cpu-steps 0 --steps 10 --warmup 3 ) > $O/train_full.json 2> $O/err.txt
tail -1 $O/train_full.json | cut -c1-700
