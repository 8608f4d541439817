#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04s; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py tests/test_gpu_points.py tests/test_gpu_train_ddp.py -m gpu -q -x --timeout=600 2>&1 | tail -15 ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
( timeout 500 python tools/abi_call_profile.py --top 140 -- tools/bench_train.py --scene --cpu-steps 0 --steps 2 --warmup 1 ) > $O/train_full.log 2> $O/train_full_calls.txt
tail -1 $O/train_full.log | cut -c1-400
grep "afm_linear_wgrad \|afm_linear " $O/train_full_calls.txt | head -70 | cut -c1-200
