#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04w; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -6 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
bash tools/gpu_call_r4t.sh
( timeout 500 python tools/abi_call_profile.py --top 140 -- tools/bench_train.py --scene --cpu-steps 0 --steps 2 --warmup 1 ) > $O/train_full.log 2> $O/train_full_calls.txt
head -2 $O/train_full_calls.txt | tail -1
grep "N=32,K=32\|N=128,K=67\|N=64,K=32" $O/train_full_calls.txt | cut -c1-150
