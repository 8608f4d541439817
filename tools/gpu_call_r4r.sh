#!/bin/bash
# which C-ABI calls (by shape) the full-model CMDM training step spends its time in
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
( timeout 500 python tools/abi_call_profile.py --top 140 -- tools/bench_train.py --scene --cpu-steps 0 --steps 2 --warmup 1 ) > $O/train_full.log 2> $O/train_full_calls.txt
tail -3 $O/train_full.log | cut -c1-300
head -145 $O/train_full_calls.txt | cut -c1-260
