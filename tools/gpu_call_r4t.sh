#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04t; mkdir -p $O
( timeout 300 python tools/bench_train.py --scene --cpu-steps 0 --steps 10 --warmup 3 ) > $O/train_full.json 2> $O/err.txt
( timeout 300 python tools/bench_train.py --cpu-steps 0 --steps 20 --warmup 3 ) > $O/train_trunk.json 2>> $O/err.txt
( timeout 300 python tools/bench_train.py --cdm --cpu-steps 0 --steps 10 --warmup 3 ) > $O/train_cdm.json 2>> $O/err.txt
for f in full trunk cdm; do tail -1 $O/train_$f.json | cut -c1-1500; done
