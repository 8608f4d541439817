#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cmdm.py -m gpu -q -x --timeout=600 -k "mha or small_batch or sharding or last_layer or full_size or golden" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
( timeout 400 python tools/small_batch_probe.py 200 ) > $O/small_batch.jsonl 2> $O/small_batch.err
( timeout 300 python tools/bench_train.py --cpu-steps 0 ) > $O/train_trunk.json 2> $O/train_trunk.err
( timeout 300 python tools/bench_train.py --scene --cpu-steps 0 --steps 10 ) > $O/train_full.json 2> $O/train_full.err
( timeout 300 python tools/bench_train.py --cdm --steps 10 ) > $O/train_cdm.json 2> $O/train_cdm.err
tail -6 $O/pytest.log; cat $O/small_batch.jsonl | cut -c1-120; for f in trunk full cdm; do tail -1 $O/train_$f.json | cut -c1-1800; tail -2 $O/train_$f.err; done
