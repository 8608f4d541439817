#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_cdm.py -m gpu -q -x --timeout=600 --deselect tests/test_gpu_cdm.py::test_two_stream_loop_soak 2>&1 | tail -25 ) > $O/pytest.log 2>&1
( timeout 600 python tools/cdm_ab.py 100 ) > $O/cdm_ab.jsonl 2> $O/cdm_ab.err
cp afford-motion_amd/afm/libafm_hip.so /tmp/lib_keep.so; cp tools/probes/libafm_hip_dbg.so afford-motion_amd/afm/libafm_hip.so
( timeout 200 python tools/probes/toklin_timeline.py ) > $O/toklin_timeline.txt 2>&1
cp /tmp/lib_keep.so afford-motion_amd/afm/libafm_hip.so
tail -6 $O/pytest.log; tail -3 $O/cdm_ab.err; cut -c1-330 $O/cdm_ab.jsonl; tail -18 $O/toklin_timeline.txt
