#!/bin/bash
# full -m gpu suite, smoke, and the kernel list of a two-stage sampling job (no at::native arithmetic expected)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 2>&1 | tail -15 ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg4 -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --quick --only config4 ) > $GRAFT_REPO_ROOT/$O/cfg4.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_cfg4 -name "*kernel_stats.csv" | head -1)
tail -8 $O/pytest_gpu.log; tail -3 $O/smoke.log; tail -2 $O/cfg4.log | cut -c1-300
echo "--- kernels of the two-stage job ($f)"; [ -n "$f" ] && cut -d, -f1-4 $f | cut -c1-120 | head -45
find $O/prof_cfg4 -name "*kernel_trace.csv" -delete
