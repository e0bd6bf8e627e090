#!/bin/bash
# one GPU call: parity suite, profiled bench of the 4K frame, bench of a 1/8-height frame
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/ -m gpu -x -q --timeout 300 --timeout-method thread > $R/t.log 2>&1
rc=$?
tail -3 $R/t.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/prof_rb -o rb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-count > $R/bench_rb.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python3 bench.py --height 270 --steps 10 --warmup 2 --no-cpu --no-count > $R/bench_small.log 2>&1 || exit 1
