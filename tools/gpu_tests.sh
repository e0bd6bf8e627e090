#!/bin/bash
# the -m gpu suite as the driver runs it, log under gpurun_out/
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/ -m gpu -x -q --timeout 300 --timeout-method thread --durations=5 > $R/t_${1:-x}.log 2>&1
rc=$?
tail -12 $R/t_${1:-x}.log
exit $rc
