#!/bin/bash
# pooled shadow stage: parity suite on the default build, then A/B timing of the library builds given as arguments
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/ -m gpu -x -q --timeout 300 --timeout-method thread > $R/t_pool.log 2>&1
rc=$?
tail -6 $R/t_pool.log
[ $rc -ne 0 ] && exit $rc
bash tools/dbg/ab_libs.sh "$@" 2>&1 | tee $R/ab_pool.log
