#!/bin/bash
# full GPU suite, random-scene fuzz (culled vs exact-only kernels), executed-test counts and frame times of C4 / C5
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
T=${1:-v}
timeout -k 10 900 python -u -m pytest tests/ -m gpu -x -q --timeout 300 --timeout-method thread --durations=5 > $R/t_$T.log 2>&1 || { tail -20 $R/t_$T.log; exit 1; }
tail -4 $R/t_$T.log
timeout -k 10 300 python -u tools/dbg/fuzz_exact.py 8 > $R/fz_$T.log 2>&1 || { tail -20 $R/fz_$T.log; exit 1; }
tail -2 $R/fz_$T.log
timeout -k 10 200 python -u tools/dbg/r3_counts.py c4 > $R/cnt_$T.log 2>&1 || { tail -20 $R/cnt_$T.log; exit 1; }
tail -12 $R/cnt_$T.log
timeout -k 10 200 python -u tools/dbg/ab.py 0 c4 > $R/ab4_$T.log 2>&1 || { tail -20 $R/ab4_$T.log; exit 1; }
cat $R/ab4_$T.log
timeout -k 10 300 python -u tools/dbg/ab.py 0 c5 > $R/ab5_$T.log 2>&1 || { tail -20 $R/ab5_$T.log; exit 1; }
cat $R/ab5_$T.log
