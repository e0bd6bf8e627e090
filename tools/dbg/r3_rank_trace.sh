#!/bin/bash
# kernel trace of a rank of 8 of the 4K frame (timeline of the last frame)
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/prof_rank8 -o rank -- python3 $GRAFT_REPO_ROOT/tools/dbg/one_rank.py > $R/rank8.log 2>&1 || { tail -5 $R/rank8.log; exit 1; }
cat $R/rank8.log | tail -1
