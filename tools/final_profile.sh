#!/bin/bash
# The round's evidence: a clean bench.py line (with cpu_baseline) and the rocprofv3 kernel stats of the same command.
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $R/bench_final.json 2> $R/bench_final.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_final -o final -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu > $R/bench_prof.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
cut -c1-600 $R/bench_final.json
ls $R/prof_final
