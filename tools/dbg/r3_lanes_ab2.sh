#!/bin/bash
# A/B of k_closest_lanes' rays-per-wave thresholds against the 5,120 resident wave slots (library built with EXTRA=-DRTX_AB_KNOBS): WL=c4|c5
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
WL=${WL:-c4}
for rep in 1 2; do
for cfg in "131072 32768" "81920 20480" "81920 32768" "131072 20480" "65536 16384" "98304 24576"; do
set -- $cfg
RTX_LANES16=$1 RTX_LANES4=$2 timeout -k 10 300 python -u tools/dbg/ab.py 0 $WL > $R/lb_$1_$2.log 2>&1 || { tail -5 $R/lb_$1_$2.log; exit 1; }
echo "lanes16=$1 lanes4=$2: $(cat $R/lb_$1_$2.log)"
done; done
