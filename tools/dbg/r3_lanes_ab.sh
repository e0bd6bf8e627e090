#!/bin/bash
# A/B of k_closest_lanes' rays-per-wave thresholds (library built with EXTRA=-DRTX_AB_KNOBS)
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for cfg in "131072 32768" "262144 32768" "524288 32768" "1048576 32768" "262144 65536" "524288 131072" "2097152 32768" "2097152 131072"; do
set -- $cfg
RTX_LANES16=$1 RTX_LANES4=$2 timeout -k 10 200 python -u tools/dbg/ab.py 0 c4 > $R/la_$1_$2.log 2>&1 || { tail -5 $R/la_$1_$2.log; exit 1; }
echo "lanes16=$1 lanes4=$2: $(cat $R/la_$1_$2.log)"
done
