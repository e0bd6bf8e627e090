#!/bin/bash
# A/B of one environment knob (library built with EXTRA=-DRTX_AB_KNOBS): r3_knob.sh NAME value value ...
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
K=$1; shift
for v in "$@"; do
  env $K=$v timeout -k 10 200 python -u tools/dbg/ab.py 0 c4 > $R/knob_$v.log 2>&1 || { tail -5 $R/knob_$v.log; exit 1; }
  echo "$K=$v c4: $(cat $R/knob_$v.log)"
  env $K=$v timeout -k 10 200 python -u tools/dbg/ab.py 0 c5 > $R/knob5_$v.log 2>&1 || { tail -5 $R/knob5_$v.log; exit 1; }
  echo "$K=$v c5: $(cat $R/knob5_$v.log)"
done
