#!/bin/bash
# A/B of whole library builds: variants/librtx_<tag>.so are copied over the library one by one (on the GPU box's copy of the tree), twice over
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in variants/librtx_*.so; do
  t=$(basename $f .so)
  cp $f ray-tracing_amd/librtx.so
  timeout -k 10 200 python -u tools/dbg/ab.py 0 c4 > $R/var_ab_$t.log 2>&1 || { tail -5 $R/var_ab_$t.log; exit 1; }
  echo "$t: $(cat $R/var_ab_$t.log)"
  timeout -k 10 200 python -u tools/dbg/ab.py 0 c5 > $R/var_ab5_$t.log 2>&1 || { tail -5 $R/var_ab5_$t.log; exit 1; }
  echo "$t: $(cat $R/var_ab5_$t.log)"
done
done
