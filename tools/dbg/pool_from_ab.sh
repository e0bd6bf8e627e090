#!/bin/bash
# A/B of RTX_SHADOW_POOL_FROM (first generation on the pooled shadow stages; 99 = none) for the default library and each library given
cd $GRAFT_REPO_ROOT
WL=${WL:-c4}
for rep in 1 2; do
for L in - "$@"; do
  for pf in ${PFS:-0 1 2 99}; do
    if [ "$L" = "-" ]; then echo -n "default pool_from=$pf  "; RTX_SHADOW_POOL_FROM=$pf timeout -k 10 200 python3 tools/dbg/ab.py 0 $WL || exit 1
    else echo -n "$L pool_from=$pf  "; RTX_LIB=$GRAFT_REPO_ROOT/$L RTX_SHADOW_POOL_FROM=$pf timeout -k 10 200 python3 tools/dbg/ab.py 0 $WL || exit 1; fi
  done
done; done
