#!/bin/bash
# tools/dbg/shadow_ab.sh <workload> <lib.so | -> ...
cd $GRAFT_REPO_ROOT
WL=$1; shift
for L in "$@"; do
  if [ "$L" = "-" ]; then echo -n "default   "; timeout -k 10 200 python3 tools/dbg/shadow_ab.py $WL || exit 1
  else echo -n "$L  "; RTX_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 200 python3 tools/dbg/shadow_ab.py $WL || exit 1; fi
done
