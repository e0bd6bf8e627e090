#!/bin/bash
# A/B of library builds: tools/dbg/ab_libs.sh <lib.so> ...   (paths relative to the repo root; "-" = the default build)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then echo -n "default   "; timeout -k 10 120 python3 tools/dbg/ab.py 0 || exit 1
  else echo -n "$L  "; RTX_LIB=$GRAFT_REPO_ROOT/$L timeout -k 10 120 python3 tools/dbg/ab.py 0 || exit 1; fi
done; done
