#!/bin/bash
# A/B of the shadow-pass split thresholds (rays up to which a generation is traced by 4 / 8 workgroups per 256 rays)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "1048576 131072" "2097152 131072" "4194304 131072" "4194304 524288" "8388608 1048576" "4194304 2097152"; do
  set -- $cfg
  echo -n "split4=$1 split8=$2  "; RTX_SPLIT4=$1 RTX_SPLIT8=$2 timeout -k 10 200 python3 tools/dbg/ab.py 0 ${WL:-c4} || exit 1
done; done
