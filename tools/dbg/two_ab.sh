#!/bin/bash
# A/B of RTX_TWO_LEVEL (unset = the context's own choice by frame size, 0 = never, 1 = always)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for t in auto 0 1; do
  if [ "$t" = "auto" ]; then unset RTX_TWO_LEVEL; else export RTX_TWO_LEVEL=$t; fi
  echo -n "two_level=$t  "; timeout -k 10 300 python3 tools/dbg/ab.py 0 ${WL:-c4} || exit 1
done; done
