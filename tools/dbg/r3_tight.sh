#!/bin/bash
# tight table entries: full suite, fuzz, counters and frame times (library built with EXTRA=-DRTX_AB_KNOBS)
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/ -m gpu -x -q --timeout 300 --timeout-method thread > $R/tt_tests.log 2>&1 || { tail -30 $R/tt_tests.log; exit 1; }
tail -2 $R/tt_tests.log
timeout -k 10 400 python -u tools/dbg/fuzz_exact.py 16 > $R/tt_fuzz.log 2>&1 || { tail -20 $R/tt_fuzz.log; exit 1; }
tail -2 $R/tt_fuzz.log
for t in 0 1; do
RTX_CELL_TIGHT=$t RTX_SHADOW_FORM=1 timeout -k 10 300 python -u tools/dbg/r3_counts.py c4 > $R/tt_c4_$t.log 2>&1 || { tail -5 $R/tt_c4_$t.log; exit 1; }
echo "== c4 tight=$t"; cat $R/tt_c4_$t.log
done
RTX_SHADOW_FORM=1 timeout -k 10 300 python -u tools/dbg/r3_counts.py c5 > $R/tt_c5_1.log 2>&1 || { tail -5 $R/tt_c5_1.log; exit 1; }
echo "== c5 tight=1 (tables forced)"; cat $R/tt_c5_1.log
timeout -k 10 300 python -u tools/dbg/ab.py 0 c4 && timeout -k 10 300 python -u tools/dbg/ab.py 0 c5
