#!/bin/bash
# round 3: the measured choice between the two forms of the shadow stage -- full GPU suite, C4 / C5 frame times forced either way and chosen
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/ -m gpu -x -q --timeout 300 --timeout-method thread > $R/f_tests.log 2>&1 || { tail -30 $R/f_tests.log; exit 1; }
tail -2 $R/f_tests.log
for wl in c4 c5; do
for form in 8192 4096 0; do
timeout -k 10 300 python -u tools/dbg/ab.py $form $wl > $R/f_${wl}_$form.log 2>&1 || { tail -5 $R/f_${wl}_$form.log; exit 1; }
echo "$wl flags=$form (8192 packets, 4096 tables, 0 measured): $(cat $R/f_${wl}_$form.log)"
done
done
