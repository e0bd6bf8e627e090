#!/bin/bash
# round 3: (light, cell) table kernels -- smoke, parity subset, fuzz, A/B timing (library built with EXTRA=-DRTX_AB_KNOBS)
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > $R/c_smoke.log 2>&1 || { tail -5 $R/c_smoke.log; exit 1; }
tail -1 $R/c_smoke.log
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > $R/c_parity.log 2>&1 || { tail -30 $R/c_parity.log; exit 1; }
tail -2 $R/c_parity.log
timeout -k 10 300 python -u tools/dbg/fuzz_exact.py 6 > $R/c_fuzz.log 2>&1 || { tail -8 $R/c_fuzz.log; exit 1; }
tail -3 $R/c_fuzz.log
RTX_SHADOW_DIRECT=0 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "culling or random or headline" > $R/c_parity_q.log 2>&1 || { tail -30 $R/c_parity_q.log; exit 1; }
tail -2 $R/c_parity_q.log
RTX_CELLS=0 timeout -k 10 200 python -u tools/dbg/ab.py 0 c4 > $R/c_ab0.log 2>&1 || { tail -5 $R/c_ab0.log; exit 1; }
echo "no tables: $(cat $R/c_ab0.log)"
RTX_SHADOW_DIRECT=0 timeout -k 10 200 python -u tools/dbg/ab.py 0 c4 > $R/c_ab1.log 2>&1 || { tail -5 $R/c_ab1.log; exit 1; }
echo "tables, sorted queue: $(cat $R/c_ab1.log)"
timeout -k 10 200 python -u tools/dbg/ab.py 0 c4 > $R/c_ab2.log 2>&1 || { tail -5 $R/c_ab2.log; exit 1; }
echo "tables, direct: $(cat $R/c_ab2.log)"
