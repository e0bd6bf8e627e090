#!/bin/bash
# round 2, first GPU call: parity suite (incl. the new C5 / bundled-scene tests), the no-alloc entry point's test,
# bench line with the executed-work roofline, multi-rank launcher rehearsal
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -u -m pytest tests/ -m gpu -x -q --timeout 300 --timeout-method thread --durations=8 > $R/t_r02a.log 2>&1
rc=$?
tail -15 $R/t_r02a.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python -u -m pytest tests/wip_no_alloc.py -m gpu -x -v --timeout 120 --timeout-method thread > $R/t_noalloc.log 2>&1
rc=$?
tail -8 $R/t_noalloc.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $R/bench_r02a.json 2> $R/bench_r02a.err || { tail -5 $R/bench_r02a.err; exit 1; }
cut -c1-1500 $R/bench_r02a.json
RTX_BENCH_REHEARSAL=1 timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --no-count > $R/bench_reh2.json 2> $R/bench_reh2.err || { tail -5 $R/bench_reh2.err; exit 1; }
cut -c1-400 $R/bench_reh2.json
