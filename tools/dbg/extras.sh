R=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 600 python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu --no-count > $R/bench_c5.json 2> $R/bench_c5.err || exit 1
cut -c1-260 $R/bench_c5.json
timeout -k 10 100 python tools/dbg/rank_time.py || exit 1
timeout -k 10 100 python tools/dbg/ablate.py
