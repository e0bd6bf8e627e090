R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/prof_small -o rs -- python3 $GRAFT_REPO_ROOT/bench.py --height 270 --steps 3 --warmup 1 --no-cpu --no-count --inflight 1 > $R/bench_rs.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/trace_last_frame.py $R/prof_small/rs_results.db v > $R/trace_small.txt
tail -25 $R/trace_small.txt
