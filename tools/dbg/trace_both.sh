# kernel traces of the 4K frame and of rank 4's share at N = 8 (one GPU), last frame's per-kernel totals
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/prof_4k -o t4k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-count > $R/bench_t4k.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/prof_rank -o trank -- python3 $GRAFT_REPO_ROOT/tools/dbg/one_rank.py > $R/one_rank.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/trace_last_frame.py $R/prof_4k/t4k_results.db v > $R/trace_4k.txt
python3 tools/trace_last_frame.py $R/prof_rank/trank_results.db v > $R/trace_rank.txt
cat $R/bench_t4k.log | cut -c1-200
echo "--- 4K"; tail -18 $R/trace_4k.txt
echo "--- rank 4 of 8"; tail -18 $R/trace_rank.txt
