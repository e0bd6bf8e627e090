R=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 100 python tools/dbg/rank_time.py || exit 1
bash tools/dbg/trace_both.sh > $R/tb.log 2>&1 || exit 1
grep -E "k_closest|k_shadow_pass" $R/trace_4k.txt | head -12
grep -E "k_closest|k_shadow_pass" $R/trace_rank.txt | head -12
