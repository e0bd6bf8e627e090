# parity subset + per-rank timing + traces
R=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $R/t.log 2>&1; rc=$?
tail -3 $R/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python tools/dbg/rank_time.py || exit 1
bash tools/dbg/trace_both.sh > $R/tb.log 2>&1 || exit 1
grep -E "k_closest|k_shadow_pass" $R/trace_4k.txt | head -12
grep -E "k_closest|k_shadow_pass" $R/trace_rank.txt | head -12
