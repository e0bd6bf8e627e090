#!/bin/bash
# kernel timeline of the last 4K frame for the default library and for each library given (paths relative to the repo root)
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in - "$@"; do
  tag=$(basename "$L" .so); [ "$L" = "-" ] && tag=default
  if [ "$L" = "-" ]; then unset RTX_LIB; else export RTX_LIB=$GRAFT_REPO_ROOT/$L; fi
  rm -rf $R/prof_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/prof_$tag -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-count $BENCH_ARGS > $R/bench_$tag.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/tools/trace_last_frame.py $R/prof_$tag/t_results.db v > $R/trace_$tag.txt
  echo "--- $tag"; cut -c1-120 $R/bench_$tag.log | tail -1; tail -16 $R/trace_$tag.txt
done
