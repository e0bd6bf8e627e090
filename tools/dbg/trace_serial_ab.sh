#!/bin/bash
# per-launch times of a SERIAL 4K frame (nothing overlaps) for the default library and each library given
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in - "$@"; do
  tag=$(basename "$L" .so); [ "$L" = "-" ] && tag=default
  if [ "$L" = "-" ]; then unset RTX_LIB; else export RTX_LIB=$GRAFT_REPO_ROOT/$L; fi
  rm -rf $R/profs_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/profs_$tag -o t -- python3 $GRAFT_REPO_ROOT/tools/dbg/serial_frames.py ${WL:-c4} > $R/serial_$tag.log 2>&1 || exit 1
  python3 $GRAFT_REPO_ROOT/tools/trace_last_frame.py $R/profs_$tag/t_results.db v > $R/traces_$tag.txt
  echo "--- $tag"; grep "k_shadow_pass\|k_shadow_affine\|k_closest_lanes\|frame span" $R/traces_$tag.txt | head -24
done
