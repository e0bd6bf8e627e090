#!/bin/bash
# last serial frame's launches one by one (kernel trace), and context creation phase by phase (library built with EXTRA=-DRTX_AB_KNOBS)
R=$GRAFT_REPO_ROOT/gpurun_out
W=${1:-c4}
N=${2:-1}
S=${3:-1}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/prof_tl -o tl -- python3 $GRAFT_REPO_ROOT/tools/dbg/serial_frames.py $W $N $S > $R/tl_$W.log 2>&1 || { tail -5 $R/tl_$W.log; exit 1; }
grep -a "rays per" $R/tl_$W.log
python3 - $R/prof_tl <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.3f %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r["Kernel_Name"].split("(")[0].replace("rtx::", "")[:50]))
PY
cd $GRAFT_REPO_ROOT
[ "$N" = 1 ] || exit 0
RTX_VERBOSE=1 timeout -k 10 200 python -u tools/dbg/r3_create.py $W > $R/create_$W.log 2>&1 || { tail -5 $R/create_$W.log; exit 1; }
cat $R/create_$W.log
