#!/bin/bash
# per-kernel times of perf_survey.py scenes (kernel trace): survey_prof.sh name ...
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_sv_$n -o sv -- python3 $GRAFT_REPO_ROOT/tools/dbg/perf_survey.py $n > $R/sv_$n.log 2>&1 || { tail -5 $R/sv_$n.log; exit 1; }
grep -a "frame" $R/sv_$n.log | cut -c1-200
python3 - $R/prof_sv_$n <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("  %-60s calls %5s  total %9.3f ms  avg %9.3f ms" % (r["Name"].split("(")[0].replace("rtx::", "").replace("void ", "")[:60], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
done
