#!/bin/bash
# Round-2 evidence.  $1 = tag (e.g. r02a).  Kernel trace of the SERIAL bench (non-overlapped launches: the durations
# roofline.* divides by), kernel trace of the default bench, four counter passes over the 4K frame, C5 and C4r bench lines.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_${TAG}_serial -o serial -- python3 $GRAFT_REPO_ROOT/bench.py --serial --no-cpu --steps 20 --warmup 5 > $R/bench_${TAG}_serial.json 2> $R/bench_${TAG}_serial.err || { tail -5 $R/bench_${TAG}_serial.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_${TAG} -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-count --steps 20 --warmup 5 > $R/bench_${TAG}_prof.json 2> $R/bench_${TAG}_prof.err || { tail -5 $R/bench_${TAG}_prof.err; exit 1; }
bash $GRAFT_REPO_ROOT/tools/pmc_passes.sh > $R/pmc_${TAG}.log 2>&1 || { tail -5 $R/pmc_${TAG}.log; exit 1; }
cd $GRAFT_REPO_ROOT
cp $R/pmc_summary.csv $R/${TAG}_pmc_c4.csv
timeout -k 10 400 python3 bench.py --workload c5 --steps 3 --warmup 1 > $R/bench_${TAG}_c5.json 2> $R/bench_${TAG}_c5.err || { tail -5 $R/bench_${TAG}_c5.err; exit 1; }
cut -c1-700 $R/bench_${TAG}_c5.json
timeout -k 10 300 python3 bench.py --workload c4r --steps 5 --warmup 1 --no-cpu > $R/bench_${TAG}_c4r.json 2> $R/bench_${TAG}_c4r.err || { tail -5 $R/bench_${TAG}_c4r.err; exit 1; }
cut -c1-700 $R/bench_${TAG}_c4r.json
find $R/prof_${TAG}_serial $R/prof_${TAG} -name "*stats*"
