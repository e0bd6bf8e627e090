#!/bin/bash
# Round-3 evidence.  $1 = tag (e.g. r03a).  The -m gpu suite, the driver-style bench line, kernel traces of the SERIAL bench
# (non-overlapped launches: the durations roofline.* divides by) and of the default bench, four counter passes over the 4K
# frame, and for C5 and C4r a bench line plus the kernel stats of its serial run.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/ -m gpu -x -q --timeout 300 --timeout-method thread > $R/${TAG}_tests.log 2>&1 || { tail -30 $R/${TAG}_tests.log; exit 1; }
tail -2 $R/${TAG}_tests.log
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/${TAG}_bench_c4.json 2> $R/${TAG}_bench_c4.err || { tail -5 $R/${TAG}_bench_c4.err; exit 1; }
cut -c1-300 $R/${TAG}_bench_c4.json
cd /tmp && export TMPDIR=/tmp
for wl in c4 c4r c5; do
steps=20; [ $wl = c5 ] && steps=3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_${TAG}_${wl}_serial -o serial -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --serial --no-cpu --steps $steps --warmup 2 > $R/${TAG}_bench_${wl}_serial.json 2> $R/${TAG}_bench_${wl}_serial.err || { tail -5 $R/${TAG}_bench_${wl}_serial.err; exit 1; }
cp $(find $R/prof_${TAG}_${wl}_serial -name "*kernel_stats.csv" | head -1) $R/${TAG}_bench_${wl}_serial_kernel_stats.csv
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_${TAG} -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-count --steps 20 --warmup 5 > $R/${TAG}_bench_c4_prof.json 2> $R/${TAG}_bench_c4_prof.err || { tail -5 $R/${TAG}_bench_c4_prof.err; exit 1; }
cp $(find $R/prof_${TAG} -name "*kernel_stats.csv" | head -1) $R/${TAG}_bench_c4_kernel_stats.csv
bash $GRAFT_REPO_ROOT/tools/pmc_passes.sh > $R/${TAG}_pmc.log 2>&1 || { tail -5 $R/${TAG}_pmc.log; exit 1; }
cd $GRAFT_REPO_ROOT
cp $R/pmc_summary.csv $R/${TAG}_pmc_c4.csv
timeout -k 10 500 python3 bench.py --workload c5 --steps 3 --warmup 1 > $R/${TAG}_bench_c5.json 2> $R/${TAG}_bench_c5.err || { tail -5 $R/${TAG}_bench_c5.err; exit 1; }
cut -c1-300 $R/${TAG}_bench_c5.json
timeout -k 10 300 python3 bench.py --workload c4r --steps 10 --warmup 3 --no-cpu > $R/${TAG}_bench_c4r.json 2> $R/${TAG}_bench_c4r.err || { tail -5 $R/${TAG}_bench_c4r.err; exit 1; }
cut -c1-300 $R/${TAG}_bench_c4r.json
