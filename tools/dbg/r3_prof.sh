#!/bin/bash
# round 3: cell-table grid sizes, serial kernel trace, bench line with executed-work counters
R=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
for g in 256 1024; do
RTX_CELL_G=$g timeout -k 10 200 python -u tools/dbg/ab.py 0 c4 > $R/p_ab_g$g.log 2>&1 || { tail -5 $R/p_ab_g$g.log; exit 1; }
echo "G=$g: $(cat $R/p_ab_g$g.log)"
done
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu > $R/p_bench.json 2> $R/p_bench.err || { tail -5 $R/p_bench.err; exit 1; }
cut -c1-400 $R/p_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_serial -o ser -- python3 $GRAFT_REPO_ROOT/bench.py --serial --steps 5 --warmup 2 --no-cpu --no-count > $R/p_bench_serial.log 2>&1 || { tail -5 $R/p_bench_serial.log; exit 1; }
find $R/prof_serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/p_serial_kernel_stats.csv
head -30 $R/p_serial_kernel_stats.csv | cut -c1-200
