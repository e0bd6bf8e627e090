#!/bin/bash
# usage: pmc_passes.sh [c4 | c4r | c5].  Four counter passes over the benchmark frame (two SQ passes, FETCH_SIZE, WRITE_SIZE: the two together exceed the TCC counters of one run), summed per kernel
# into gpurun_out/pmc_summary.csv.  One pass per rocprofv3 run (the tool does not split counters over passes).
R=$GRAFT_REPO_ROOT/gpurun_out
WL=${1:-c4}
rm -rf $R/pmc1 $R/pmc2 $R/pmc3 $R/pmc4
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/pmc1 -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dbg/one_frame.py $WL > $R/pmc1.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/pmc2 -o p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dbg/one_frame.py $WL > $R/pmc2.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $R/pmc3 -o p3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dbg/one_frame.py $WL > $R/pmc3.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $R/pmc4 -o p4 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dbg/one_frame.py $WL > $R/pmc4.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_csv_summary.py $(find $R/pmc1 $R/pmc2 $R/pmc3 $R/pmc4 -name "*counter_collection.csv") > $R/pmc_summary.csv
cat $R/pmc_summary.csv
