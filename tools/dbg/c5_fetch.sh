#!/bin/bash
# HBM bytes fetched by the kernels of two C5 frames (FETCH_SIZE, one counter pass; KB, before the guide's x2 for wide reads)
R=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_c5 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/dbg/serial_frames.py c5 1 0 > $R/pmc_c5.log 2>&1 || { tail -5 $R/pmc_c5.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/pmc_csv_summary.py $(find $R/pmc_c5 -name "*counter_collection.csv") | cut -c1-160
tail -1 $R/pmc_c5.log
