#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/main -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-extras > $OUT/main.log 2>&1; echo rc=$?
find $OUT -name "*kernel_trace.csv" -size +20M -delete
tail -2 $OUT/main.log | cut -c1-300
