#!/bin/bash
# usage (on the GPU box): bash tools/collect_profiles.sh <tag>  -> gpurun_out/<tag>/{main,arxiv,wide,graph}/p_kernel_stats.csv,
# gpurun_out/<tag>/{fetch,write,mfma}/...counter_collection.csv.  One rocprofv3 run per pass (counters never share a
# run with each other's groups or with --stats domains beyond --kernel-trace).
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
run() { name=$1; shift; timeout -k 10 400 rocprofv3 "$@" > $OUT/$name.log 2>&1 && echo "$name ok" || { echo "$name FAILED"; tail -3 $OUT/$name.log; exit 1; }; }
run main  --kernel-trace --stats --output-format csv -d $OUT/main  -o p -- python3 $B --steps 5 --warmup 1 --no-extras
run arxiv --kernel-trace --stats --output-format csv -d $OUT/arxiv -o p -- python3 $B --steps 5 --warmup 1 --workload tabgnn-arxiv
run wide  --kernel-trace --stats --output-format csv -d $OUT/wide  -o p -- python3 $B --steps 5 --warmup 1 --workload wide64-c256
run graph --kernel-trace --stats --output-format csv -d $OUT/graph -o p -- python3 $B --workload reference-batch-graph
run fetch --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 $B --steps 5 --warmup 1 --no-extras
run write --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 $B --steps 5 --warmup 1 --no-extras
run mfma  --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -o p -- python3 $B --steps 2 --warmup 1 --no-extras
# the trace csvs are large: keep the summaries only
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls $OUT
