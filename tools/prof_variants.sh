#!/bin/bash
# per-kernel time of one kernel family under several library builds, same box:
# tools/prof_variants.sh <kernel substring> <dir under build/ or "base"> ...
R=$GRAFT_REPO_ROOT; K=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$R/gpurun_out/pv_$v; rm -rf $OUT; mkdir -p $OUT
  if [ $v = base ]; then unset TABGNN_LIB_PATH; else export TABGNN_LIB_PATH=$R/models-for-relational-multimodal-data_amd/build/$v/libtabgnn_hip.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-extras > $OUT/log.txt 2>&1
  python3 - "$OUT" "$K" "$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/p_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"]:
        print(f"{sys.argv[3]:8s} {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us  max {float(r['MaxNs'])/1e3:8.1f} us")
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
