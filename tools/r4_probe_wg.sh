#!/bin/bash
# A/B of the fused encoder kernels at 2 workgroups per CU (shipped) vs 1 (512 registers per wave): per-kernel times by rocprofv3
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4_wg; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base wg1; do
  if [ $v = base ]; then unset TABGNN_LIB_PATH; else export TABGNN_LIB_PATH=$R/models-for-relational-multimodal-data_amd/build/$v/libtabgnn_hip.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o p -- python3 $R/tools/encoder_probe3.py > $OUT/$v.log 2>&1
  tail -1 $OUT/$v.log
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/$v/**/p_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Calls']:>4} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:70]}")
PY
done
