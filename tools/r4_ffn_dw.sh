#!/bin/bash
# parity of the fused layer (incl. the DW feed-forward backward) + per-kernel timing of the probe
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4_ffn_dw; mkdir -p $OUT
cd $R && timeout -k 10 600 python -m pytest tests/test_gpu_encoder_fused.py -x -q > $OUT/test.log 2>&1; echo "test rc=$?"; tail -5 $OUT/test.log
cd /tmp && export TMPDIR=/tmp
for v in dw nodw; do
  if [ $v = nodw ]; then export TABGNN_NO_DW_FFN=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o p -- python3 $R/tools/encoder_probe3.py > $OUT/$v.log 2>&1
  tail -1 $OUT/$v.log
  python3 - <<PY
import csv,glob
f=glob.glob("$OUT/$v/**/p_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Calls']:>4} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:70]}")
PY
done
