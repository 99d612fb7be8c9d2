#!/bin/bash
# per-kernel times of the column-transformer layer at the bench's big shape (tools/encoder_probe3.py under rocprofv3)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/enc_prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o p -- python3 $R/tools/encoder_probe3.py > $OUT/log 2>&1
tail -1 $OUT/log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/p/**/p_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"{r['Calls']:>4} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:70]}")
PY
