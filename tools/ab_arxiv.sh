#!/bin/bash
# same-box comparison of library builds on the long-row workloads: tools/ab_arxiv.sh <dir under build/ or "base"> ...
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  for v in "$@"; do
    if [ $v = base ]; then E=""; else E="TABGNN_LIB_PATH=$R/models-for-relational-multimodal-data_amd/build/$v/libtabgnn_hip.so"; fi
    for w in tabgnn-arxiv wide64-c256; do
      ms=$(env $E timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
      echo "rep $rep  $v  $w  $ms ms/step"
    done
  done
done
