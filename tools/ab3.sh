#!/bin/bash
# same-box comparison of N library builds: tools/ab3.sh <dir under build/ or "base"> ...
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  for v in "$@"; do
    if [ $v = base ]; then E=""; else E="TABGNN_LIB_PATH=$R/models-for-relational-multimodal-data_amd/build/$v/libtabgnn_hip.so"; fi
    ms=$(env $E timeout -k 10 300 python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    echo "rep $rep  $v  $ms ms/step"
  done
done
