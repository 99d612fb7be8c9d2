#!/bin/bash
# same-box comparison of library builds by name: tools/ab_libs.sh [-r REPS] <name|main> ...   (tabgnn_amd/libtabgnn_hip_<name>.so)
R=$GRAFT_REPO_ROOT; cd $R
REPS=2; if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
for rep in $(seq $REPS); do
  for v in "$@"; do
    if [ $v = main ]; then E="X=1"; else E="TABGNN_LIB_PATH=$R/models-for-relational-multimodal-data_amd/tabgnn_amd/libtabgnn_hip_$v.so"; fi
    ms=$(env $E timeout -k 10 300 python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    echo "rep $rep  $v  $ms ms/step"
  done
done
