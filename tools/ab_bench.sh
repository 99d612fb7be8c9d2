#!/bin/bash
# same-box A/B of the default bench step: tools/ab_bench.sh "<env assignments of variant A>" "<... of variant B>" [bench args]
# e.g. tools/ab_bench.sh "" "TABGNN_NO_DW_FFN=1"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r4_ab; mkdir -p $OUT
A="$1"; B="$2"; shift 2
cd $R
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    ms=$(env $E timeout -k 10 300 python bench.py --no-extras --steps 20 --warmup 5 "$@" 2>$OUT/err_$v.log | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    echo "rep $rep  $v [$E]  $ms ms/step"
  done
done
