#!/bin/bash
# same-box A/B of an environment switch: tools/ab_env.sh VAR  -> B=8192 step and B=200 graph replay with VAR=0 / VAR=1
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do
  for v in 0 1; do
    ms=$(env $1=$v timeout -k 10 300 python bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    echo "rep $rep  $1=$v  B=8192 $ms ms/step"
  done
done
for v in 0 1; do
  env $1=$v timeout -k 10 300 python bench.py --workload reference-batch-graph 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('reference_batch',d)
print('$1=$v  B=200 replay', round(r.get('ms_per_step',0),3), 'ms/step, launches', r.get('launches_per_step'), 'eager', round(r.get('eager_ms_per_step',0),3))"
done
