#!/bin/bash
# A/B the train step under two environments, alternating runs: tools/ab_bench.sh "ENV_A=.." "ENV_B=.." [reps]
A="$1"; B="$2"; N=${3:-4}
for i in $(seq 1 $N); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    v=$(env $E python bench.py --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))")
    echo "$tag [$E] $v"
  done
done
