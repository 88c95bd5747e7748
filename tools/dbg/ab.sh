#!/bin/bash
# A/B of two trees on ONE box: tools/dbg/ab.sh <dirA> <dirB> [reps] [extra bench args] -- alternating runs of the default bench line
A=$1; B=$2; N=${3:-3}; shift 3
for i in $(seq 1 $N); do
  for D in $A $B; do
    v=$(cd $D && python bench.py --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))")
    echo "$D : $v"
  done
done
