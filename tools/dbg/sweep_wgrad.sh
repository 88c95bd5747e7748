#!/bin/bash
# A/B sweep on ONE box: weight-gradient row splits (variant libraries) x placement of the layer-0 weight gradient.
# usage: tools/dbg/sweep_wgrad.sh "<libs: - = product lib>" "<RN_WGRAD0_STREAM values>" [reps]
LIBS=${1:-"- 54 60 63"}; W0=${2:-"0 1 2"}; N=${3:-2}
for i in $(seq 1 $N); do
  for L in $LIBS; do
    for W in $W0; do
      if [ "$L" = "-" ]; then LIB=relationnetworks-clevr_amd/librn_hip.so; else LIB=tools/dbg/libs/librn_kb$L.so; fi
      v=$(RN_WGRAD0_STREAM=$W python tools/dbg/bench_with_lib.py $LIB --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))")
      echo "splits=$L wgrad0_stream=$W : $v"
    done
  done
done
