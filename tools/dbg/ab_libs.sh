#!/bin/bash
# A/B of library variants on ONE box: tools/dbg/ab_libs.sh "<libs: - = product lib, else tools/dbg/libs/librn_<name>.so>" [reps] [bench args]
LIBS=$1; N=${2:-2}; shift 2
for i in $(seq 1 $N); do
  for L in $LIBS; do
    if [ "$L" = "-" ]; then LIB=relationnetworks-clevr_amd/librn_hip.so; else LIB=tools/dbg/libs/librn_$L.so; fi
    v=$(python tools/dbg/bench_with_lib.py $LIB --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))")
    echo "lib=$L : $v"
  done
done
