#!/bin/bash
# round 6, second pass: row-split budget x where the launch starts x which stream carries the conv weight gradients; then the other
# configs at the two candidate settings.  DIAG build, alternating, one box.
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
run() { env RN_DIAG=1 $1 python tools/dbg/exp_bench.py $2 -- --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))"; }
for i in 1 2; do
  for t in 32 36 40 44; do for l in 0 1; do for c in 1 2; do echo "KB_TOTAL=$t wgrad_late=$l conv_wgrad_stream=$c : $(run RN_KB_TOTAL=$t "wgrad_late=$l conv_wgrad_stream=$c")"; done; done; done
done
for i in 1 2; do
  for r in 16 20 24 28; do echo "KB_TOTAL=40 late=0 ratio_x8=$r : $(run "RN_KB_TOTAL=40 RN_KBW_RATIO_X8=$r" "wgrad_late=0")"; done
done
for i in 1 2; do
  for s in "RN_KB_TOTAL=48|wgrad_late=1" "RN_KB_TOTAL=40|wgrad_late=1" "RN_KB_TOTAL=40|wgrad_late=0" "RN_KB_TOTAL=36|wgrad_late=0"; do
    e=${s%%|*}; k=${s##*|}
    echo "ir-fp  $e $k : $(run $e $k "--config ir-fp")"
    echo "stress $e $k : $(run $e $k "--hw 224 --batch 32 --steps 40")"
    echo "b640   $e $k : $(run $e $k "--batch 640 --steps 10 --warmup 3")"
  done
done
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
