#!/bin/bash
# round 6: does a LONG weight-gradient launch (B = 640: 1.7 ms; 14 x 14 / B = 32: 0.8 ms) want another workgroup budget than the 160
# that the headline shape's step prefers?  DIAG build, alternating, one box.
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
run() { env RN_DIAG=1 $1 python tools/dbg/exp_bench.py $2 -- --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))"; }
for i in 1 2; do
  for t in 40 48 56 64; do for l in 0 1; do
    echo "b640   KB_TOTAL=$t wgrad_late=$l : $(run RN_KB_TOTAL=$t wgrad_late=$l "--batch 640 --steps 10 --warmup 3")"
    echo "stress KB_TOTAL=$t wgrad_late=$l : $(run RN_KB_TOTAL=$t wgrad_late=$l "--hw 224 --batch 32 --steps 40 --warmup 5")"
  done; done
done
for i in 1 2; do for t in 40 48 56; do echo "b256   KB_TOTAL=$t wgrad_late=0 : $(run RN_KB_TOTAL=$t wgrad_late=0 "--batch 256 --steps 20 --warmup 5")"; done; done
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
