#!/bin/bash
# round 6: with the wide units the weight gradient is no longer the step's tail -- re-sweep how much of the chip it gets
# (RN_KB_TOTAL: row-split budget, workgroups = 4 x that) and where it starts.  DIAG build, alternating runs, one box.
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
run() { env RN_DIAG=1 $1 python tools/dbg/exp_bench.py $2 -- --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  for t in 32 36 40 44 48 52 56 64; do echo "KB_TOTAL=$t : $(run RN_KB_TOTAL=$t)"; done
done
for i in 1 2; do
  for t in 40 48; do for e in "wgrad_late=0" "wgrad_late=1" "conv_wgrad_stream=0" "conv_wgrad_stream=1"; do echo "KB_TOTAL=$t $e : $(run RN_KB_TOTAL=$t $e)"; done; done
done
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
