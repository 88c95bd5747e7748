#!/bin/bash
# round 6: at B = 640 the weight gradient takes 1.7 ms beside the conv stack's backward and 1.37 ms alone on 160 workgroups.
# Is the overlap still worth it there?  Serial (RN_NO_WGRAD_OVERLAP=1: everything on one stream) with bigger budgets vs the product.
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
run() { env RN_DIAG=1 $1 python bench.py --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))"; }
for i in 1 2; do
  echo "b640 product                       : $(run "" "--batch 640 --steps 10 --warmup 3")"
  for t in 48 56 64; do echo "b640 serial KB_TOTAL=$t            : $(run "RN_NO_WGRAD_OVERLAP=1 RN_KB_TOTAL=$t" "--batch 640 --steps 10 --warmup 3")"; done
  echo "stress product                     : $(run "" "--hw 224 --batch 32 --steps 40")"
  for t in 56 64; do echo "stress serial KB_TOTAL=$t          : $(run "RN_NO_WGRAD_OVERLAP=1 RN_KB_TOTAL=$t" "--hw 224 --batch 32 --steps 40")"; done
done
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
