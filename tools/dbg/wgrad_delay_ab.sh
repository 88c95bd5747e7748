#!/bin/bash
# round 6: the early weight-gradient launch starts together with the partial sums and dx / dq -- two short, latency-bound kernels ON the
# critical path that run 1.3-1.8x slower beside its stream.  Does it pay to let the launch idle for its first microseconds?
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
run() { env RN_DIAG=1 $1 python bench.py --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do for d in 0 10 20 30 45; do echo "delay_us=$d : $(run RN_KB_DELAY_US=$d)"; done; done
for i in 1 2; do for d in 0 20 40; do echo "stress delay_us=$d : $(run RN_KB_DELAY_US=$d "--hw 224 --batch 32 --steps 40")"; done; done
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
