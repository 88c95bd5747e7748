#!/bin/bash
# round 6: the wide-unit weight gradient -- tests, alone timings, ratio sweep (DIAG build), step A/B against the quad-only mapping
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "wgrad" 2>&1 | tail -5
python tools/time_wgrad.py 2>/dev/null | head -3
python bench.py --no-cpu-baseline --no-other-modes --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product: %.1f q/s %.4f ms' % (d['value'], d['ms_per_step']), {k: round(v['ms']*1e3,1) for k,v in d['roofline']['kernels'].items()})"
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
for r in 16 20 24 28 32; do
  echo "== ratio x8 = $r"; RN_DIAG=1 RN_KBW_RATIO_X8=$r python tools/time_wgrad.py 2>/dev/null | sed -n 3p
  RN_DIAG=1 RN_KBW_RATIO_X8=$r python bench.py --no-cpu-baseline --no-other-modes --no-parity --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  step: %.1f q/s %.4f ms' % (d['value'], d['ms_per_step']))"
done
echo "== quad only"; RN_DIAG=1 RN_KB_NO_WIDE=1 python tools/time_wgrad.py 2>/dev/null | sed -n 3p
RN_DIAG=1 RN_KB_NO_WIDE=1 python bench.py --no-cpu-baseline --no-other-modes --no-parity --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  step: %.1f q/s %.4f ms' % (d['value'], d['ms_per_step']))"
RN_DIAG=1 python tools/time_wgrad.py 2>/dev/null | grep -i "three jobs"
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
