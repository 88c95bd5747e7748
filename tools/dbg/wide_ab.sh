#!/bin/bash
# round 6: wide-unit weight gradient -- tests, then alternating A/B of the step (DIAG build: both mappings in one library)
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wgrad" 2>&1 | tail -5
python -m pytest tests/test_convergence.py -m gpu -q -k "many_trainers" 2>&1 | tail -3
python tools/time_wgrad.py 2>/dev/null | head -3
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
for r in 20 24; do echo "== ratio x8 = $r"; RN_DIAG=1 RN_KBW_RATIO_X8=$r python tools/time_wgrad.py 2>/dev/null | sed -n 3p; done
RN_DIAG=1 bash tools/ab_bench.sh "RN_KB_NO_WIDE=1" "RN_KBW_RATIO_X8=24" 4
RN_DIAG=1 bash tools/ab_bench.sh "RN_KBW_RATIO_X8=20" "RN_KBW_RATIO_X8=28" 2
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
