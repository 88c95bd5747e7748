#!/bin/bash
# round 6: how deep the weight gradient's request ring has to be.  Its 5 x 24 KB in flight per CU (19 MB on 160 CUs) sit in
# front of every memory request of the latency-bound kernels beside it; the launch's own rate barely depends on the depth.
for L in - ring5 ring4 ring3; do
  if [ "$L" = "-" ]; then LIB=relationnetworks-clevr_amd/librn_hip.so; else LIB=tools/dbg/libs/librn_$L.so; fi
  echo "== $L alone: $(RN_LIB=$LIB python tools/time_wgrad.py 2>/dev/null | sed -n 3p)"
done
bash tools/dbg/ab_libs.sh "- ring5 ring4 ring3" 3
bash tools/dbg/ab_libs.sh "- ring4 ring3" 2 --config ir-fp
bash tools/dbg/ab_libs.sh "- ring4 ring3" 2 --hw 224 --batch 32 --steps 40
