#!/bin/bash
# round 6: in a rocprofv3 trace of the REPLAYED step the question encoder (pack -> dropout mask -> lstm_fwd, 58 us) runs BEHIND the conv
# stack, not beside it.  Is that the tracer, or what the graph executor does with the capture order?  Unprofiled A/B.
run() { env $1 python tools/dbg/exp_bench.py $2 -- --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  echo "default                 : $(run "" "")"
  echo "RN_OVERLAP_STREAMS=0    : $(run RN_OVERLAP_STREAMS=0 "")"
  echo "text_first=1            : $(run "" text_first=1)"
done
for i in 1 2; do
  echo "ir-fp default           : $(run "" "" "--config ir-fp")"
  echo "ir-fp text_first=1      : $(run "" text_first=1 "--config ir-fp")"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tf
rocprofv3 --kernel-trace -d /tmp/p_tf -o p -- python $GRAFT_REPO_ROOT/tools/dbg/exp_bench.py text_first=1 -- --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing > /dev/null 2>&1
echo "== text_first=1, replayed step"; python $GRAFT_REPO_ROOT/tools/dbg/graph_gaps.py $(find /tmp/p_tf -name "*.db" | head -1) 10 one 2>&1 | grep -v "^  *[0-9.]* us x0" | head -40
