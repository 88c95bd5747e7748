#!/bin/bash
# round 6: the clone of the question encoder's final hidden state is a memcpy node of the captured step in front of the join with
# the conv stack.  A/B of returning the view instead (functional.SCHED["lstm_hn_view"] = 1), and the replayed step's kernel order.
for i in 1 2 3; do for v in 0 1; do
  r=$(python tools/dbg/exp_bench.py lstm_hn_view=$v -- --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))")
  echo "lstm_hn_view=$v : $r"
done; done
for v in 0 1; do r=$(python tools/dbg/exp_bench.py lstm_hn_view=$v -- --config ir-fp --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))"); echo "ir-fp lstm_hn_view=$v : $r"; done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/p_kg$v
  rocprofv3 --kernel-trace -d /tmp/p_kg$v -o p -- python $GRAFT_REPO_ROOT/tools/dbg/exp_bench.py lstm_hn_view=$v -- --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing > /dev/null 2>&1
  echo "== gaps, lstm_hn_view=$v"; python $GRAFT_REPO_ROOT/tools/dbg/graph_gaps.py $(find /tmp/p_kg$v -name "*.db" | head -1) 10 one 2>&1 | head -75
done
