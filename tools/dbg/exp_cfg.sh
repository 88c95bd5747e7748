# usage: exp_cfg.sh reps "<bench args>" "<exp settings>" "<exp settings>" ...
N=$1; CFG=$2; shift 2
for i in $(seq 1 $N); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
    v=$(python tools/dbg/exp_bench.py $EE -- --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 100 --warmup 10 $CFG 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))")
    echo "exp=[$E] cfg=[$CFG] : $v"
  done
done
