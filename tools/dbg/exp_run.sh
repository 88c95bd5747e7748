# usage: exp_run.sh reps "<exp settings>" "<exp settings>" ...   (a setting: "wgrad_late=1 conv_wgrad_stream=2"; "-" = none)
N=$1; shift
for i in $(seq 1 $N); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
    v=$(python tools/dbg/exp_bench.py $EE -- --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))")
    echo "exp=[$E] : $v"
  done
done
