for i in 1 2; do
for E in "wgrad_late=0 conv_wgrad_stream=0" "wgrad_late=1 conv_wgrad_stream=2"; do
  for C in "--config ir-fp" "--hw 224 --batch 32 --steps 20"; do
    v=$(python tools/dbg/exp_bench.py $E -- --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 100 --warmup 10 $C 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f %.4f' % (d['value'], d['ms_per_step']))")
    echo "exp=[$E] cfg=[$C] : $v"
  done
done
done
