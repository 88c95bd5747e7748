for i in 1 2 3; do
  for tag in new old; do
    if [ $tag = new ]; then L=relationnetworks-clevr_amd/librn_hip.so; else L=tools/dbg/libs/librn_oldqs.so; fi
    v=$(python tools/dbg/bench_with_lib.py $L --config ir-fp --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %.4f %.1f' % (d['value'], d['ms_per_step'], d['sustained']['value']))")
    echo "$tag $v"
  done
done
