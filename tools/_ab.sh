# usage: bash tools/_ab.sh ENVVAR   -> bench with ENVVAR=0/1 alternating, same box
for i in 1 2; do for m in 0 1; do
env $1=$m python bench.py --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$m', d['value'], d['ms_per_step'])"
done; done
