mkdir -p gpurun_out/r02b
python -m pytest tests -m gpu -q > gpurun_out/r02b/test_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b/test_gpu.log; tail -15 gpurun_out/r02b/test_gpu.log
for late in 0 1 2; do for zs in 64 48 32; do
  echo "LATE=$late ZS=$zs" >> gpurun_out/r02b/sweep.txt
  RN_WGRAD_LATE=$late RN_WGRAD_ZS=$zs python bench.py --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r02b/sweep.txt
done; done
cat gpurun_out/r02b/sweep.txt
