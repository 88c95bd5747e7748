python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r3_t4.log
tail -12 gpurun_out/r3_t4.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes > gpurun_out/r3_bench2.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench2.json'))
print(d['value'], d['ms_per_step'])
r=d['roofline']
print({k:(round(v['ms'],4),round(v['frac'],3),round(v['ms_serial'],4)) for k,v in r['kernels'].items()})
print(r['all_g_theta']['frac'], r['all_g_theta_serial']['frac'])
print(r['breakdown_ms_per_step'])
print(d['parity'])
PY
