#!/bin/bash
# round 6: the new defaults (160 workgroups, early launch for the non-injected models) -- GPU suite + the bench lines of four configs
python -m pytest tests -m gpu -q 2>&1 | tail -6
python bench.py --no-cpu-baseline > gpurun_out/bench_mid2.json 2> gpurun_out/bench_mid2.err
python bench.py --config ir-fp --no-cpu-baseline --no-other-modes > gpurun_out/bench_irfp_mid2.json 2>> gpurun_out/bench_mid2.err
python bench.py --hw 224 --batch 32 --steps 40 --no-cpu-baseline --no-other-modes > gpurun_out/bench_stress_mid2.json 2>> gpurun_out/bench_mid2.err
python bench.py --batch 640 --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-parity > gpurun_out/bench_b640_mid2.json 2>> gpurun_out/bench_mid2.err
python - <<'PY'
import json
for f in ("bench_mid2", "bench_irfp_mid2", "bench_stress_mid2", "bench_b640_mid2"):
    d = json.load(open("gpurun_out/%s.json" % f)); r = d["roofline"]
    print(f, "%.1f q/s %.4f ms sustained %s" % (d["value"], d["ms_per_step"], (d.get("sustained") or {}).get("value")), {k: (round(v["ms"] * 1e3, 1), round(v["frac"], 3)) for k, v in r["kernels"].items()},
          "alone", r.get("ms_alone"), r.get("frac_alone"), "all_g", round(r["all_g_theta"]["frac"], 3), "k1", (d.get("pair_build_k1") or {}).get("frac"), "parity", (d.get("parity") or {}).get("meets_1e-3"))
PY
python tools/step_timeline.py > gpurun_out/timeline_mid2.txt 2>&1
