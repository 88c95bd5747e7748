#!/bin/bash
# round 6 mid-round check: whole GPU suite, K1 flat walk A/B (DIAG build), bench lines
python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python tools/time_k1.py 2>/dev/null
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
for i in 1 2; do for f in 0 1; do echo "RN_K1_FLAT=$f"; RN_DIAG=1 RN_K1_FLAT=$f python tools/time_k1.py 2>/dev/null | head -3; done; done
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
python bench.py --no-cpu-baseline > gpurun_out/bench_mid.json 2> gpurun_out/bench_mid.err; tail -3 gpurun_out/bench_mid.err
python bench.py --batch 640 --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-parity > gpurun_out/bench_b640_mid.json 2>> gpurun_out/bench_mid.err
python - <<'PY'
import json
for f in ("bench_mid", "bench_b640_mid"):
    d = json.load(open("gpurun_out/%s.json" % f)); r = d["roofline"]
    print(f, "%.1f q/s %.4f ms" % (d["value"], d["ms_per_step"]), {k: (round(v["ms"] * 1e3, 1), round(v["frac"], 3)) for k, v in r["kernels"].items()},
          "alone", r.get("ms_alone"), r.get("frac_alone"), "all_g", round(r["all_g_theta"]["frac"], 3), d.get("pair_build_k1", {}).get("frac"))
PY
