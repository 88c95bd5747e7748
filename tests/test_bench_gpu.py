"""GPU: bench.py's JSON line -- the fields the driver and the judge read -- on short runs of two BASELINE.json configs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench(*flags):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--sustain", "0", "--no-other-modes"] + list(flags)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_reports_the_dominant_kernels_roofline():
    """`roofline` is the kernel with the LONGEST launch among g_theta's, not the one that scores best; the
    forward chain (north_star's GEMM chain) sits beside it as `gemm_chain`; the cross-checks the judge runs hold."""
    d = _bench("--no-cpu-baseline")
    r = d["roofline"]
    per = {k: v["ms"] for k, v in r["kernels"].items()}
    assert r["kernel_key"] == max(per, key=per.get) and abs(r["ms_per_launch"] - per[r["kernel_key"]]) < 1e-12
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["ms_per_launch"] < d["ms_per_step"]                       # a kernel of the step is shorter than the step
    g = r["gemm_chain"]
    assert "g_chain_rr_f16s_kernel" in g["kernel"] and 0.0 < g["frac"] < 1.0 and 0.0 < g["frac_executed"] < g["frac"]
    assert d["dtype"] == "f16s" and d["parity"]["meets_1e-3"] is True and d["vs_baseline"] is None
    assert "value_definition" in d and d["config"]["workload"].startswith("original-fp train step")
    # an entry's flops are those of exactly what its launches compute.  The weight-gradient launch is dW_1..3
    # (3 x 2 M 256^2); layer 0's weight gradient -- another launch on another stream -- is an entry of its own (2 M 180 256)
    M = 64 * 64 * 64
    assert r["kernel_key"] == "g_wgrad" and r["algorithmic_flops_per_launch"] == 3 * 2 * M * 256 ** 2
    assert r["kernels"]["g_wgrad0"]["algorithmic_flops"] == 2 * M * 180 * 256 and r["kernels"]["g_dgrad"]["algorithmic_flops"] == 3 * 2 * M * 256 ** 2
    assert r["kernels"]["g_fwd"]["algorithmic_flops"] == 2 * M * (180 + 3 * 256) * 256
    # the dominant launch on its own beside its in-step bracket; the MFMA-stream probe is a diagnostic, never a denominator
    assert 0.0 < r["frac_alone"] < 1.0 and abs(r["frac_alone"] - r["algorithmic_flops_per_launch"] / (r["ms_alone"] * 1e-3) / 1e12 / r["peak"]) < 1e-9
    assert "frac_of_sustained" not in r and "sustained_mfma" not in r
    pr = r["diagnostics"]["mfma_stream_probe"]
    assert 0 < pr["bf16"]["tflops"] <= pr["bf16_zero_operands"]["tflops"] * 1.05 and pr["bf16_zero_operands"]["tflops"] < 2600
    for kk, v in r["kernels"].items():                                 # traffic is the tracked PMC summary's or null, never invented
        assert v["traffic"] is None or (v["traffic_source"].startswith("profiles/") and v["traffic"] > 0.5 * v["algorithmic_hbm_bytes"])


def test_state_description_config_runs_with_both_cpu_baselines():
    """BASELINE.json configs[0] (original-sd, B=4, 12-object state descriptions; /root/reference/config.json:17-30): the bench takes
    (B, 12, 7) object rows instead of images, runs the per-layer fp32 kernels, checks G-sd4 live and times the CPU restatement of
    the same config (`cpu_baseline`)."""
    d = _bench("--config", "original-sd")
    assert d["config"]["global_batch"] == 4 and "state descriptions" in d["config"]["workload"] and d["dtype"] == "fp32"
    assert d["value"] > 0 and d["parity"]["G-sd4"]["log_prob_rel_err"] < 1e-4 and d["parity"]["meets_1e-3"] is True
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and "original-sd" in c["sample"] and "B=4" in c["sample"] and "cpu_baseline_sd4" not in d
    assert d["roofline"]["peak"] == 157.3 and "pair_build_k1" not in d
