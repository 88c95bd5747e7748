#!/usr/bin/env python3
"""Turn gpurun_out/prof_final/ (tools/collect_profiles.sh) into the tracked summaries under profiles/.
usage: python tools/publish_profiles.py [tag]   (tag defaults to r01_final)"""
import csv, os, re, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import pretty
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_final")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
dst = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))
shutil.copy(os.path.join(src, "kernel_stats.csv"), dst("kernel_stats_rocprofv3.csv"))
shutil.copy(os.path.join(src, "bench_line.json"), dst("bench_line.json"))
shutil.copy(os.path.join(src, "step_timeline.txt"), dst("step_timeline.txt"))
for extra in ("bench_original_sd_b64_fp32.json", "bench_original_sd_b64_bf16x3.json", "step_timeline_original_sd_b4.txt", "step_timeline_ir_fp.txt", "bench_ir_fp.json", "bench_stress_b32_n196.json", "bench_original_fp_b640.json", "bench_original_sd_b4.json", "small_kernels_alone.txt", "k1_alone.txt", "wgrad_alone.txt",
              "fwd_chain_alone.txt", "bwd_chain_alone.txt", "bwd_chain_alone_b32_n196.txt", "extract_alone.txt", "pmc_extract_write.txt", "convergence.txt", "convergence_pairs.txt", "convergence_pairs_ir_fp.txt", "convergence_seeds.txt", "convergence_seeds.jsonl",
              "clocks.txt", "graph_gaps.txt", "parity_report.jsonl", "kernel_resources.txt", "smoke.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), dst(extra))
OURS = re.compile(r"(rr_kernel|rr_f16s|rr_bwd|wgrad|pair_|f_phi|cn_|lstm_|emb_bwd|conv3x3s2|conv_wgrad|clip_adam|sumsq|nll_|segsum|pack_frag|debug_stamp)")
rows = list(csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))))
# eager steps in the trace = launches of the forward chain (one per step; 13 with --sustain 0, ~1400 when the sustained run is traced too)
steps = 13.0
for r_ in rows:
    if "rr_f16s_kernel" in r_["kernel"] and float(r_["calls"]) > 0:
        steps = float(r_["calls"])
with open(dst("hot_path_kernels.txt"), "w") as f:
    f.write("# this repo's HIP kernels in one training step (same rocprofv3 --kernel-trace run as %s_kernel_stats_rocprofv3.csv,\n"
            "# `bench.py --no-graph --steps 10 --warmup 3`, eager steps incl. the sustained run): launches/step, average duration.  Profiled clocks are ~5-10 %%\n"
            "# slower than the timed bench and the queues are serialised by the tracer (no stream overlap).\n" % tag)
    f.write("%-58s %9s %9s %9s\n" % ("kernel", "per step", "avg us", "us/step"))
    tot = 0.0
    for r in rows:
        name = r["kernel"]
        if name == "TOTAL" or not OURS.search(name):
            continue
        short = pretty(name)[-58:]
        per = float(r["calls"]) / steps
        us = float(r["avg_us"])
        tot += per * us
        f.write("%-58s %9.2f %9.1f %9.1f\n" % (short, per, us, per * us))
    f.write("%-58s %9s %9s %9.1f\n" % ("sum (serialised; the step overlaps three streams)", "", "", tot))
with open(dst("pmc_hbm_traffic.txt"), "w") as f:
    f.write("# HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB units, mean of 3 launches;\n"
            "# FETCH_SIZE is doubled when quoted in DESIGN.md / bench.py, as MI355X_MICROARCH.md prescribes for gfx950)\n"
            "# shape: original-fp B=64 n=64 (M=262144), forward chain in the module-default f16s arithmetic; command: rocprofv3 --pmc <counter> --kernel-trace -- python tools/run_kernels_once.py all\n")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f.write(open(os.path.join(src, "pmc_%s.txt" % c)).read())
with open(dst("pmc_sq_counters.txt"), "w") as f:
    f.write("# rocprofv3 --pmc passes over python tools/run_kernels_once.py all (mean of 3 launches per kernel); SQ_* in quad-cycles except\n"
            "# SQ_VALU_MFMA_BUSY_CYCLES (cycles = 32 x MFMA count for v_mfma_f32_32x32x16_bf16), MI355X_MICROARCH.md units\n")
    for c in ("sq1", "sq2"):
        f.write(open(os.path.join(src, "pmc_%s.txt" % c)).read())
# the A/B and ablation logs DESIGN.md quotes: the DIAG-build ablations of the collection run + whatever A/B logs of the round sit in
# gpurun_out/ (one gpurun call each: same box within a file)
abl_dst = os.path.join(ROOT, "profiles", "%s_ablations" % tag)
os.makedirs(abl_dst, exist_ok=True)
abl_src = os.path.join(src, "ablations")
if os.path.isdir(abl_src):
    for f in sorted(os.listdir(abl_src)):
        if f.endswith(".txt"):
            shutil.copy(os.path.join(abl_src, f), os.path.join(abl_dst, f))
for f in sorted(os.listdir(os.path.join(ROOT, "gpurun_out"))):
    if (f.startswith("ab_") or f.startswith("wide_")) and f.endswith(".txt"):
        shutil.copy(os.path.join(ROOT, "gpurun_out", f), os.path.join(abl_dst, f))
print("published to profiles/%s_* and profiles/%s_ablations/" % (tag, tag))
