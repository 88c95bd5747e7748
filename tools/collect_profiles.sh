#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box into gpurun_out/prof_final/ (copied to profiles/ afterwards by
# tools/publish_profiles.py).  usage (through gpurun): bash tools/collect_profiles.sh [quick]
# Everything DESIGN.md quotes comes out of THIS script (or out of an A/B log under gpurun_out/ that publish_profiles.py copies into
# profiles/rNN_ablations/): nothing a reader cannot open.
R=$PWD
OUT=$R/gpurun_out/prof_final
ABL=$OUT/ablations
rm -rf $OUT; mkdir -p $OUT $ABL
QUICK=${1:-}
# 1. the bench line (default command) and the concurrent timeline -- FIRST: right behind a rocprofv3 --pmc pass the first replays run at
# the profiler's clocks (seen once: 57 k q/s over the 20 timed steps, 92 k sustained three seconds later)
cd $R
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
python tools/step_timeline.py > $OUT/step_timeline.txt 2>&1
ls -la $OUT
# 2. the other BASELINE configs: original-sd B=4 (configs[0]), ir-fp (configs[3]) and the 14x14 / B=32 stress shape (configs[4])
python bench.py --config original-sd --no-other-modes > $OUT/bench_original_sd_b4.json 2>> $OUT/bench.err
# ... and the same model at a training-sized batch in the exact-fp32 mode and in the split-bf16 one (precision "bf16x3")
python bench.py --config original-sd --batch 64 --precision fp32 --no-cpu-baseline --no-other-modes > $OUT/bench_original_sd_b64_fp32.json 2>> $OUT/bench.err
python bench.py --config original-sd --batch 64 --precision bf16x3 --no-cpu-baseline --no-other-modes > $OUT/bench_original_sd_b64_bf16x3.json 2>> $OUT/bench.err
python tools/step_timeline.py --config original-sd --batch 4 > $OUT/step_timeline_original_sd_b4.txt 2>&1
python tools/step_timeline.py --config ir-fp > $OUT/step_timeline_ir_fp.txt 2>&1
python bench.py --config ir-fp --no-cpu-baseline > $OUT/bench_ir_fp.json 2>> $OUT/bench.err
python bench.py --hw 224 --batch 32 --steps 40 --no-cpu-baseline > $OUT/bench_stress_b32_n196.json 2>> $OUT/bench.err
# the reference's DEFAULT batch size (train.py:370): one rank, B = 640
python bench.py --batch 640 --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-parity > $OUT/bench_original_fp_b640.json 2>> $OUT/bench.err
python tools/time_small.py > $OUT/small_kernels_alone.txt 2>/dev/null
python tools/time_wgrad.py > $OUT/wgrad_alone.txt 2>/dev/null
python tools/time_fwd_f16s.py > $OUT/fwd_chain_alone.txt 2>/dev/null
python tools/time_k1.py > $OUT/k1_alone.txt 2>/dev/null
python tools/time_bwd.py > $OUT/bwd_chain_alone.txt 2>/dev/null
B=32 N_OBJ=196 python tools/time_bwd.py > $OUT/bwd_chain_alone_b32_n196.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# 3. kernel trace of the bench command (eager launches: per-kernel durations; the timed bench line itself uses the hipGraph)
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --no-graph > $OUT/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/p_kt -name "*.db" | head -1) 13 > $OUT/kernel_stats.csv
# ... and of the REPLAYED step (hipGraph): idle intervals between the kernels, node count
rocprofv3 --kernel-trace -d /tmp/p_kg -o p -- python $R/bench.py --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing > $OUT/kg.log 2>&1
python $R/tools/dbg/graph_gaps.py $(find /tmp/p_kg -name "*.db" | head -1) 10 > $OUT/graph_gaps.txt 2>&1
# 4. HBM traffic + SQ counters of the hot-path kernels, one counter set per pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/p_$c -o p -- python $R/tools/run_kernels_once.py all > $OUT/pmc_$c.log 2>&1
  python $R/tools/pmc_table.py $(find /tmp/p_$c -name "*counter_collection.csv" | head -1) > $OUT/pmc_$c.txt
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p_sq1 -o p -- python $R/tools/run_kernels_once.py all > $OUT/pmc_sq1.log 2>&1
python $R/tools/pmc_table.py $(find /tmp/p_sq1 -name "*counter_collection.csv" | head -1) > $OUT/pmc_sq1.txt
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/p_sq2 -o p -- python $R/tools/run_kernels_once.py all > $OUT/pmc_sq2.log 2>&1
python $R/tools/pmc_table.py $(find /tmp/p_sq2 -name "*counter_collection.csv" | head -1) > $OUT/pmc_sq2.txt
cd $R
# 5. the fused extraction op alone + its written bytes, the convergence runs, the sustained run's clocks
python tools/time_extract.py > $OUT/extract_alone.txt 2>/dev/null
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_xw -o p -- python $R/tools/time_extract.py once > $OUT/pmc_extract.log 2>&1; python $R/tools/pmc_table.py $(find /tmp/p_xw -name "*counter_collection.csv" | head -1) | grep -i "extract" > $OUT/pmc_extract_write.txt)
python tools/convergence.py 300 > $OUT/convergence.txt 2>/dev/null
python -c "
import json; d = json.load(open('$OUT/bench_line.json')); s = d.get('sustained') or {}
print('# rocm-smi clocks / power sampled every 0.25 s during the sustained run of the default bench line (bench.py --sustain 3)')
print(json.dumps({'value_20_steps': d['value'], 'sustained': s}, indent=1))" > $OUT/clocks.txt
# 6. every measured parity value of the round: the parity tests append to gpurun_out/parity_report.jsonl
rm -f $R/gpurun_out/parity_report.jsonl
python -m pytest tests/test_gpu_parity.py tests/test_train_driver.py -m gpu -q > $OUT/parity_tests.log 2>&1
cp $R/gpurun_out/parity_report.jsonl $OUT/parity_report.jsonl 2>/dev/null
python tools/kernel_resources.py > $OUT/kernel_resources.txt 2>/dev/null
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
if [ "$QUICK" = "quick" ]; then ls -la $OUT; exit 0; fi
# 7. does the default arithmetic train like fp32?  24 seeds x {fp32, auto, auto with 16-bit copies} x {original-fp, ir-fp} on the
# relational task with device-made batches (~3 s a run, 13 s in fp32: ~16 min), per-run lines + the summary / paired tables
rm -f $OUT/convergence_seeds.jsonl
python tools/convergence_seeds.py --seeds 24 --out $OUT/convergence_seeds.jsonl > $OUT/convergence_seeds_runs.txt 2>&1
( echo "# python tools/convergence_seeds.py --seeds 24   (one MI355X box; per-run lines: convergence_seeds.jsonl)"
  echo "# task: train.PairRelationTaskOnDevice (three squares; column / closest / same-row questions), 3000 Adam steps, lr 5e-4, B = 64, clip 50, wd 1e-4"
  echo "# modes: fp32 = per-layer fp32-MFMA kernels; auto = f16s chains + e4m3 activation copies (the default); auto16 = the same with 16-bit copies"
  python tools/convergence_seeds.py --summary-of $OUT/convergence_seeds.jsonl ) > $OUT/convergence_seeds.txt 2>&1
# 8. ablations quoted in DESIGN.md, from a DIAGNOSTICS build of the library (timing-only variants with wrong results; the product
# library is rebuilt afterwards)
RN_DIAG=1 python relationnetworks-clevr_amd/_build.py --force > $ABL/diag_build.log 2>&1
RN_DIAG=1 python tools/time_fwd_f16s.py > $ABL/fwd_chain_ablations.txt 2>/dev/null
RN_DIAG=1 python tools/time_wgrad.py > $ABL/wgrad_ablations.txt 2>/dev/null
RN_DIAG=1 python tools/time_bwd_abl.py > $ABL/bwd_chain_ablations.txt 2>/dev/null
# the weight-gradient launch alone by workgroup budget (4 x RN_KB_TOTAL workgroups; product: 40) and with the quad mapping for every job
( for t in 32 40 48 56 64; do echo "== RN_KB_TOTAL=$t (wide units for the stored jobs)"; RN_DIAG=1 RN_KB_TOTAL=$t python tools/time_wgrad.py 2>/dev/null | sed -n 1,3p; done
  for t in 40 48 64; do echo "== RN_KB_TOTAL=$t RN_KB_NO_WIDE=1 (rounds 3-5: quad units for every job)"; RN_DIAG=1 RN_KB_TOTAL=$t RN_KB_NO_WIDE=1 python tools/time_wgrad.py 2>/dev/null | sed -n 1,3p; done ) > $ABL/wgrad_alone_by_budget.txt 2>&1
python relationnetworks-clevr_amd/_build.py --force > /dev/null 2>&1
mkdir -p tools/dbg/libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/dbg/libs/reg_stream_bench tools/dbg/reg_stream_bench.hip > /dev/null 2>&1 && tools/dbg/libs/reg_stream_bench > $ABL/reg_stream_bench.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/dbg/libs/mfma_filler_bench tools/dbg/mfma_filler_bench.hip > /dev/null 2>&1 && tools/dbg/libs/mfma_filler_bench > $ABL/mfma_filler_bench.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/dbg/libs/ldsdma_offset_probe tools/dbg/ldsdma_offset_probe.hip > /dev/null 2>&1 && tools/dbg/libs/ldsdma_offset_probe > $ABL/ldsdma_offset_probe.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/dbg/libs/ovfl_probe tools/dbg/ovfl_probe.hip > /dev/null 2>&1 && tools/dbg/libs/ovfl_probe > $ABL/fp16_ovfl_probe.txt 2>&1
ls -la $OUT $ABL
