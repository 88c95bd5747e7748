#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box into gpurun_out/prof_final/ (copied to profiles/ afterwards).
# usage (through gpurun): bash tools/collect_profiles.sh
R=$PWD
OUT=$R/gpurun_out/prof_final
rm -rf $OUT; mkdir -p $OUT
# 1. the bench line (default command) and the concurrent timeline -- FIRST: right behind a rocprofv3 --pmc pass the first replays run at
# the profiler's clocks (seen once: 57 k q/s over the 20 timed steps, 92 k sustained three seconds later)
cd $R
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err
python tools/step_timeline.py > $OUT/step_timeline.txt 2>&1
ls -la $OUT
# 2. the other BASELINE configs: ir-fp (configs[3]) and the 14x14 / B=32 stress shape (configs[4])
python bench.py --config ir-fp --no-cpu-baseline > $OUT/bench_ir_fp.json 2>> $OUT/bench.err
python bench.py --hw 224 --batch 32 --steps 10 --no-cpu-baseline > $OUT/bench_stress_b32_n196.json 2>> $OUT/bench.err
python tools/time_small.py > $OUT/small_kernels_alone.txt 2>/dev/null
python tools/time_wgrad.py > $OUT/wgrad_alone.txt 2>/dev/null
python tools/time_fwd_f16s.py > $OUT/fwd_chain_alone.txt 2>/dev/null
python tools/time_k1.py > $OUT/k1_alone.txt 2>/dev/null
python tools/time_bwd.py > $OUT/bwd_chain_alone.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# 3. kernel trace of the bench command (eager launches: per-kernel durations; the timed bench line itself uses the hipGraph)
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-modes --no-parity --no-kernel-timing --no-graph > $OUT/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/p_kt -name "*.db" | head -1) 13 > $OUT/kernel_stats.csv
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
# 5. round 4: the fused extraction op alone + its written bytes, the convergence runs, the sustained run's clocks
python tools/time_extract.py > $OUT/extract_alone.txt 2>/dev/null
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_xw -o p -- python $R/tools/time_extract.py once > $OUT/pmc_extract.log 2>&1; python $R/tools/pmc_table.py $(find /tmp/p_xw -name "*counter_collection.csv" | head -1) | grep -i "extract" > $OUT/pmc_extract_write.txt)
python tools/convergence.py 300 > $OUT/convergence.txt 2>/dev/null
python -c "
import json; d = json.load(open('$OUT/bench_line.json')); s = d.get('sustained') or {}
print('# rocm-smi clocks / power sampled every 0.25 s during the sustained run of the default bench line (bench.py --sustain 3)')
print(json.dumps({'value_20_steps': d['value'], 'sustained': s}, indent=1))" > $OUT/clocks.txt
ls -la $OUT
