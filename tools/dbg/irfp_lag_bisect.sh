#!/bin/bash
# round 6: the chain path leaves the plateau of the relational task ~100 steps later than fp32 on ir-fp (72 seed pairs, 2 sigma).
# Arms that switch ONE ir-/chain-specific piece of the default mode off, seeds 0..47 (the fp32 runs of those seeds exist):
O=gpurun_out/convergence_seeds_irfp_bisect.jsonl; rm -f $O
python tools/convergence_seeds.py --seeds 48 --models ir-fp --modes auto --override chain_reduce=0 --tag +stored_dz0 --out $O 2>&1 | tail -2
python tools/convergence_seeds.py --seeds 48 --models ir-fp --modes auto --sched dq_async=0 --tag +sync_dq --out $O 2>&1 | tail -2
python tools/convergence_seeds.py --seeds 48 --models ir-fp --modes auto --override wgrad_overlap=0 --tag +no_side_streams --out $O 2>&1 | tail -2
python tools/convergence_seeds.py --seeds 48 --models ir-fp --modes auto --override fphi_split=0 --tag +rowsplit_fphi --out $O 2>&1 | tail -2
