#!/bin/bash
# second set of arms on ir-fp, seeds 0..47: the tile dithering of the forward chain's weight images (product: 4 images per layer)
O=gpurun_out/convergence_seeds_irfp_bisect2.jsonl; rm -f $O
python tools/convergence_seeds.py --seeds 48 --models ir-fp --modes auto --dither 1 --tag +dither1 --out $O 2>&1 | tail -1
python tools/convergence_seeds.py --seeds 48 --models ir-fp --modes auto --dither 8 --tag +dither8 --out $O 2>&1 | tail -1
python tools/convergence_seeds.py --seeds 48 --models original-fp --modes auto --dither 1 --tag +dither1 --out $O 2>&1 | tail -1
