#!/usr/bin/env python3
"""bench.py with scheduling experiments switched on: exp_bench.py wgrad_late=1 conv_wgrad_stream=2 -- [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import relationnetworks_clevr_amd as pkg
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
for kv in args[:cut]:
    k, v = kv.split("=")
    pkg.functional.SCHED[k] = int(v)
sys.argv = ["bench.py"] + args[cut + 1:]
import bench
bench.main()
