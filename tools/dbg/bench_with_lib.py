#!/usr/bin/env python3
"""bench.py against another build of the library (A/B of compile-time variants on one box): bench_with_lib.py <lib.so> [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import relationnetworks_clevr_amd as pkg
pkg.rn_hip.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
