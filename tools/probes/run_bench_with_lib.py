#!/usr/bin/env python3
"""Run bench.py on another build of the library (ablation / A-B builds):
   python tools/probes/run_bench_with_lib.py path/to/librefid_variant.so [bench.py arguments]"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import refid_amd._lib as l

l.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(l.__file__), "..", "bench.py"), run_name="__main__")
