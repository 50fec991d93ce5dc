import os, sys, runpy
import refid_amd._lib as l
l.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py", "--dtype", "bf16", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-roofline"]
runpy.run_path("bench.py", run_name="__main__")
