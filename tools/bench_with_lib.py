"""bench.py with another build of the library: python tools/bench_with_lib.py <lib.so> <bench.py arguments>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointasnl_amd import _hip
_hip.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
