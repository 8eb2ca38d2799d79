"""Ball-query sweep with another build of the library: python tools/ball_ab.py [lib.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointasnl_amd import _hip
if len(sys.argv) > 1:
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
rows, _ = bench.ball_query_sweep(batches=(64, 256, 768, 1024, 2048, 4096, 8192))
print(os.path.basename(_hip.LIB_PATH), " | ".join(f"B={r['B']} {r['median_us']:.1f} us ({100 * r['hbm_frac']:.1f}%)" for r in rows), flush=True)
