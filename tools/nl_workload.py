"""nl_attention_direct alone, for counter passes: python tools/nl_workload.py <lib.so> [shape]  (shape: cls | scannet | kitti)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pointasnl_amd import _hip
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _hip.LIB_PATH = os.path.abspath(sys.argv[1])
from pointasnl_amd.utils import pointasnl_util as U
shape = {"cls": (64, 512, 1024, 32), "scannet": (16, 1024, 8192, 32), "kitti": (8, 1280, 10240, 32), "cls2": (64, 128, 512, 64)}[sys.argv[2] if len(sys.argv) > 2 else "scannet"]
b, p, n, cb = shape
q = torch.randn((b, p, cb), device="cuda"); kv = torch.randn((b, n, 2 * cb), device="cuda")
for _ in range(6):
    U.nl_attention(q, kv, variant=2)
torch.cuda.synchronize()
