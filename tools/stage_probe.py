"""Diagnostics: durations of the search-prefix graph P, the rest graph R, and both on two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pointasnl_amd.utils import tf_util, pointasnl_util as U
from pointasnl_amd.models import pointasnl_cls

x = torch.from_numpy(B.synth_clouds(3, 64, 1024)).cuda()
tf_util.set_store(tf_util.VariableStore(seed=1))
fl = pointasnl_cls.first_layer(1024)

def cap(fn, st):
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        out = fn()
    torch.cuda.current_stream().wait_stream(st)
    return g, out

def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

with torch.no_grad():
    for _ in range(2):
        pointasnl_cls.get_model(x)
    sp, sr = torch.cuda.Stream(), torch.cuda.Stream()
    P, srch = cap(lambda: U.sa_search(x, x, **fl), sp)
    R, out = cap(lambda: pointasnl_cls.get_model(x, search=srch)[0], sr)
    F, out2 = cap(lambda: pointasnl_cls.get_model(x)[0], sr)
    def runP():
        with torch.cuda.stream(sp): P.replay()
    def runR():
        with torch.cuda.stream(sr): R.replay()
    def runF():
        with torch.cuda.stream(sr): F.replay()
    def both():  # independent: no events, P and R just run side by side
        runP(); runR()
    print("P alone %.0f us, R alone %.0f us, full %.0f us, P||R (no dependency) %.0f us" % (timeit(runP), timeit(runR), timeit(runF), timeit(both)))
