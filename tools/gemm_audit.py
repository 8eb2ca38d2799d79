"""Every vendor GEMM of a model forward (the dense layers of tf_util._dense), its shape, and how fast it runs alone in a
replayed HIP graph: finds the odd-sized ones that fall off the vendor library's fast kernels.
    python tools/gemm_audit.py cls|cls_as|sem_seg|sem_seg_res [batch]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import bench  # noqa: E402
from pointasnl_amd.utils import tf_util  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cls"
shapes = {}
orig_addmm, orig_act = torch.addmm, torch._addmm_activation


def rec(kind):
    def f(b, x, w, *a, **k):
        key = (x.shape[0], x.shape[1], w.shape[1], kind, x.stride(0))
        shapes[key] = shapes.get(key, 0) + 1
        return (orig_addmm if kind == "addmm" else orig_act)(b, x, w, *a, **k)
    return f


torch.addmm, torch._addmm_activation = rec("addmm"), rec("relu")
torch.manual_seed(0)
with torch.no_grad():
    if which.startswith("cls"):
        from pointasnl_amd.models import pointasnl_cls
        B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
        tf_util.set_store(tf_util.VariableStore(seed=1))
        x = torch.rand(B, 1024, 3, device="cuda")
        pointasnl_cls.get_model(x, is_training=False, adaptive_sample=which == "cls_as")
    else:
        import importlib
        m = importlib.import_module("pointasnl_amd.models.pointasnl_" + which)
        B = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if which == "sem_seg" else 8)
        N = 8192 if which == "sem_seg" else 10240
        tf_util.set_store(tf_util.VariableStore(seed=1))
        x = torch.rand(B, N, 3 if which == "sem_seg" else 4, device="cuda") if which == "sem_seg" else torch.rand(B, N, 3, device="cuda")
        try:
            m.get_model(x, False, 20)
        except Exception as e:  # feature channels differ per model: fall back to bench's generator
            print("direct call failed:", e)
            raise
torch.cuda.synchronize()
torch.addmm, torch._addmm_activation = orig_addmm, orig_act


def timed(fn, n=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(5):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n // 5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n // 5 * 5)


rows = []
for (M, K, N, kind, lda), cnt in shapes.items():
    a = torch.randn(M, lda, device="cuda")[:, :K]
    w = torch.randn(K, N, device="cuda"); b = torch.randn(N, device="cuda")
    f = (lambda: orig_act(b, a, w)) if kind == "relu" else (lambda: orig_addmm(b, a, w))
    t = timed(f)
    rows.append((t * cnt, t, M, K, N, kind, cnt, lda))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{which}: {len(rows)} distinct GEMM shapes, {tot:.0f} us when run alone")
for tt, t, M, K, N, kind, cnt, lda in rows:
    print(f"  {t:8.1f} us x{cnt}  M={M:7d} K={K:5d} N={N:5d} lda={lda:5d} {kind:5s} {2 * M * K * N / t / 1e6:6.1f} TF")
