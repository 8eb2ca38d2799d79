import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes
from pointasnl_amd import _hip
_hip.lib()
for (b, n, k0, c) in ((64, 512, 132, (128, 256, 512)), (64, 128, 260, (256, 512, 1024))):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(b, n, k0, device="cuda", generator=g)
    ws = []
    cin = k0
    for co in c:
        ws += [torch.randn(cin, co, device="cuda", generator=g) * 0.05, torch.randn(co, device="cuda", generator=g) * 0.05]
        cin = co
    def packed(w):
        pk = torch.empty(int(_hip.lib().pasnl_mlp3_packed_weights_bytes(w.shape[0], w.shape[1])) // 4, device="cuda")
        _hip.launch("pasnl_mlp3_pack_weights", "mlp3_pack", w.shape[0], w.shape[1], _hip.ptr(w), _hip.ptr(pk))
        return pk
    wp = [packed(t) if t.dim() == 2 else t for t in ws]
    out = torch.zeros(b, c[2], device="cuda")
    wsb = torch.empty(int(_hip.lib().pasnl_mlp3_max_pool_workspace_bytes(b, n, c[2])), dtype=torch.uint8, device="cuda")
    def run():
        _hip.launch("pasnl_mlp3_max_pool", "mlp3", b, n, k0, c[0], c[1], c[2], _hip.ptr(x), *[_hip.ptr(t) for t in wp], _hip.ptr(out), ctypes.c_long(c[2]), _hip.ptr(wsb), ctypes.c_size_t(wsb.numel()))
    def vendor():
        h = x.reshape(b * n, k0)
        for i in range(3):
            h = torch._addmm_activation(ws[2 * i + 1], h, ws[2 * i])
        return h.reshape(b, n, -1).amax(1)
    for f, name in ((run, "fused"), (vendor, "vendor chain")):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        fl = 2 * b * n * (k0 * c[0] + c[0] * c[1] + c[1] * c[2])
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{name:14s} b={b} n={n} {c}: {us:7.1f} us  {fl / us / 1e6:6.1f} TF")
    print("max diff", float((out - vendor()).abs().max()), float(out.abs().max()))
