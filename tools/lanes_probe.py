"""Diagnostics: which part of the ScanNet graph dead-locks when two graph instances replay on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import pointasnl_amd as P
from pointasnl_amd.utils import tf_util, pointasnl_util as U
from pointasnl_amd.models import pointasnl_sem_seg

what = sys.argv[1]
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 16
x = torch.from_numpy(B.synth_clouds(3, bsz, 8192)).cuda()
tf_util.set_store(tf_util.VariableStore(seed=1))
kw = dict(is_training=False, bn_decay=None, weight_decay=None)

DEC = {"dec4": (8192, 1024, 3, 64, [128, 128, 128]), "dec3": (1024, 256, 64, 128, [256, 128]),
       "dec2": (256, 64, 128, 256, [256, 256]), "dec1": (64, 32, 256, 512, [512, 512])}
if what.startswith("dec"):
    n1, n2, c1, c2, _ = DEC[what[:4]]
    X1, X2 = x[:, :n1].contiguous(), x[:, :n2].contiguous()
    P1 = torch.randn((bsz, n1, c1), device="cuda")
    P2 = torch.randn((bsz, n2, c2), device="cuda")


if what.startswith("gemm"):
    shapes = [(131072, 4192, 128), (131072, 131, 128), (131072, 128, 128), (16384, 8288, 256), (16384, 320, 128)]
    if what == "gemm_bigk":  # fa_layer2 / fa_layer1 decode_after_conv: few rows, K = (3+c)*32
        shapes = [(4096, 16480, 256), (1024, 16480, 512), (4096, 384, 256), (16384, 8288, 256)]
    if what.startswith("gemm_") and what[5:6].isdigit():  # gemm_M_K_N
        shapes = [tuple(int(v) for v in what.split("_")[1:4])]
    if what == "gemm_small":
        shapes = [(32768, 2048, 128), (8192, 4096, 256), (32768, 131, 128), (32768, 128, 256), (32768, 256, 512)]
    GEMMS = [(torch.randn((m_, k_), device="cuda"), (torch.randn((k_, n_), device="cuda") * 0.01, torch.zeros(n_, device="cuda")))
             for (m_, k_, n_) in shapes]


XC = torch.from_numpy(B.synth_clouds(5, 64, 1024)).cuda()
XR = torch.from_numpy(B.synth_clouds(6, 8, 10240)).cuda() if what == "res" else None


def fwd():
    if what == "fps":
        return P.tf_sampling.farthest_point_sample(1024, x)
    if what == "knn":
        q = x[:, :1024].contiguous()
        return P.nearest_neighbors.knn_batch(x, q, 32, dtype=torch.int32)
    if what == "knn16":
        return P.nearest_neighbors.knn_batch(x, x, 16, dtype=torch.int32)
    if what == "layer1":
        return U.PointASNLSetAbstraction(x, x, npoint=1024, nsample=32, mlp=[32, 32, 64], scope='layer1', as_neighbor=8, **kw)[1]
    if what == "layer1_noas":
        return U.PointASNLSetAbstraction(x, x, npoint=1024, nsample=32, mlp=[32, 32, 64], scope='layer1', as_neighbor=0, **kw)[1]
    if what == "layer1_nonl":
        return U.PointASNLSetAbstraction(x, x, npoint=1024, nsample=32, mlp=[32, 32, 64], scope='layer1', as_neighbor=0, NL=False, **kw)[1]
    if what.startswith("enc"):
        nl = int(what[3:])
        xyz, pts = x, x
        cfg = [(1024, [32, 32, 64], 8), (256, [64, 64, 128], 4), (64, [128, 128, 256], 0), (32, [256, 256, 512], 0)]
        for li in range(nl):
            npnt, mlp, asn = cfg[li]
            xyz, pts = U.PointASNLSetAbstraction(xyz, pts, npoint=npnt, nsample=32, mlp=mlp, scope='layer%d' % (li + 1), as_neighbor=asn, **kw)
        return pts
    if what.startswith("dec"):
        n1, n2, c1, c2, mlp = DEC[what[:4]]
        if what.endswith("_interp"):
            d, i = P.tf_interpolate.three_nn(X1, X2)
            return P.tf_interpolate.three_interpolate(P2, i, P.tf_interpolate.three_weights(d))
        if what.endswith("_cell"):
            return U.decode_cell(X1, P1[:, :, :1].expand(-1, -1, c2).contiguous(), U.knn_query(16, X1, X1))
        return U.PointASNLDecodingLayer(X1, X2, P1, P2, 16, mlp, False, None, None, scope=what)
    if what.startswith("gemm"):
        # only vendor GEMMs: the decode_after_conv / fc shapes of the ScanNet graph
        out = None
        for (a, w_) in GEMMS:
            out = torch._addmm_activation(w_[1], a, w_[0])
        return out
    if what.startswith("model"):
        ndec = int(what[5:])  # encoder + the first ndec decoding layers (the reference graph without the fc head)
        l0_xyz = l0_points = x
        cfg = [(1024, [32, 32, 64], 8), (256, [64, 64, 128], 4), (64, [128, 128, 256], 0), (32, [256, 256, 512], 0)]
        xs, ps = [l0_xyz], [l0_points]
        for li in range(4):
            npnt, mlp, asn = cfg[li]
            a, b_ = U.PointASNLSetAbstraction(xs[-1], ps[-1], npoint=npnt, nsample=32, mlp=mlp, scope='layer%d' % (li + 1), as_neighbor=asn, **kw)
            xs.append(a); ps.append(b_)
        dm = [[512, 512], [256, 256], [256, 128], [128, 128, 128]]
        cur = ps[4]
        for di in range(ndec):
            lvl = 3 - di
            cur = U.PointASNLDecodingLayer(xs[lvl], xs[lvl + 1], ps[lvl], cur, 16, dm[di], False, None, None, scope='fa_layer%d' % (di + 1))
        return cur
    if what == "res":
        from pointasnl_amd.models import pointasnl_sem_seg_res
        return pointasnl_sem_seg_res.get_model(XR, False, 20)[0]
    if what == "cls":
        from pointasnl_amd.models import pointasnl_cls
        return pointasnl_cls.get_model(XC)[0]
    if what == "cls_as":
        from pointasnl_amd.models import pointasnl_cls
        return pointasnl_cls.get_model(XC, adaptive_sample=True)[0]
    if what == "full":
        return pointasnl_sem_seg.get_model(x, False, 20)[0]
    raise SystemExit("unknown")

with torch.no_grad():
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    lanes = []
    for _ in range(2):
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            out = fwd()
        lanes.append((st, g, out))
        print("captured", len(lanes), flush=True)
    torch.cuda.synchronize()
    nrep = int(os.environ.get("NREP", "20"))
    for i in range(nrep):
        st, g, _ = lanes[i % 2]
        with torch.cuda.stream(st):
            g.replay()
            if os.environ.get("JITTER") and i % 7 == 3:
                torch.cuda._sleep(int(2.4e3 * (37 * i % 400)))  # up to 400 us of idle on this lane: shifts the lanes' phase
        if os.environ.get("SYNC_EACH"):
            torch.cuda.synchronize()
        if i < 4 or os.environ.get("SYNC_EACH"):
            print("replayed", i, flush=True)
    torch.cuda.synchronize()
print(f"lanes_probe {what} B={bsz}: OK", flush=True)
