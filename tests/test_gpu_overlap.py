"""GPU: the forked searches (pointasnl_util.Forked: FPS / kNN / three_nn of a level on side streams beside the dense work
of the level above) change WHEN kernels run, never what they compute: the forward with forks -- eagerly and as a replayed
HIP graph, first replay included (a missing join shows up there: the consumer reads pool garbage) -- is bit-identical to
the forward with every kernel on one stream."""
import importlib

import pytest
import torch

import bench as B

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,bsz,n", [("cls", 16, 1024), ("sem_seg", 2, 8192), ("sem_seg_res", 2, 10240)])
def test_forked_forward_is_bit_identical_to_single_stream(model, bsz, n, monkeypatch):
    from pointasnl_amd.utils import pointasnl_util as U
    from pointasnl_amd.utils import tf_util

    M = importlib.import_module(f"pointasnl_amd.models.pointasnl_{model}")
    x = torch.from_numpy(B.synth_clouds(5, bsz, n)).cuda()
    tf_util.set_store(tf_util.VariableStore(seed=1))

    def fwd():
        with torch.no_grad():
            if model == "cls":
                return M.get_model(x, is_training=False, adaptive_sample=True)[0]
            return M.get_model(x, False, 20)[0]

    monkeypatch.setattr(U, "OVERLAP", False)
    ref = fwd().clone()
    torch.cuda.synchronize()
    monkeypatch.setattr(U, "OVERLAP", True)
    for _ in range(5):
        out = fwd()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        gout = fwd()
    torch.cuda.current_stream().wait_stream(s)
    for i in range(5):
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(gout, ref), f"graph replay {i}"
