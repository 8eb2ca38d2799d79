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


def test_split_search_equals_plain_search():
    """sa_search_split (the sampler alone on a side stream, neighbour rows joined at use, one gather for every consumer) returns
    what sa_search returns on the same coordinates, with the self-kNN given as a tensor, as a Forked, or wider than nsample."""
    from pointasnl_amd.utils import pointasnl_util as U

    x = torch.from_numpy(B.synth_clouds(9, 3, 4096)).cuda()
    k_all = U.knn_query(32, x, x)
    ref = U.sa_search(x, None, 512, 32, knn_all=k_all)
    for knn_all in (k_all, U.Forked(lambda: U.knn_query(32, x, x), slot=1)):
        d = U.sa_search_split(x, 512, 32, knn_all, slot=0)
        got = d.get()
        assert d.get() is got  # joined once
        torch.cuda.synchronize()
        assert torch.equal(got[0], ref[0]) and got[1] is None and torch.equal(got[2], ref[2])
    got16 = U.sa_search_split(x, 512, 16, k_all).get()
    ref16 = U.sa_search(x, None, 512, 16, knn_all=k_all)
    torch.cuda.synchronize()
    assert torch.equal(got16[2], ref16[2]) and got16[2].shape[-1] == 16
