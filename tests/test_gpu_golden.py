"""The HIP path against the committed golden vectors (tests/golden/golden.npz) -- no oracle in the
loop: inputs are regenerated from seeds, outputs compared bit for bit (fp32 L2 assignments, k-NN
indices, and -- with the strict-parity update, KMCUDA_AMD_EXACT_UPDATE=1 -- whole kmeans_cuda()
runs: centroids, assignments, per-iteration reassignment counts).  Angular cases: acosf is libm in
the vectors and ocml on the GPU, so near-ties inside an acos plateau may flip (tolerance below)."""
import os
import sys

import numpy
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as mg  # noqa: E402
from test_gpu_kmeans import StdoutListener  # noqa: E402

GOLDEN = numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.npz"))
COS_FLIPS = 0.002   # fraction of rows that may differ on the angular metric (acosf ulp differences)


@pytest.mark.parametrize("name", sorted(mg.ASSIGN_CASES))
@pytest.mark.parametrize("cached", [False, True])
def test_assign_passes(name, cached):
    from kmcuda_amd.engine import Engine
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    case = mg.ASSIGN_CASES[name]
    metric = case[4]
    x, cs = mg.assign_inputs(*case)
    dev = torch.device("cuda", 0)
    n, k = x.shape[0], cs[0].shape[0]
    eng = Engine(n, x.shape[1], k, metric, device=0)
    if cached:
        eng.set_row_cache(True)
    xs = torch.from_numpy(x).to(dev)
    asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    for p, c in enumerate(cs):
        eng.reset_counters(0)
        eng.lloyd_assign(xs, torch.from_numpy(c).to(dev), asg, prev)
        changed = eng.counters()[0]   # synchronises the engine's stream
        got = asg.cpu().numpy().view(numpy.uint32)
        want = GOLDEN["%s/pass%d/assignments" % (name, p)]
        if metric == "cos":
            # only rows whose two candidates are a last place apart in the oracle's own arithmetic (the one use of the
            # oracle in this file: the distances of the rows that differ, tests/_angular.py)
            from _angular import assert_only_acos_matters
            assert_only_acos_matters(x, c, got, want, "%s pass %d" % (name, p), max_fraction=COS_FLIPS)
            # later passes start from the engine's own previous assignments: put the vectors' in
            asg.copy_(torch.from_numpy(want.view(numpy.int32).copy()).to(dev))
        else:
            assert (got == want).all()
            assert changed == int(GOLDEN["%s/pass%d/changed" % (name, p)][0])
    eng.close()


@pytest.mark.parametrize("name", sorted(mg.KMEANS_CASES))
def test_kmeans_runs_strict(name, monkeypatch):
    from kmcuda_amd import kmeans_cuda
    data, clusters, kw = mg.KMEANS_CASES[name]
    monkeypatch.setenv("KMCUDA_AMD_EXACT_UPDATE", "1")
    out = StdoutListener()
    with out:
        cen, asg = kmeans_cuda(mg.kmeans_data(data), clusters, device=1, verbosity=1, **kw)
    log = [int(l.split(":")[1].split()[0]) for l in out.text.split("\n") if l.startswith("iteration")]
    want_log = list(GOLDEN[name + "/log"])
    if kw.get("metric") == "cos":
        assert abs(len(log) - len(want_log)) <= 1
        assert (asg != GOLDEN[name + "/assignments"]).mean() < 0.01
        return
    assert log == want_log
    assert (asg == GOLDEN[name + "/assignments"]).all()
    assert (mg.bits(cen) == GOLDEN[name + "/centroid_bits"]).all()


@pytest.mark.parametrize("name", sorted(mg.KNN_CASES))
def test_knn(name):
    from kmcuda_amd import knn_cuda
    case = mg.KNN_CASES[name]
    x, init = mg.knn_inputs(*case)
    cen = GOLDEN[name + "/centroid_bits"].view(numpy.float32).copy()
    nb = knn_cuda(case[4], x, cen, GOLDEN[name + "/assignments"].copy(), metric=case[5], device=1)
    assert (nb == GOLDEN[name + "/neighbors"]).all()
