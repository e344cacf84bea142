"""The duo list of the default Lloyd filter (kmcuda_amd/csrc/lloyd_duo.hip; reference: the assignment of
src/kmeans.cu:293-364 that every path must reproduce).

Stage 1 keeps four top-2 trackers per row; an undecided row whose contenders are the bests of two different quarters
skips stage 2's sweep and is settled from (row, contender, contender).  Whatever the setting -- never (0), where the
lists are long enough to pay (1, the default), always (2, the suite's default: conftest.py) -- assignments, previous
assignments and the reassignment counter are the oracle's, bit for bit, and the duo rows are counted."""
import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _assign(x, c, metric="L2", cached=False):
    from kmcuda_amd.engine import Engine
    dev = torch.device("cuda", 0)
    n, d = x.shape
    k = c.shape[0]
    xs = torch.from_numpy(x).to(dev)
    cs = torch.from_numpy(c.astype(numpy.float32)).to(dev)
    asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    eng = Engine(n, d, k, metric, device=0)
    if cached:
        eng.set_row_cache(True)
    eng.lloyd_assign(xs, cs, asg, prev)
    counters = eng.counters()
    duo = eng.duo_rows()
    eng.close()
    return asg.cpu().numpy().view(numpy.uint32), prev.cpu().numpy().view(numpy.uint32), counters, duo


def _near_ties(rs, n, d, k, spread):
    """Rows between pairs of centroids: most rows have exactly two contenders."""
    c = rs.rand(k, d).astype(numpy.float32)
    a, b = rs.randint(0, k, n), rs.randint(0, k, n)
    t = (0.5 + spread * rs.randn(n, 1)).astype(numpy.float32)
    x = (t * c[a] + (1 - t) * c[b] + 1e-3 * rs.randn(n, d)).astype(numpy.float32)
    return x, c


@pytest.mark.parametrize("setting", ["0", "1", "2"])
@pytest.mark.parametrize("cached", [False, True])
@pytest.mark.parametrize("n,d,k", [(6000, 16, 64), (5000, 32, 130), (4000, 64, 300), (4096, 100, 257), (6000, 128, 96),
                                   (8000, 256, 1024), (3000, 200, 77)])
def test_rows_between_two_centroids(n, d, k, cached, setting, monkeypatch):
    monkeypatch.setenv("KMCUDA_AMD_DUO", setting)
    rs = numpy.random.RandomState(n + d + k)
    x, c = _near_ties(rs, n, d, k, 1e-4)
    x[5] = numpy.nan                      # kmeans.cu:312
    x[17, 3 % d] = numpy.inf
    got, prev, counters, duo = _assign(x, c, cached=cached)
    ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c)
    assert (got == ref).all()
    assert (prev == ref_prev).all()
    assert counters[0] == ref_changed
    assert (duo > 0) == (setting == "2"), duo   # (these lists are far below one round of stage-2 blocks)


@pytest.mark.parametrize("setting", ["0", "2"])
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("d", [12, 32, 256])
def test_angular_rows_between_two_centroids(d, half, setting, monkeypatch):
    """The angular metric's clamp (products at or beyond 1 tie, the lowest index wins) through the duo kernel."""
    monkeypatch.setenv("KMCUDA_AMD_DUO", setting)
    rs = numpy.random.RandomState(d + 5 * half)
    n, k = 5000, 90
    x, c = _near_ties(rs, n, d, k, 1e-3)
    x /= numpy.linalg.norm(x, axis=1, keepdims=True)
    c /= numpy.linalg.norm(c, axis=1, keepdims=True)
    x[100:140] = c[7]                     # products of 1 with centroid 7 ...
    c[3] = c[7]                           # ... and with its duplicate at the lower index
    if half:   # (rows that exist in both precisions)
        x = x.astype(numpy.float16).astype(numpy.float32)
    from test_gpu_angular_clamp import _check
    for variant in ("f16", "f16-cached"):
        _check(x, [c], variant, half, monkeypatch, exact_rows=numpy.arange(100, 140))


def test_the_duo_list_is_used_and_counted(monkeypatch):
    """KMCUDA_AMD_DUO=2: rows do leave on the duo list (the library's own count: kmamd_engine_duo_rows)."""
    from kmcuda_amd.engine import Engine
    monkeypatch.setenv("KMCUDA_AMD_DUO", "2")
    rs = numpy.random.RandomState(3)
    n, d, k = 20000, 256, 512
    x, c = _near_ties(rs, n, d, k, 1e-4)
    dev = torch.device("cuda", 0)
    xs, cs = torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)
    asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    eng = Engine(n, d, k, "L2", device=0)
    eng.lloyd_assign(xs, cs, asg, prev)
    duo_rows = eng.duo_rows()
    eng.close()
    ref, _, _ = oracle.lloyd_assign(x, c)
    assert (asg.cpu().numpy().view(numpy.uint32) == ref).all()
    assert duo_rows > n // 10, duo_rows
    monkeypatch.setenv("KMCUDA_AMD_DUO", "0")
    eng = Engine(n, d, k, "L2", device=0)
    eng.lloyd_assign(xs, cs, asg, prev)
    assert eng.duo_rows() == 0
    eng.close()


def test_a_whole_call_is_the_same_with_and_without_the_duo_list(monkeypatch):
    """kmeans_cuda() end to end (device-side stop, update, prepared passes): bit-identical centroids and assignments."""
    from kmcuda_amd import kmeans_cuda
    rs = numpy.random.RandomState(11)
    x, _ = _near_ties(rs, 60000, 64, 200, 0.05)
    out = []
    for setting in ("0", "2"):
        monkeypatch.setenv("KMCUDA_AMD_DUO", setting)
        out.append(kmeans_cuda(x, 200, init="random", seed=5, tolerance=0.001, yinyang_t=0, device=1, verbosity=0))
    assert (out[0][1] == out[1][1]).all()
    assert (out[0][0].view(numpy.uint32) == out[1][0].view(numpy.uint32)).all()
