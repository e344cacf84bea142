"""The row-sharded Lloyd loop with the PRODUCT backend (HipBackend: HIP kernels through the C ABI)
under torch.distributed, world_size 2.  A test box has one GPU, and RCCL refuses two ranks on one
device, so both ranks sit on GPU 0 and the all-reduce goes over gloo (device tensors); what is
covered is everything else the N > 1 path of bench.py runs: shard bookkeeping, the fused fp64
reduce buffer, ordering between the engine's stream and torch's collectives, the stop rule.

Bar: identical per-iteration reassignment counts and assignments as ONE process over all rows
(the deltas are exact fp64 sums of fp32 values for these sizes, so the split does not change
them), and the first assignment pass bit-exact against the oracle."""
import datetime
import os
import socket
import sys

import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _data():
    rs = numpy.random.RandomState(11)
    x = numpy.concatenate([rs.randn(1500, 64) + 3 * rs.randn(1, 64) for _ in range(8)]).astype(numpy.float32)
    x = x[rs.permutation(len(x))]
    init = x[rs.choice(len(x), 40, replace=False)].copy()
    return x, init


def _worker(rank, world, port, out):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd, row_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    x, init = _data()
    lo, hi = row_block(len(x), rank, world)
    xs = torch.from_numpy(x[lo:hi]).to(dev)
    try:
        probe = torch.ones(4, dtype=torch.float64, device=dev)
        dist.all_reduce(probe)
        torch.cuda.synchronize(dev)
        assert float(probe[0].item()) == world
    except (RuntimeError, AssertionError) as e:   # a gloo build without device-tensor support
        if rank == 0:
            numpy.savez(out, skipped=numpy.array([1]), why=numpy.array([str(e)[:200]]))
        dist.destroy_process_group()
        return
    loop = ShardedLloyd(HipBackend(xs, len(init), "L2", device_index=0), len(x))
    c0 = torch.from_numpy(init).to(dev) if rank == 0 else torch.zeros((len(init), x.shape[1]), device=dev)
    loop.set_centroids(c0)
    first = None
    log = []
    for it in range(60):
        # the stop rule is decided on the device: step() enqueues pass it + 1 and returns pass it's count (one
        # pass late, None at first); once a pass has fired the rule the later ones are no-ops
        changed = loop.step(0.002)
        if changed is not None:
            log.append(changed)
        if it == 0:
            torch.cuda.synchronize(dev)
            first = loop.b.assignments.cpu().numpy().view(numpy.uint32).copy()
        if loop.stopped:
            break
    torch.cuda.synchronize(dev)
    mine = loop.b.assignments.cpu().numpy().view(numpy.uint32).copy()
    gathered, firsts = [None] * world, [None] * world
    dist.all_gather_object(gathered, mine)
    dist.all_gather_object(firsts, first)
    if rank == 0:
        numpy.savez(out, skipped=numpy.array([0]), log=numpy.array(log), asg=numpy.concatenate(gathered),
                    first=numpy.concatenate(firsts), cen=loop.b.centroids.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_hip_backend_world2_matches_single(tmp_path):
    import torch.multiprocessing as mp
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "w2.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = numpy.load(out)
    if int(got["skipped"][0]):
        pytest.skip("gloo cannot reduce device tensors here: %s" % got["why"][0])
    x, init = _data()
    dev = torch.device("cuda", 0)
    loop = ShardedLloyd(HipBackend(torch.from_numpy(x).to(dev), len(init), "L2", device_index=0), len(x))
    loop.set_centroids(torch.from_numpy(init).to(dev))
    log = loop.run(tolerance=0.002, max_iter=60)
    torch.cuda.synchronize(dev)
    single = loop.b.assignments.cpu().numpy().view(numpy.uint32)
    assert list(got["log"]) == log
    assert len(log) > 3 and log[-1] <= 0.002 * len(x)
    assert (got["asg"] == single).all()
    numpy.testing.assert_allclose(got["cen"], loop.b.centroids.cpu().numpy(), rtol=1e-6, atol=1e-7)
    ref, _, _ = oracle.lloyd_assign(x, init)
    assert (got["first"] == ref).all()


def _rccl_worker(rank, port, out):
    """ONE rank, backend "nccl" (= RCCL on ROCm): the all-reduce of the loop's fp64 buffer goes through RCCL on the
    buffer the engine's kernels fill and consume (KMCUDA_AMD reduce_always hook)."""
    import torch.distributed as dist
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    x, init = _data()
    loop = ShardedLloyd(HipBackend(torch.from_numpy(x).to(dev), len(init), "L2", device_index=0), len(x), reduce_always=True)
    loop.set_centroids(torch.from_numpy(init).to(dev))
    loop.time_collective = True   # (bench.py's collective_ms_per_step: a pair of events around every all-reduce)
    log = loop.run(tolerance=0.002, max_iter=60)
    torch.cuda.synchronize(dev)
    coll_ms, coll_n = loop.collective_ms()
    numpy.savez(out, log=numpy.array(log), asg=loop.b.assignments.cpu().numpy().view(numpy.uint32),
                cen=loop.b.centroids.cpu().numpy(), coll=numpy.array([coll_ms, coll_n]))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_one_rank_all_reduce_in_the_loop(tmp_path):
    """The collective of a real multi-GPU run, as far as one GPU can go: RCCL (torch's "nccl" backend) reduces the
    loop's buffer every iteration in a one-rank group; the run must equal the loop without any process group."""
    import torch.multiprocessing as mp
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rccl1.npz")
    mp.spawn(_rccl_worker, args=(port, out), nprocs=1, join=True)
    got = numpy.load(out)
    x, init = _data()
    dev = torch.device("cuda", 0)
    loop = ShardedLloyd(HipBackend(torch.from_numpy(x).to(dev), len(init), "L2", device_index=0), len(x))
    loop.set_centroids(torch.from_numpy(init).to(dev))
    log = loop.run(tolerance=0.002, max_iter=60)
    torch.cuda.synchronize(dev)
    assert list(got["log"]) == log
    assert (got["asg"] == loop.b.assignments.cpu().numpy().view(numpy.uint32)).all()
    assert numpy.array_equal(got["cen"], loop.b.centroids.cpu().numpy(), equal_nan=True)
    # every enqueued iteration's all-reduce was bracketed (the speculative loop enqueues one pass past the stop)
    assert got["coll"][1] >= len(log) and 0.0 < got["coll"][0] < 1000.0, got["coll"]


def test_rccl_one_rank_inside_kmeans_cuda(monkeypatch):
    """kmeans_cuda() with KMCUDA_AMD_FORCE_RCCL=1: the library dlopens RCCL, creates a one-rank communicator
    (ncclCommInitAll) and sends every iteration's reduce buffer through a grouped ncclAllReduce on the shard's stream --
    the multi-GPU code path's calls, signatures and enum values on a single GPU.  Same run as without."""
    from kmcuda_amd import kmeans_cuda
    x, _ = _data()
    c0, a0 = kmeans_cuda(x, 24, init="k-means++", seed=5, tolerance=0.002, yinyang_t=0, device=1)
    monkeypatch.setenv("KMCUDA_AMD_FORCE_RCCL", "1")
    monkeypatch.setenv("KMCUDA_AMD_TIME_COLLECTIVE", "1")
    c1, a1 = kmeans_cuda(x, 24, init="k-means++", seed=5, tolerance=0.002, yinyang_t=0, device=1)
    assert (a0 == a1).all()
    assert numpy.array_equal(c0, c1, equal_nan=True)
    # the library's own clock around its grouped ncclAllReduce (kmamd_last_run_collective; bench.py --api reports it)
    import ctypes
    from kmcuda_amd import _lib
    it, ranks, ms, n = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_double(), ctypes.c_uint32()
    _lib.lib().kmamd_last_run_stats(ctypes.byref(it), None, None, None, ctypes.byref(ranks))
    _lib.lib().kmamd_last_run_collective(ctypes.byref(ms), ctypes.byref(n))
    assert ranks.value == 1 and n.value >= it.value > 0 and 0.0 < ms.value < 1000.0, (ranks.value, n.value, it.value, ms.value)


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu_reports_its_communicator_and_collective():
    """bench.py --gpus 2 as the driver starts it (python bench.py --gpus N: it becomes the launcher of N ranks), with the
    test hooks that put both ranks on GPU 0 over gloo: the line must say how many ranks the communicator of the timed
    steps really had and how long the all-reduce took per step, by events on its stream -- so that the first real
    multi-GPU run explains itself -- and the timed state must verify against the oracle."""
    import json
    import subprocess
    env = dict(os.environ, KMCUDA_AMD_BENCH_SINGLE_DEVICE="1", KMCUDA_AMD_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--samples", "200000", "--clusters",
                          "256", "--steps", "4", "--warmup", "3", "--no-cpu-baseline", "--verify-rows", "20000"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen_by_communicator"] == 2
    assert d["config"]["collectives_timed"] == 4 and d["config"]["collective_ms_per_step"] > 0.0
    assert d["verify"]["ok"], d["verify"]
    assert d["rows_pair_refined_last_step"] + d["rows_full_exact_scan_last_step"] >= 0 and d["reassigned_last_step"] > 0

