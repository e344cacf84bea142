"""World-size-2 gloo test of the row-sharded Lloyd loop (kmcuda_amd/distributed.py): the
reduction / stop-rule / bookkeeping logic that bench.py --gpus N and multi-GPU deployments run,
exercised on CPU with a CHECKER backend built on the oracle (test infrastructure; the product
backend is HipBackend).  The sharded run must reproduce a single-process run of the same loop:
identical per-iteration reassignment counts and assignments, centroids equal to fp64 round-off."""
import os
import socket
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

torch = pytest.importorskip("torch")


class OracleBackend:
    """Same interface as HipBackend, computed with the CPU oracle (assignments bit-exact with the
    HIP kernels; deltas in fp64 like update.hip)."""

    def __init__(self, samples, clusters):
        import oracle
        self.oracle = oracle
        self.x = numpy.ascontiguousarray(samples, dtype=numpy.float32)
        self.n_local, self.features = self.x.shape
        self.clusters = clusters
        self.assignments = numpy.full(self.n_local, 0xFFFFFFFF, numpy.uint32)
        self.assignments_prev = numpy.full(self.n_local, 0xFFFFFFFF, numpy.uint32)
        self.ccounts = numpy.zeros(clusters, numpy.int64)
        self.centroids = torch.empty((clusters, self.features), dtype=torch.float32)
        self.changed = 0

    def new_reduce_buffer(self):
        return torch.zeros(self.clusters * self.features + self.clusters + 4, dtype=torch.float64)

    def reset_changed(self):
        self.changed = 0

    def assign(self):
        a, p, ch = self.oracle.lloyd_assign(self.x, self.centroids.numpy(), assignments=self.assignments)
        self.assignments, self.assignments_prev = a, p
        self.changed += ch

    def fill_reduce_buffer(self, buf):
        k, d = self.clusters, self.features
        delta = numpy.zeros((k, d), numpy.float64)
        dcount = numpy.zeros(k, numpy.float64)
        moved = self.assignments != self.assignments_prev
        x64 = self.x.astype(numpy.float64)
        for s in numpy.nonzero(moved)[0]:
            a, p = self.assignments[s], self.assignments_prev[s]
            if a < k:
                delta[a] += x64[s]
                dcount[a] += 1
            if p < k:
                delta[p] -= x64[s]
                dcount[p] -= 1
        out = buf.numpy()
        out[:k * d] = delta.ravel()
        out[k * d:k * d + k] = dcount
        out[k * d + k:] = [self.changed, 0, 0, 0]

    def apply(self, buf):
        k, d = self.clusters, self.features
        b = buf.numpy()
        delta = b[:k * d].reshape(k, d)
        dcount = b[k * d:k * d + k].astype(numpy.int64)
        new_counts = self.ccounts + dcount
        c = self.centroids.numpy().astype(numpy.float64) * self.ccounts[:, None] + delta
        with numpy.errstate(divide="ignore", invalid="ignore"):
            c = c / new_counts[:, None]
        self.centroids.copy_(torch.from_numpy(c.astype(numpy.float32)))
        self.ccounts = new_counts

    def synchronize(self):
        pass


class OracleBackendStop(OracleBackend):
    """+ the device-side stop rule's interface (HipBackend.apply_stop / read_report / stop_clear): the
    loop then enqueues pass i + 1 before looking at pass i's count, and a pass after the stop must be a
    no-op.  Here everything is synchronous; what is exercised is ShardedLloyd's lagged bookkeeping."""

    def stop_clear(self):
        self.stopped = False
        self.changed = 0

    def assign(self):
        if self.stopped:
            return
        OracleBackend.assign(self)

    def apply_stop(self, buf, threshold, seq):
        k, d = self.clusters, self.features
        changed = int(buf.numpy()[k * d + k])
        stop = self.stopped or numpy.float32(changed) <= numpy.float32(threshold)
        if stop:
            self.stopped = True
        else:
            self.apply(buf)
            self.changed = 0
        return seq, changed, stop

    def read_report(self, handle):
        return handle[1], handle[2]


BACKENDS = {"host-stop": OracleBackend, "device-stop": OracleBackendStop}


def _data():
    rs = numpy.random.RandomState(5)
    x = numpy.concatenate([rs.randn(700, 8) + 4 * rs.randn(1, 8) for _ in range(6)]).astype(numpy.float32)
    init = x[rs.choice(len(x), 12, replace=False)].copy()
    return x, init


def _worker(rank, world, port, out, kind):
    import torch.distributed as dist
    from kmcuda_amd.distributed import ShardedLloyd, row_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, init = _data()
    lo, hi = row_block(len(x), rank, world)
    loop = ShardedLloyd(BACKENDS[kind](x[lo:hi], 12), len(x))
    loop.set_centroids(torch.from_numpy(init) if rank == 0 else torch.zeros_like(torch.from_numpy(init)))
    log = loop.run(tolerance=0.005, max_iter=50)
    gathered = [None] * world
    dist.all_gather_object(gathered, loop.b.assignments)
    if rank == 0:
        numpy.savez(out, log=numpy.array(log), asg=numpy.concatenate(gathered), cen=loop.b.centroids.numpy())
    dist.destroy_process_group()


def test_row_block_partition():
    from kmcuda_amd.distributed import row_block
    for n, w in [(10, 3), (8000000, 8), (7, 8), (13000, 2)]:
        blocks = [row_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        # balanced up to the 256-row alignment of the shard starts (kmcuda_api.cpp: row_plan(), shards of >= 1024 rows)
        aligned = w > 1 and n // w >= 1024
        assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= (511 if aligned else 1)
        if aligned:
            assert all(b[0] % 256 == 0 for b in blocks)


@pytest.mark.parametrize("kind", ["host-stop", "device-stop"])
def test_sharded_lloyd_world2_matches_single(tmp_path, kind):
    import torch.multiprocessing as mp
    from kmcuda_amd.distributed import ShardedLloyd
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "w2.npz")
    mp.spawn(_worker, args=(2, port, out, kind), nprocs=2, join=True)
    got = numpy.load(out)
    # single process, same loop, no process group
    x, init = _data()
    loop = ShardedLloyd(BACKENDS[kind](x, 12), len(x))
    loop.set_centroids(torch.from_numpy(init))
    log = loop.run(tolerance=0.005, max_iter=50)
    assert list(got["log"]) == log
    assert loop.iterations == len(log)
    assert len(log) > 3 and log[-1] <= 0.005 * len(x)
    assert (got["asg"] == loop.b.assignments).all()
    numpy.testing.assert_allclose(got["cen"], loop.b.centroids.numpy(), rtol=1e-6, atol=1e-7)
    # and it is the reference's Lloyd: same stop iteration and assignments as the oracle's own
    # kmeans loop from the same initial centroids (update: fp64 vs the reference's fp32 chain)
    import oracle
    ocen, oasg, olog = oracle.kmeans(x, 12, init=init, tolerance=0.005, yinyang_t=0)
    assert len(olog) == len(log)
    assert (oasg != loop.b.assignments).mean() < 0.002


@pytest.mark.parametrize("kind", ["host-stop", "device-stop"])
def test_loop_runs_again_after_a_stop(kind):
    """a ShardedLloyd that has stopped must run again -- run() / set_centroids() start a new run (stop
    flag lowered, lagged report dropped, iteration count from zero), and set_centroids() tells the backend that the
    centroid buffer was written behind the engine's back (HipBackend: the fused preparation is void)."""
    from kmcuda_amd.distributed import ShardedLloyd
    x, init = _data()

    class Backend(BACKENDS[kind]):
        written = 0

        def centroids_written(self):
            self.written += 1

    loop = ShardedLloyd(Backend(x, 12), len(x))
    loop.set_centroids(torch.from_numpy(init))
    assert loop.b.written == 1
    first = loop.run(tolerance=0.005, max_iter=50)
    assert len(first) > 3 and loop.stopped == (kind == "device-stop")
    asg = loop.b.assignments.copy()
    # again from where it stands: the stop test fires on the first pass (the converged state), nothing moves
    again = loop.run(tolerance=0.005, max_iter=50)
    assert len(again) >= 1 and again[-1] <= 0.005 * len(x) and len(again) < len(first)
    # new seeds on the same loop: the same trajectory as a fresh loop's, counts and assignments
    loop.b.assignments[:] = 0xFFFFFFFF
    loop.b.assignments_prev[:] = 0xFFFFFFFF
    loop.b.ccounts[:] = 0
    loop.set_centroids(torch.from_numpy(init))
    assert loop.b.written == 2 and loop.iterations == 0 and not loop.stopped
    third = loop.run(tolerance=0.005, max_iter=50)
    assert third == first
    assert (loop.b.assignments == asg).all()
