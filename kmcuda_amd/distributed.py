"""Row-sharded Lloyd iterations, one process per GPU, over torch.distributed (RCCL on ROCm).

Reference counterpart: the multi-device loops of src/kmeans.cu:934-1026 (FOR_EACH_DEVI launches
+ CUP2P peer copies of assignment / centroid slices).  The reference replicates all samples on
every GPU; here every rank owns a contiguous block of rows and the ONLY per-iteration exchange is
one all-reduce of a fused fp64 buffer  [delta (K*D) | dcount (K) | counters (4)]  -- 2.0 MiB at
K=1024, D=256 -- after which every rank applies the same update to its replica of the centroids.

The loop is written against a small backend interface so that the reduction / bookkeeping logic
can be exercised on CPU (gloo, world_size 2) in tests with a checker backend; the product
backend is `HipBackend` (HIP kernels through the C ABI) and nothing else ships.
"""
import os

import torch
import torch.distributed as dist


def row_block(n_total, rank, world):
    """Contiguous balanced row blocks, the C++ host's rule (kmcuda_api.cpp: row_plan()): with more than one shard of
    at least 1024 rows each, every shard start is rounded down to a multiple of 256 rows -- the sharded k-means++
    chooser (launch_kmpp_choose) takes whole 256-row blocks from every shard but the last."""
    align = world > 1 and n_total // world >= 1024

    def start(i):
        if i >= world:
            return n_total
        o = (n_total * i) // world
        return (o & ~255) if align else o

    return start(rank), start(rank + 1)


class HipBackend:
    """This rank's rows on its GPU; all compute is libKMCUDA.so kernels."""

    def __init__(self, samples, clusters, metric="L2", device_index=0, half_rows=None, row_cache=True):
        """half_rows: the same rows as a float16 tensor (fp16x2 path): the assignment filter then runs
        on the f16 matrix cores reading the halves; `samples` stays the widened fp32 copy the exact
        refine / update kernels read.  row_cache: this backend's rows never change, so the coarse filter
        stage may keep its centred half copy of them across iterations (Engine.set_row_cache)."""
        from .engine import Engine
        assert samples.is_cuda and samples.dtype == torch.float32 and samples.is_contiguous()
        self.samples = samples
        self.n_local, self.features = samples.shape
        self.clusters = clusters
        self.device = samples.device
        self.engine = Engine(self.n_local, self.features, clusters, metric, device=device_index)
        if half_rows is not None:
            assert half_rows.dtype == torch.float16 and half_rows.shape == samples.shape and half_rows.is_contiguous()
            self.engine.set_half_rows(half_rows)
        self.half = half_rows is not None
        if row_cache:
            self.engine.set_row_cache(True)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.assignments = torch.full((self.n_local,), -1, **i32)   # 0xFFFFFFFF (prepare_mem)
        self.assignments_prev = torch.full((self.n_local,), -1, **i32)
        self.ccounts = torch.zeros(clusters, **i32)
        self.centroids = torch.empty((clusters, self.features), dtype=torch.float32, device=self.device)

    def new_reduce_buffer(self):
        return torch.zeros(self.engine.reduce_len(), dtype=torch.float64, device=self.device)

    def reset_changed(self):
        self.engine.reset_counters(0)

    def assign(self):
        self.engine.lloyd_assign(self.samples, self.centroids, self.assignments, self.assignments_prev)

    def fill_reduce_buffer(self, buf):
        self.engine.reduce_fill(self.samples, self.assignments_prev, self.assignments, buf)

    def apply(self, buf):
        self.engine.reduce_apply(buf, self.centroids, self.ccounts)
        if self.half:   # fp16x2: centroids live in half2 in the reference -> rounded after every update
            self.centroids.copy_(self.centroids.to(torch.float16).to(torch.float32))

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    # ---- the stop rule on the device (kmamd_reduce_apply_stop): the loop enqueues pass i + 1 before it has
    # ---- seen pass i's count, and reads the outcome one pass late from pinned words
    def stop_clear(self):
        self.engine.stop_clear()
        self.engine.reset_counters(0)

    def centroids_written(self):
        """set_centroids() wrote self.centroids in place: the engine knows the buffer by address only, and a
        preparation fused into the last update (apply_stop) would be taken for the new values' (ADVICE r3)."""
        self.engine.centroids_written()

    def apply_stop(self, buf, threshold, seq):
        """The update, unless the reduced reassignment count is <= threshold (then nothing is touched and
        later assign() calls are no-ops).  Returns a handle for read_report()."""
        if self.half:   # the centroids are rounded to halves after the update: no fused preparation
            self.engine.reduce_apply_stop(buf, self.centroids, self.ccounts, threshold, seq)
            self.centroids.copy_(self.centroids.to(torch.float16).to(torch.float32))
        else:           # + the next pass's centroid preparation in the same launch (nobody touches them in between)
            self.engine.reduce_apply_prepare(buf, self.centroids, self.ccounts, threshold, seq)
        return seq

    def read_report(self, handle):
        """(global number of reassigned rows, stopped?) of the pass behind `handle`; waits for that pass only."""
        counters, stopped = self.engine.stop_report(handle)
        return counters[0], stopped


def stop_threshold(tolerance, n_total):
    """tolerance * N as the reference forms it: a float product (kmeans.cu:707)."""
    import numpy
    return float(numpy.float32(tolerance) * numpy.float32(n_total))


class ShardedLloyd:
    """kmeans_cuda_lloyd (kmeans.cu:934-1026) over row shards."""

    def __init__(self, backend, n_total, group=None, reduce_always=False):
        """reduce_always: call the all-reduce also when the group has ONE rank (a test hook: the collective's
        plumbing -- RCCL on the engine's buffer, its ordering with the engine's stream -- on a single-GPU box)."""
        self.b = backend
        self.n_total = n_total
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if reduce_always and dist.is_initialized():
            self.world = max(self.world, 2)   # only ever compared with 1 below
        self.buf = backend.new_reduce_buffer()
        self.iterations = 0
        self.stopped = False
        self.time_collective = False   # bench.py: bracket every all-reduce with events (collective_ms())
        self._collective_events = []
        # The collective is enqueued with the ENGINE's stream as torch's current stream: RCCL then orders with the
        # iteration's kernels through two events on the device.  On torch's default stream -- the legacy NULL stream,
        # which orders with the engine's blocking stream implicitly -- the same call cost 110 us per iteration on a
        # one-rank group (scripts/rccl_one_rank_overhead.py).
        self._collective_stream = None
        # (RCCL only: gloo stages a device tensor through the host and is slower that way -- 85 ms against 3 ms per
        #  step for two ranks sharing one GPU)
        rccl = dist.is_initialized() and dist.get_backend(group) == "nccl"
        handle = backend.engine.stream_handle() if self.world > 1 and rccl and hasattr(backend, "engine") else 0
        if handle and os.environ.get("KMCUDA_AMD_COLLECTIVE_STREAM", "engine") == "engine":
            self._collective_stream = torch.cuda.ExternalStream(handle, device=backend.device)

    def _all_reduce(self):
        stream = self._collective_stream
        ctx = torch.cuda.stream(stream) if stream is not None else None
        timed = self.time_collective and self.buf.is_cuda
        if ctx is not None:
            ctx.__enter__()
        try:
            if timed:   # a pair of events on the stream the collective is enqueued on (collective_ms())
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(torch.cuda.current_stream(self.buf.device))
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)
            if timed:
                b.record(torch.cuda.current_stream(self.buf.device))
                self._collective_events.append((a, b))
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def collective_ms(self):
        """(summed milliseconds, number) of the all-reduces enqueued while `time_collective` was set, as events on their
        stream bracketed them -- the wait for the slowest rank's move sums included: what an iteration pays for the
        exchange.  Waits for the device; clears the record."""
        if not self._collective_events:
            return 0.0, 0
        torch.cuda.synchronize(self.buf.device)
        total = sum(a.elapsed_time(b) for a, b in self._collective_events)
        n = len(self._collective_events)
        self._collective_events = []
        return total, n

    def set_centroids(self, centroids):
        """Replicated initial centroids (rank 0's are broadcast).  Starts a new run: the stop state of an earlier
        one is dropped."""
        self.b.centroids.copy_(centroids)
        if self.world > 1:
            dist.broadcast(self.b.centroids, src=0, group=self.group)
        if hasattr(self.b, "centroids_written"):
            self.b.centroids_written()
        self._new_run()

    def _new_run(self):
        self.iterations = 0          # the next device-side step lowers the engine's stop flag (stop_clear)
        self.stopped = False
        self._pending = None

    def step(self, tolerance=None):
        """One Lloyd iteration: assign, reduce, (stop test), update.  With `tolerance` the reference's
        stop rule is evaluated BEFORE the update (kmeans.cu:991-1000) and the update is skipped when it
        fires.

        A backend with `apply_stop` (HipBackend) decides the rule on the device: this call enqueues the
        whole iteration without waiting and returns the PREVIOUS iteration's global number of reassigned
        rows (None for the first call), read from pinned words -- the way kmeans_cuda() iterates.
        `self.stopped` turns True once an iteration has fired the rule; the passes enqueued after it
        leave the state untouched.  Other backends wait for the count before the update and return it."""
        b = self.b
        kd = b.clusters * b.features
        if tolerance is not None and hasattr(b, "apply_stop"):
            if self.iterations == 0:
                b.stop_clear()
            b.assign()
            b.fill_reduce_buffer(self.buf)
            if self.world > 1:
                self._all_reduce()
            self.iterations += 1
            handle = b.apply_stop(self.buf, stop_threshold(tolerance, self.n_total), self.iterations)
            prev, self._pending = getattr(self, "_pending", None), handle
            if prev is None:
                return None
            changed, stopped = b.read_report(prev)
            self.stopped = self.stopped or stopped
            return changed
        b.reset_changed()
        b.assign()
        b.fill_reduce_buffer(self.buf)
        if self.world > 1:
            self._all_reduce()
        changed = None
        if tolerance is not None:
            changed = int(self.buf[kd + b.clusters].item())   # host sync, like check_changed()
            if changed <= tolerance * self.n_total:
                self.iterations += 1
                return changed
        b.apply(self.buf)
        self.iterations += 1
        return changed

    def drain(self):
        """The report of the last enqueued iteration (device-side stop rule): (changed, stopped)."""
        pending, self._pending = getattr(self, "_pending", None), None
        if pending is None:
            return None
        changed, stopped = self.b.read_report(pending)
        self.stopped = self.stopped or stopped
        return changed

    def changed_last(self):
        kd = self.b.clusters * self.b.features
        return int(self.buf[kd + self.b.clusters].item())

    def run(self, tolerance, max_iter=10000, verbosity=0):
        """Iterates from the current centroids until the stop rule fires.  Every call is a run of its own: a loop
        that has stopped before starts again (flag lowered, counters reset), as a second kmeans_cuda() would."""
        self._new_run()
        log = []

        def note(changed):
            log.append(changed)
            if verbosity > 0 and (not dist.is_initialized() or dist.get_rank(self.group) == 0):
                print("iteration %d: %d reassignments" % (len(log), changed))

        if hasattr(self.b, "apply_stop"):
            # pass i + 1 is enqueued before pass i's count is looked at; after the stop the extra pass is a no-op
            for _ in range(max_iter):
                changed = self.step(tolerance)
                if changed is not None:
                    note(changed)
                if self.stopped:
                    self._pending = None
                    self.iterations = len(log)
                    return log
            note(self.drain())
            return log
        for it in range(1, max_iter + 1):
            changed = self.step(tolerance)
            note(changed)
            if changed <= tolerance * self.n_total:
                break
        return log
