"""Step-level host API over include/kmcuda_amd.h (one Engine = one GPU's row shard).

torch is used for device memory and streams only; every compute step is a HIP kernel of
libKMCUDA.so reached through the C ABI with raw device pointers.
"""
import ctypes

import torch

from . import _lib

L2, COS = 0, 1
_METRICS = {"L2": L2, "l2": L2, "euclidean": L2, "cos": COS, "cosine": COS, "angular": COS}


def metric_id(metric):
    if isinstance(metric, int):
        return metric
    return _METRICS[metric]


class Engine:
    """Workspace + kernels for `n_rows` local rows of D features against K centroids."""

    def __init__(self, n_rows, features, clusters, metric="L2", device=0, use_torch_stream=True):
        self.lib = _lib.lib()
        self.n_rows, self.features, self.clusters = int(n_rows), int(features), int(clusters)
        self.metric = metric_id(metric)
        self.device = torch.device("cuda", device)
        stream = None
        if use_torch_stream:
            # torch's default stream is the legacy NULL stream (handle 0): ask for an own BLOCKING
            # stream then (-1), which orders with it implicitly; a NULL argument would mean an own
            # NON-blocking stream, racing with every torch op (copies, NCCL) on these buffers
            handle = torch.cuda.current_stream(self.device).cuda_stream
            stream = ctypes.c_void_p(handle if handle else ctypes.c_void_p(-1).value)
        h = ctypes.c_void_p()
        rc = self.lib.kmamd_engine_create(ctypes.byref(h), device, self.n_rows, self.features, self.clusters,
                                          self.metric, 0, stream)
        _lib.check(rc, "kmamd_engine_create")
        self.h = h

    def stream_handle(self):
        """The hipStream_t every step of this engine is enqueued on (kmamd_engine_stream)."""
        return int(self.lib.kmamd_engine_stream(self.h) or 0)

    def close(self):
        if getattr(self, "h", None):
            self.lib.kmamd_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr())

    def lloyd_assign(self, samples, centroids, assignments, assignments_prev, exact=False):
        fn = self.lib.kmamd_lloyd_assign_exact if exact else self.lib.kmamd_lloyd_assign
        _lib.check(fn(self.h, self._p(samples), self._p(centroids), self._p(assignments), self._p(assignments_prev)),
                   "kmamd_lloyd_assign")

    def set_filter(self, mode):
        """"f16" (default): two-stage f16 matrix-core filter; "f32": f32 matrix cores (cross-check)."""
        _lib.check(self.lib.kmamd_set_filter(self.h, {"f16": 0, "f32": 1}[mode]), "kmamd_set_filter")

    def set_half_rows(self, rows16):
        """rows16: float16 CUDA tensor with the same values as the fp32 rows (or None)."""
        self._half_rows = rows16  # keep alive
        _lib.check(self.lib.kmamd_set_half_rows(self.h, self._p(rows16) if rows16 is not None else None),
                   "kmamd_set_half_rows")

    def set_row_cache(self, on=True):
        """Promise that lloyd_assign() gets the same, unmodified rows from now on: the coarse filter
        stage then keeps x - mean as halves in matrix-core operand order (built by the next
        lloyd_assign, mean frozen) and streams that copy instead of converting the rows every pass.
        Assignments are unchanged.  Calling it again drops the copy."""
        _lib.check(self.lib.kmamd_set_row_cache(self.h, 1 if on else 0), "kmamd_set_row_cache")

    def set_carry(self, on=True):
        """Carry per-row distance bounds from pass to pass (kmamd_set_carry): in the two-stage filter's steady state
        (row cache valid; both metrics) a pass only looks at the rows whose bounds -- read off the last pass's coarse scores,
        moved by the centroids' drifts -- no longer certify their assignment.  Results are those of plain passes.

        Contract (include/kmcuda_amd.h): between two passes the centroids change only through reduce_apply* /
        apply_delta / adjust_exact or are announced with centroids_written() (which voids the bounds), and
        `assignments` / `assignments_prev` are the previous pass's buffers, untouched.  A caller that rewrites the
        assignments between passes (a re-seed, a verification step) calls set_carry(True) again -- or
        centroids_written() -- first: the bounds describe the LAST pass's assignments and nothing else voids them."""
        _lib.check(self.lib.kmamd_set_carry(self.h, 1 if on else 0), "kmamd_set_carry")

    def carry_stats(self):
        """(row passes the carried bounds have decided so far, length of the newest row list the host knows)."""
        spared, last = ctypes.c_uint64(0), ctypes.c_uint32(0)
        _lib.check(self.lib.kmamd_carry_stats(self.h, ctypes.byref(spared), ctypes.byref(last)), "kmamd_carry_stats")
        return int(spared.value), int(last.value)

    def carry_pair_stats(self):
        """Row passes the carried pair certificates have sent straight to the two-contender kernel so far (L2)."""
        paired = ctypes.c_uint64(0)
        _lib.check(self.lib.kmamd_carry_pair_stats(self.h, ctypes.byref(paired)), "kmamd_carry_pair_stats")
        return int(paired.value)

    def duo_rows(self):
        """Rows of the last assignment pass settled from stage 1's duo list (lloyd_duo.hip), without stage 2's sweep."""
        rows = ctypes.c_uint32(0)
        _lib.check(self.lib.kmamd_duo_rows(self.h, ctypes.byref(rows)), "kmamd_duo_rows")
        return int(rows.value)

    def counters(self):
        out = (ctypes.c_uint32 * 4)()
        _lib.check(self.lib.kmamd_counters_read(self.h, out), "kmamd_counters_read")
        return list(out)

    def reset_counters(self, which=-1):
        _lib.check(self.lib.kmamd_counters_reset(self.h, which), "kmamd_counters_reset")

    def move_deltas(self, samples, prev, cur, delta, dcount):
        _lib.check(self.lib.kmamd_move_deltas(self.h, self._p(samples), self._p(prev), self._p(cur), self._p(delta),
                                              self._p(dcount)), "kmamd_move_deltas")

    def apply_delta(self, delta, dcount, centroids, ccounts):
        _lib.check(self.lib.kmamd_apply_delta(self.h, self._p(delta), self._p(dcount), self._p(centroids),
                                              self._p(ccounts)), "kmamd_apply_delta")

    def adjust_exact(self, samples, prev, cur, centroids, ccounts):
        _lib.check(self.lib.kmamd_adjust_exact(self.h, self._p(samples), self._p(prev), self._p(cur),
                                               self._p(centroids), self._p(ccounts)), "kmamd_adjust_exact")

    def reduce_len(self):
        """Doubles in the fused reduce buffer: [delta K*D | dcount K | counters 4]."""
        return int(self.lib.kmamd_reduce_len(self.h))

    def reduce_fill(self, samples, prev, cur, buf):
        """This shard's move sums, count changes and counters into `buf` (float64, reduce_len())."""
        _lib.check(self.lib.kmamd_reduce_fill(self.h, self._p(samples), self._p(prev), self._p(cur), self._p(buf)),
                   "kmamd_reduce_fill")

    def reduce_apply(self, buf, centroids, ccounts):
        """The centroid update from the (all-reduced) buffer."""
        _lib.check(self.lib.kmamd_reduce_apply(self.h, self._p(buf), self._p(centroids), self._p(ccounts)),
                   "kmamd_reduce_apply")

    def reduce_apply_stop(self, buf, centroids, ccounts, stop_threshold, seq):
        """reduce_apply with the stop rule decided on the device (kmamd_reduce_apply_stop): nothing is
        modified and the engine's stop flag is raised when the reduced reassignment count is <=
        stop_threshold.  stop_report(seq) returns what the kernel reported."""
        _lib.check(self.lib.kmamd_reduce_apply_stop(self.h, self._p(buf), self._p(centroids), self._p(ccounts),
                                                    float(stop_threshold), int(seq)), "kmamd_reduce_apply_stop")

    def reduce_apply_prepare(self, buf, centroids, ccounts, stop_threshold, seq):
        """reduce_apply_stop + the next lloyd_assign's centroid preparation in one launch where the engine can
        (kmamd_reduce_apply_prepare); the caller must leave `centroids` alone until that lloyd_assign."""
        _lib.check(self.lib.kmamd_reduce_apply_prepare(self.h, self._p(buf), self._p(centroids), self._p(ccounts),
                                                       float(stop_threshold), int(seq)), "kmamd_reduce_apply_prepare")

    def stop_report(self, seq):
        """(reduced counters [4], stopped?) of the reduce_apply_stop call numbered `seq`; waits for that call only."""
        out = (ctypes.c_uint32 * 6)()
        _lib.check(self.lib.kmamd_stop_report(self.h, int(seq), out), "kmamd_stop_report")
        return list(out[:4]), bool(out[4])

    def stop_clear(self):
        _lib.check(self.lib.kmamd_stop_clear(self.h), "kmamd_stop_clear")

    def centroids_written(self):
        """The caller wrote the centroid buffer itself: drop any preparation fused into the last update."""
        _lib.check(self.lib.kmamd_centroids_written(self.h), "kmamd_centroids_written")

    def set_update_mode(self, mode):
        """"auto" | "radix" | "sync" | "bucket": the update's host logic (kmamd_set_update_mode); sums are
        bit-identical on every path."""
        _lib.check(self.lib.kmamd_set_update_mode(self.h, {"auto": 0, "radix": 1, "sync": 2, "bucket": 3}[mode]),
                   "kmamd_set_update_mode")

    def transpose(self, src, rows, cols, dst):
        _lib.check(self.lib.kmamd_transpose(self.h, self._p(src), rows, cols, self._p(dst)), "kmamd_transpose")

    def yy_configure(self, groups_n, groups):
        """groups: host numpy uint32[K] (centroid -> group)."""
        import numpy
        g = numpy.ascontiguousarray(groups, dtype=numpy.uint32)
        _lib.check(self.lib.kmamd_yy_configure(self.h, groups_n, ctypes.c_void_p(g.ctypes.data)), "kmamd_yy_configure")

    def yy_init(self, samples, centroids, assignments, bounds):
        _lib.check(self.lib.kmamd_yy_init(self.h, self._p(samples), self._p(centroids), self._p(assignments),
                                          self._p(bounds)), "kmamd_yy_init")

    def yy_drifts(self, centroids, drifts, gdrifts):
        _lib.check(self.lib.kmamd_yy_drifts(self.h, self._p(centroids), self._p(drifts), self._p(gdrifts)),
                   "kmamd_yy_drifts")

    def yy_filters(self, samples, centroids, drifts, gdrifts, assignments, assignments_prev, bounds, passed):
        _lib.check(self.lib.kmamd_yy_filters(self.h, self._p(samples), self._p(centroids), self._p(drifts),
                                             self._p(gdrifts), self._p(assignments), self._p(assignments_prev),
                                             self._p(bounds), self._p(passed)), "kmamd_yy_filters")

    def yy_hint_stats(self):
        """[rows the hinted local filter processed, rows it handed to the plain kernel, then the hand-overs
        by first reason: no estimate, skipped candidate below its bound, candidate bound above the estimate, final second
        minimum above the estimate] since creation."""
        out = (ctypes.c_uint32 * 6)()
        _lib.check(self.lib.kmamd_yy_hint_stats(self.h, out), "kmamd_yy_hint_stats")
        return list(out)

    def sync(self):
        _lib.check(self.lib.kmamd_engine_sync(self.h), "kmamd_engine_sync")

    def filter_kind(self):
        """(kind, padded width): 1 register-resident filter, 2 LDS-streamed filter (features > 256), 0 exact kernels
        alone (kmamd_filter_kind)."""
        w = ctypes.c_uint32()
        kind = self.lib.kmamd_filter_kind(self.h, ctypes.byref(w))
        return kind, w.value

    def profile(self, on=True):
        self.lib.kmamd_profile_enable(self.h, 1 if on else 0)
        self.lib.kmamd_profile_reset(self.h)

    def profile_read(self):
        f, e, u = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = ctypes.c_uint32()
        self.lib.kmamd_profile_read(self.h, ctypes.byref(f), ctypes.byref(n), ctypes.byref(e), ctypes.byref(u))
        c = ctypes.c_double()
        self.lib.kmamd_profile_read_coarse(self.h, ctypes.byref(c))
        return {"filter_ms": f.value, "filter_launches": n.value, "exact_ms": e.value, "update_ms": u.value,
                "coarse_ms": c.value}
