"""TEST INFRASTRUCTURE ONLY -- ctypes loader for the CPU oracle (oracle/kmcuda_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (kmcuda_amd/) never does; it fails loudly when its HIP library is missing.

Parity pinning: the reference is CUDA-only, so there is no oracle/_ref build; the oracle is
pinned on the reference's own known-answer tests (tests/test_oracle_pins.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkmcuda_oracle.so")

L2, COS = 0, 1
INIT_RANDOM, INIT_PLUSPLUS, INIT_AFKMC2, INIT_IMPORT = 0, 1, 2, 3
_INITS = {"random": INIT_RANDOM, "kmeans++": INIT_PLUSPLUS, "k-means++": INIT_PLUSPLUS, "afkmc2": INIT_AFKMC2, "afk-mc2": INIT_AFKMC2}
_METRICS = {"L2": L2, "l2": L2, "euclidean": L2, "cos": COS, "cosine": COS, "angular": COS}


def build(force=False):
    src = os.path.join(_HERE, "kmcuda_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libkmcuda_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
u32, i32, f32 = ctypes.c_uint32, ctypes.c_int, ctypes.c_float


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.kmo_fma_rd.restype = f32
        L.kmo_fma_rd.argtypes = [f32, f32, f32]
        L.kmo_fma_rd_portable.restype = f32
        L.kmo_fma_rd_portable.argtypes = [f32, f32, f32]
        L.kmo_have_avx512.restype = i32
        L.kmo_num_threads.restype = i32
        L.kmo_kahan_dot.restype = f32
        L.kmo_kahan_dot.argtypes = [_f32p, _f32p, u32]
        L.kmo_distance.restype = f32
        L.kmo_distance.argtypes = [i32, _f32p, _f32p, u32]
        L.kmo_sum_squares.argtypes = [i32, u32, u32, _f32p, _f32p]
        L.kmo_lloyd_assign.argtypes = [i32, u32, u32, u32, _f32p, _f32p, _u32p, _u32p, _u32p]
        L.kmo_adjust.argtypes = [i32, u32, u32, u32, _f32p, _u32p, _u32p, _f32p, _u32p]
        L.kmo_yy_init.argtypes = [i32, u32, u32, u32, u32, _f32p, _f32p, _u32p, _u32p, _f32p]
        L.kmo_yy_calc_drifts.argtypes = [i32, u32, u32, _f32p, _f32p]
        L.kmo_yy_group_max_drifts.argtypes = [u32, u32, u32, _u32p, _f32p]
        L.kmo_yy_global_filter.restype = u32
        L.kmo_yy_global_filter.argtypes = [i32, u32, u32, u32, u32, _f32p, _f32p, _u32p, _f32p,
                                           _u32p, _u32p, _f32p, _u32p]
        L.kmo_yy_local_filter.restype = u32
        L.kmo_yy_local_filter.argtypes = [i32, u32, u32, u32, u32, _f32p, _u32p, u32, _f32p, _u32p,
                                          _f32p, _u32p, _f32p]
        L.kmo_init_centroids.restype = i32
        L.kmo_init_centroids.argtypes = [i32, i32, u32, u32, u32, u32, _f32p, _f32p]
        L.kmo_average_distance.restype = f32
        L.kmo_average_distance.argtypes = [i32, u32, u32, _f32p, _f32p, _u32p]
        L.kmo_kmeans.restype = i32
        L.kmo_kmeans.argtypes = [i32, f32, f32, i32, u32, u32, u32, u32, _f32p, _f32p, _u32p, _f32p,
                                 _u32p, u32, _u32p]
        L.kmo_set_fp16_storage.argtypes = [i32]
        L.kmo_set_fp16_mode.argtypes = [i32]
        L.kmo_h_rn.restype = f32
        L.kmo_h_rn.argtypes = [ctypes.c_double]
        L.kmo_h_from_int_rd.restype = f32
        L.kmo_h_from_int_rd.argtypes = [ctypes.c_longlong]
        L.kmo_quantize_half.restype = f32
        L.kmo_quantize_half.argtypes = [f32]
        L.kmo_knn_inverse.argtypes = [u32, u32, _u32p, _u32p, _u32p]
        L.kmo_knn_radiuses.argtypes = [i32, u32, u32, u32, _f32p, _f32p, _u32p, _u32p, _f32p]
        L.kmo_knn_cluster_distances.argtypes = [i32, u32, u32, _f32p, _f32p]
        L.kmo_knn.restype = i32
        L.kmo_knn.argtypes = [u32, i32, u32, u32, u32, _f32p, _f32p, _u32p, _u32p, _u64p]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _up(a):
    return a.ctypes.data_as(_u32p)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _metric(m):
    return m if isinstance(m, int) else _METRICS[m]


def fma_rd(a, b, c):
    return lib().kmo_fma_rd(a, b, c)


def fma_rd_portable(a, b, c):
    return lib().kmo_fma_rd_portable(a, b, c)


def sum_squares(centroids, metric=L2):
    c = _c32(centroids)
    out = np.empty(c.shape[0], np.float32)
    lib().kmo_sum_squares(_metric(metric), c.shape[0], c.shape[1], _fp(c), _fp(out))
    return out


def distance(a, b, metric=L2):
    a, b = _c32(a), _c32(b)
    return lib().kmo_distance(_metric(metric), _fp(a), _fp(b), a.shape[0])


def lloyd_assign(samples, centroids, assignments=None, metric=L2):
    """One kmeans_assign_lloyd pass.  Returns (assignments, assignments_prev, changed)."""
    x, c = _c32(samples), _c32(centroids)
    n = x.shape[0]
    asg = (np.full(n, 0xFFFFFFFF, np.uint32) if assignments is None
           else np.array(assignments, dtype=np.uint32, copy=True))
    prev = np.full(n, 0xFFFFFFFF, np.uint32)
    changed = u32(0)
    lib().kmo_lloyd_assign(_metric(metric), n, x.shape[1], c.shape[0], _fp(x), _fp(c), _up(asg),
                           _up(prev), ctypes.byref(changed))
    return asg, prev, changed.value


def adjust(samples, prev, cur, centroids, ccounts, metric=L2):
    """kmeans_adjust.  Returns (new_centroids, new_ccounts)."""
    x = _c32(samples)
    c = np.array(centroids, dtype=np.float32, copy=True)
    cc = np.array(ccounts, dtype=np.uint32, copy=True)
    p = np.ascontiguousarray(prev, dtype=np.uint32)
    a = np.ascontiguousarray(cur, dtype=np.uint32)
    lib().kmo_adjust(_metric(metric), x.shape[0], x.shape[1], c.shape[0], _fp(x), _up(p), _up(a),
                     _fp(c), _up(cc))
    return c, cc


def init_centroids(samples, clusters, init="kmeans++", seed=0, metric=L2):
    x = _c32(samples)
    c = np.empty((clusters, x.shape[1]), np.float32)
    rc = lib().kmo_init_centroids(_INITS[init], _metric(metric), x.shape[0], x.shape[1], clusters,
                                  seed, _fp(x), _fp(c))
    if rc:
        raise ValueError("kmo_init_centroids failed: %d" % rc)
    return c


def kmeans(samples, clusters, tolerance=0.01, init="kmeans++", yinyang_t=0.1, metric="L2",
           average_distance=False, seed=0, half2=False):
    """Mirror of libKMCUDA.kmeans_cuda on the CPU oracle.
    Returns (centroids, assignments, iteration_log[, average_distance]).
    float16 samples select fp16x2: by default this repository's product semantics (fp32 arithmetic on the
    half values, centroids rounded to half after every update); half2=True the REFERENCE's half2 arithmetic
    (fp_abstraction.h:100-182, kmo_set_fp16_mode(2)).  Centroids come back as float16."""
    fp16 = isinstance(samples, np.ndarray) and samples.dtype == np.float16
    if half2 and not fp16:
        raise ValueError("half2 arithmetic needs float16 samples")
    lib().kmo_set_fp16_mode((2 if half2 else 1) if fp16 else 0)
    try:
        out = _kmeans(samples, clusters, tolerance, init, yinyang_t, metric, average_distance, seed)
    finally:
        lib().kmo_set_fp16_mode(0)
    if fp16:
        out = (out[0].astype(np.float16),) + out[1:]
    return out


def _kmeans(samples, clusters, tolerance, init, yinyang_t, metric, average_distance, seed):
    x = _c32(samples)
    n, d = x.shape
    if isinstance(init, np.ndarray):
        cen = np.array(init, dtype=np.float32, copy=True)
        method = INIT_IMPORT
    else:
        cen = np.empty((clusters, d), np.float32)
        afk_m = 0
        if isinstance(init, tuple):   # ("afkmc2", m), kmcuda.h:168-174
            init, afk_m = init[0], int(init[1])
        method = _INITS[init]
        lib().kmo_set_afkmc2_m.argtypes = [u32]
        lib().kmo_set_afkmc2_m(afk_m)
    asg = np.empty(n, np.uint32)
    log = np.zeros(4096, np.uint32)
    nlog = u32(0)
    avg = f32(0)
    rc = lib().kmo_kmeans(method, tolerance, yinyang_t, _metric(metric), n, d, clusters, seed,
                          _fp(x), _fp(cen), _up(asg),
                          ctypes.byref(avg) if average_distance else None, _up(log), log.size,
                          ctypes.byref(nlog))
    if rc:
        raise ValueError("kmo_kmeans failed: %d" % rc)
    out = (cen, asg, log[:nlog.value].copy())
    if average_distance:
        out += (avg.value,)
    return out


def knn(k, samples, centroids, assignments, metric="L2", half2=False):
    """knn_cuda on the CPU oracle.  half2=True (float16 samples and centroids): the reference's half2 arithmetic
    (fp_abstraction.h:100-182) for the radii, the centroid distances and every candidate distance."""
    if half2 and not (samples.dtype == np.float16 and centroids.dtype == np.float16):
        raise ValueError("half2 arithmetic needs float16 samples and centroids")
    x, c = _c32(samples), _c32(centroids)
    a = np.ascontiguousarray(assignments, dtype=np.uint32)
    nb = np.empty((x.shape[0], k), np.uint32)
    calced = ctypes.c_uint64(0)
    lib().kmo_set_fp16_mode(2 if half2 else 0)
    try:
        rc = lib().kmo_knn(k, _metric(metric), x.shape[0], x.shape[1], c.shape[0], _fp(x), _fp(c),
                           _up(a), _up(nb), ctypes.byref(calced))
    finally:
        lib().kmo_set_fp16_mode(0)
    if rc:
        raise ValueError("kmo_knn failed: %d" % rc)
    return nb, calced.value


def yy_init(samples, centroids, assignments, groups, n_groups, metric=L2):
    """kmeans_yy_init.  Returns bounds, shape (G+1, N): [0] upper, [1+g] lower to group g."""
    x, c = _c32(samples), _c32(centroids)
    a = np.ascontiguousarray(assignments, dtype=np.uint32)
    g = np.ascontiguousarray(groups, dtype=np.uint32)
    bounds = np.empty((n_groups + 1, x.shape[0]), np.float32)
    lib().kmo_yy_init(_metric(metric), x.shape[0], x.shape[1], c.shape[0], n_groups, _fp(x), _fp(c),
                      _up(a), _up(g), _fp(bounds))
    return bounds


def yy_drifts(old_centroids, centroids, groups, n_groups, metric=L2):
    """kmeans_yy_calc_drifts + kmeans_yy_find_group_max_drifts.  Returns the reference's drifts
    buffer (K*D + K floats; [0, G) = group maxima overlaying the old centroids, [K*D, K*D+K) =
    per-centroid drifts)."""
    oc, c = _c32(old_centroids), _c32(centroids)
    k, d = c.shape
    drifts = np.empty(k * d + k, np.float32)
    drifts[:k * d] = oc.ravel()
    g = np.ascontiguousarray(groups, dtype=np.uint32)
    lib().kmo_yy_calc_drifts(_metric(metric), d, k, _fp(c), _fp(drifts))
    lib().kmo_yy_group_max_drifts(d, k, n_groups, _up(g), _fp(drifts))
    return drifts


def yy_filters(samples, centroids, groups, n_groups, drifts, assignments, bounds, metric=L2):
    """kmeans_yy_global_filter + kmeans_yy_local_filter on copies.
    Returns (assignments, assignments_prev, bounds, passed (sorted), changed)."""
    x, c = _c32(samples), _c32(centroids)
    n = x.shape[0]
    g = np.ascontiguousarray(groups, dtype=np.uint32)
    a = np.array(assignments, dtype=np.uint32, copy=True)
    prev = np.empty(n, np.uint32)
    b = np.array(bounds, dtype=np.float32, copy=True)
    dr = np.ascontiguousarray(drifts, dtype=np.float32)
    passed = np.empty(n, np.uint32)
    m = _metric(metric)
    npassed = lib().kmo_yy_global_filter(m, n, x.shape[1], c.shape[0], n_groups, _fp(x), _fp(c), _up(g),
                                         _fp(dr), _up(a), _up(prev), _fp(b), _up(passed))
    changed = lib().kmo_yy_local_filter(m, n, x.shape[1], c.shape[0], n_groups, _fp(x), _up(passed), npassed,
                                        _fp(c), _up(g), _fp(dr), _up(a), _fp(b))
    return a, prev, b, np.sort(passed[:npassed]), changed


def set_afkmc2_seeding(curand):
    """AFK-MC2's seed scrambling: False = rocRAND's constants (default), True = cuRAND's as quoted in kmcuda_oracle.c."""
    lib().kmo_set_afkmc2_seeding.argtypes = [i32]
    lib().kmo_set_afkmc2_seeding(1 if curand else 0)


def xorwow_draws(seed, subsequence, offset, n, curand_seeding=False):
    """n raw 32-bit draws of the XORWOW stream (seed, subsequence, offset) of AFK-MC2's generator (kmeans.cu:112-116:
    curand_init(seed, thread, step)) under rocRAND's seed scrambling (default) or cuRAND's (kmcuda_oracle.c: the one
    place the two libraries' XORWOW differ)."""
    out = np.empty(n, np.uint32)
    f = lib().kmo_xorwow_draws
    f.restype = None
    f.argtypes = [i32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, u32, _u32p]
    f(1 if curand_seeding else 0, seed, subsequence, offset, n, _up(out))
    return out
