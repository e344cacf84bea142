"""Python mirror of the reference's libKMCUDA module (reference: src/python.cc).

Same functions, argument grammar, defaults, return shapes and exception mapping as
libKMCUDA.kmeans_cuda / knn_cuda (python.cc:159-410, :412-632), implemented over the C ABI of
include/kmcuda.h with ctypes (the reference binds the same two C functions from a CPython
extension; `import libKMCUDA` at the repo root re-exports this module under the reference's name).
"""
import ctypes
import operator
import time

import numpy

from . import _lib

supports_fp16 = True  # fp16x2 boundary: half buffers, fp32 arithmetic on the half values

_INIT = {"kmeans++": 1, "k-means++": 1, "afkmc2": 2, "afk-mc2": 2, "random": 0}  # kmcuda.h:168-174
_METRIC = {"euclidean": 0, "L2": 0, "l2": 0, "cos": 1, "cosine": 1, "angular": 1}  # kmcuda.h:177-184


def _raise_for(rc, fn):
    # python.cc:365-409 / :601-631
    if rc == 1:
        raise ValueError("Invalid arguments were passed to %s" % fn)
    if rc == 2:
        raise ValueError("No such CUDA device exists")
    if rc == 3:
        raise MemoryError("Failed to allocate memory on GPU")
    if rc == 5:
        raise RuntimeError("cudaMemcpy failed")
    if rc == 4:
        raise AssertionError("%s failure (bug?)" % fn)
    if rc != 0:
        raise AssertionError("Unknown error code returned from %s" % fn)


def _get_metric(metric):
    if metric is None:
        return 0
    if not isinstance(metric, str):
        raise TypeError("\"metric\" must be either None or string.")
    if metric not in _METRIC:
        raise ValueError("Unknown metric. Supported values are \"L2\" and \"cos\".")
    return _METRIC[metric]


def _get_samples(samples):
    """python.cc:120-157: float16 input selects fp16x2, anything castable becomes float32."""
    if isinstance(samples, numpy.ndarray) and samples.dtype == numpy.float16:
        arr, fp16x2 = numpy.ascontiguousarray(samples), True
    else:
        try:
            arr = numpy.ascontiguousarray(samples, dtype=numpy.float32)
        except (TypeError, ValueError):
            raise TypeError("\"samples\" must be a 2D float32 or float16 numpy array")
        if isinstance(samples, numpy.ndarray) and samples.dtype == numpy.float64:
            raise TypeError("\"samples\" must be a 2D float32 or float16 numpy array")
        fp16x2 = False
    if arr.ndim != 2:
        raise ValueError("\"samples\" must be a 2D numpy array")
    n, d = arr.shape
    if fp16x2:
        if d % 2:
            raise ValueError("the number of features must be even in fp16 mode")
        d //= 2
    return arr, fp16x2, n, d


def _ptr_tuple(samples, sizes):
    if len(samples) not in sizes:
        raise ValueError("len(\"samples\") must be either %d or %d" % sizes)
    ptr, dev, shape = samples[0], samples[1], samples[2]
    if not isinstance(ptr, int):
        raise ValueError("\"samples\"[0] is not a pointer (integer)")
    if ptr == 0:
        raise ValueError("\"samples\"[0] is null")
    if not isinstance(shape, tuple) or len(shape) not in (2, 3):
        raise TypeError("\"samples\"[2] must be a shape tuple")
    fp16x2 = bool(shape[2]) if len(shape) == 3 else False
    return ptr, int(dev), int(shape[0]), int(shape[1]), fp16x2


def kmeans_cuda(samples, clusters, tolerance=.01, init="k-means++", yinyang_t=.1, metric="L2",
                average_distance=False, seed=None, device=0, verbosity=0):
    """libKMCUDA.kmeans_cuda (python.cc:159-410, README "Python API")."""
    lib = _lib.lib()
    if seed is None:
        seed = int(time.time())
    # any object with __index__ (python.cc parses it with the "I" format: numpy.int64(k) works), not bool / float
    if isinstance(clusters, (bool, float)) or not hasattr(clusters, "__index__"):
        raise TypeError("\"clusters\" must be an integer")
    clusters = operator.index(clusters)
    init_centroids = None
    if init is None:
        init_id = 1
    elif isinstance(init, str):
        if init not in _INIT:
            raise ValueError("Unknown centroids initialization method. Supported values are "
                             "\"kmeans++\", \"random\" and <numpy array>.")
        init_id = _INIT[init]
    elif isinstance(init, tuple):
        if not init or init[0] is None:
            raise ValueError("centroid initialization method may not be null.")
        if init[0] not in _INIT:
            raise ValueError("Unknown centroids initialization method.")
        init_id = _INIT[init[0]]
        init_centroids = None
    else:
        init_id = 3
    afkmc2_m = ctypes.c_uint32(operator.index(init[1]) if isinstance(init, tuple) and len(init) > 1 and init_id == 2 else 0)
    metric_id = _get_metric(metric)
    if clusters < 2 or clusters >= 0xFFFFFFFF:
        raise ValueError("\"clusters\" must be greater than 1 and less than (1 << 32) - 1")
    device_ptrs = -1
    cen_ptr = asg_ptr = None
    keep = []
    if isinstance(samples, tuple):
        ptr, device_ptrs, n, d, fp16x2 = _ptr_tuple(samples, (3, 5))
        samples_ptr = ptr
        if len(samples) == 5:
            cen_ptr, asg_ptr = int(samples[3]), int(samples[4])
    else:
        arr, fp16x2, n, d = _get_samples(samples)
        keep.append(arr)
        samples_ptr = arr.ctypes.data
    if d > 0xFFFF:
        raise ValueError("\"samples\": more than %d features is not supported" % d)
    centroids = assignments = None
    if device_ptrs < 0:
        centroids = numpy.empty((clusters, d * 2 if fp16x2 else d), numpy.float16 if fp16x2 else numpy.float32)
        assignments = numpy.empty(n, numpy.uint32)
        cen_ptr, asg_ptr = centroids.ctypes.data, assignments.ctypes.data
    elif cen_ptr is None:
        import torch  # device-pointer mode: torch owns the output allocations on that GPU
        dev = torch.device("cuda", device_ptrs)
        cen_t = torch.empty((clusters, 2 * d if fp16x2 else d), dtype=torch.float16 if fp16x2 else torch.float32,
                            device=dev)
        asg_t = torch.empty(n, dtype=torch.int32, device=dev)
        _DEVICE_ALLOCS[cen_t.data_ptr()] = cen_t
        _DEVICE_ALLOCS[asg_t.data_ptr()] = asg_t
        cen_ptr, asg_ptr = cen_t.data_ptr(), asg_t.data_ptr()
    if init_id == 3:
        imp = numpy.ascontiguousarray(init, dtype=numpy.float16 if fp16x2 else numpy.float32)
        if imp.ndim != 2:
            raise ValueError("\"init\" centroids must be a 2D numpy array")
        if imp.shape[0] != clusters:
            raise ValueError("\"init\" centroids shape[0] does not match the number of clusters")
        if imp.shape[1] != (2 * d if fp16x2 else d):
            raise ValueError("\"init\" centroids shape[1] does not match the number of features")
        if device_ptrs < 0:
            centroids[...] = imp
        else:
            # python.cc:330-345: cudaMemcpy of the imported centroids into the (possibly caller-owned) buffer
            _raise_for(lib.kmamd_copy_to_device(device_ptrs, cen_ptr, imp.ctypes.data, imp.nbytes), "kmeans_cuda")
    avg = ctypes.c_float(0)
    rc = lib.kmeans_cuda(init_id, ctypes.byref(afkmc2_m), tolerance, yinyang_t, metric_id, n, d, clusters,
                         seed & 0xFFFFFFFF, device, device_ptrs, int(fp16x2), verbosity, samples_ptr, cen_ptr,
                         asg_ptr, ctypes.cast(ctypes.byref(avg), ctypes.c_void_p) if average_distance else None)
    _raise_for(rc, "kmeans_cuda")
    if device_ptrs < 0:
        return (centroids, assignments, avg.value) if average_distance else (centroids, assignments)
    return (cen_ptr, asg_ptr, avg.value) if average_distance else (cen_ptr, asg_ptr)


_DEVICE_ALLOCS = {}  # raw-pointer results stay alive until free_device_ptr()


def free_device_ptr(ptr):
    _DEVICE_ALLOCS.pop(ptr, None)


def knn_cuda(k, samples, centroids, assignments, metric="L2", device=0, verbosity=0):
    """libKMCUDA.knn_cuda (python.cc:412-632)."""
    lib = _lib.lib()
    metric_id = _get_metric(metric)
    if not isinstance(k, int) or k <= 0 or k > 0xFFFF:
        raise ValueError("\"k\" must be greater than 0 and less than (1 << 16)")
    device_ptrs = -1
    nb_ptr = None
    keep = []
    if isinstance(samples, tuple):
        ptr, device_ptrs, n, d, fp16x2 = _ptr_tuple(samples, (3, 4))
        samples_ptr = ptr
        if len(samples) == 4:
            nb_ptr = int(samples[3])
        if not isinstance(centroids, tuple):
            raise ValueError("\"centroids\" must be a tuple of length 2")
        if len(centroids) != 2:
            raise ValueError("len(\"centroids\") must be 2")
        if not isinstance(centroids[0], int):
            raise ValueError("\"centroids\"[0] is not a pointer (integer)")
        if centroids[0] == 0:
            raise ValueError("\"centroids\"[0] is null")
        cen_ptr, clusters = centroids[0], int(centroids[1])
        if not isinstance(assignments, int):
            raise ValueError("\"assignments\" is not a pointer (integer)")
        asg_ptr = assignments
    else:
        arr, fp16x2, n, d = _get_samples(samples)
        keep.append(arr)
        samples_ptr = arr.ctypes.data
        try:
            cen = numpy.ascontiguousarray(centroids, dtype=numpy.float16 if fp16x2 else numpy.float32)
        except (TypeError, ValueError):
            raise TypeError("\"centroids\" must be a 2D float32 or float16 numpy array")
        if cen.ndim != 2:
            raise ValueError("\"centroids\" must be a 2D numpy array")
        clusters = cen.shape[0]
        if cen.shape[1] != d * (2 if fp16x2 else 1):
            raise ValueError("\"centroids\" must have same number of features as \"samples\" (shape[-1])")
        try:
            asg = numpy.ascontiguousarray(assignments, dtype=numpy.uint32)
        except (TypeError, ValueError):
            raise TypeError("\"assignments\" must be a 1D uint32 numpy array")
        if asg.ndim != 1:
            raise ValueError("\"assignments\" must be a 1D numpy array")
        if asg.shape[0] != n:
            raise ValueError("\"assignments\" must be of the same length as \"samples\"")
        keep += [cen, asg]
        cen_ptr, asg_ptr = cen.ctypes.data, asg.ctypes.data
    if d > 0xFFFF:
        raise ValueError("\"samples\": more than %d features is not supported" % d)
    neighbors = None
    if device_ptrs < 0:
        neighbors = numpy.empty((n, k), numpy.uint32)
        nb_ptr = neighbors.ctypes.data
    elif nb_ptr is None:
        import torch
        nb_t = torch.empty((n, k), dtype=torch.int32, device=torch.device("cuda", device_ptrs))
        _DEVICE_ALLOCS[nb_t.data_ptr()] = nb_t
        nb_ptr = nb_t.data_ptr()
    rc = lib.knn_cuda(k, metric_id, n, d, clusters, device, device_ptrs, int(fp16x2), verbosity, samples_ptr,
                      cen_ptr, asg_ptr, nb_ptr)
    _raise_for(rc, "knn_cuda")
    return neighbors if device_ptrs < 0 else nb_ptr
