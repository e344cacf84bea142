// pymodule.cpp -- the `libKMCUDA` CPython module inside libKMCUDA.so (reference: src/python.cc:24-55
// PyInit_libKMCUDA, :159-410 py_kmeans_cuda, :412-632 py_knn_cuda): same function names, argument
// grammar, defaults, return shapes and exception mapping; the GIL is released around the two C calls as
// python.cc:357-363 / :595-599 do; returned arrays are fresh objects the caller alone references.
//
// Built differently from the reference on purpose: NOTHING here links against libpython or numpy's C
// API.  Python.h supplies types and macros only; every C-API entry point is looked up with dlsym() when
// the interpreter calls PyInit_libKMCUDA (the symbols are then in the process by definition), numpy is
// used through the object protocol (numpy.ascontiguousarray / numpy.empty + the buffer protocol).  The
// same .so therefore still loads into a plain C program that only wants kmeans_cuda() / knn_cuda().
#include <Python.h>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "../../include/kmcuda.h"

namespace {

struct Api {
  PyObject *(*ModuleCreate2)(PyModuleDef *, int);
  int (*ParseTupleAndKeywords)(PyObject *, PyObject *, const char *, char **, ...);
  PyObject *(*BuildValue)(const char *, ...);
  void (*ErrSetString)(PyObject *, const char *);
  PyObject *(*ErrOccurred)();
  void (*ErrClear)();
  PyObject *(*ImportModule)(const char *);
  PyObject *(*CallMethod)(PyObject *, const char *, const char *, ...);
  PyObject *(*GetAttrString)(PyObject *, const char *);
  int (*SetAttrString)(PyObject *, const char *, PyObject *);
  PyObject *(*ObjStr)(PyObject *);
  const char *(*UnicodeAsUTF8)(PyObject *);
  int (*GetBuffer)(PyObject *, Py_buffer *, int);
  void (*BufferRelease)(Py_buffer *);
  void (*DecRef)(PyObject *);
  void (*IncRef)(PyObject *);
  unsigned long long (*LongAsUnsignedLongLong)(PyObject *);
  long (*LongAsLong)(PyObject *);
  Py_ssize_t (*TupleSize)(PyObject *);
  PyObject *(*TupleGetItem)(PyObject *, Py_ssize_t);
  int (*IsTrue)(PyObject *);
  PyObject *(*NumberIndex)(PyObject *);
  const char *(*GetVersion)();
  PyThreadState *(*SaveThread)();
  void (*RestoreThread)(PyThreadState *);
  PyObject *None, *True_, *False_;
  PyObject *ValueError, *TypeError, *MemoryError, *RuntimeError, *AssertionError;
  PyObject *numpy;
} P;

template <typename T>
bool sym(T &fn, const char *name) {
  fn = reinterpret_cast<T>(dlsym(RTLD_DEFAULT, name));
  return fn != nullptr;
}
PyObject *exc(const char *name) {
  PyObject **p = reinterpret_cast<PyObject **>(dlsym(RTLD_DEFAULT, name));
  return p ? *p : nullptr;
}

bool bind_python() {
  bool ok = sym(P.ModuleCreate2, "PyModule_Create2") && sym(P.ParseTupleAndKeywords, "PyArg_ParseTupleAndKeywords") &&
            sym(P.BuildValue, "Py_BuildValue") && sym(P.ErrSetString, "PyErr_SetString") &&
            sym(P.ErrOccurred, "PyErr_Occurred") && sym(P.ErrClear, "PyErr_Clear") &&
            sym(P.ImportModule, "PyImport_ImportModule") && sym(P.CallMethod, "PyObject_CallMethod") &&
            sym(P.GetAttrString, "PyObject_GetAttrString") && sym(P.SetAttrString, "PyObject_SetAttrString") &&
            sym(P.ObjStr, "PyObject_Str") && sym(P.UnicodeAsUTF8, "PyUnicode_AsUTF8") &&
            sym(P.GetBuffer, "PyObject_GetBuffer") && sym(P.BufferRelease, "PyBuffer_Release") &&
            sym(P.DecRef, "Py_DecRef") && sym(P.IncRef, "Py_IncRef") &&
            sym(P.LongAsUnsignedLongLong, "PyLong_AsUnsignedLongLong") && sym(P.LongAsLong, "PyLong_AsLong") &&
            sym(P.TupleSize, "PyTuple_Size") && sym(P.TupleGetItem, "PyTuple_GetItem") &&
            sym(P.IsTrue, "PyObject_IsTrue") && sym(P.SaveThread, "PyEval_SaveThread") &&
            sym(P.RestoreThread, "PyEval_RestoreThread") && sym(P.NumberIndex, "PyNumber_Index") &&
            sym(P.GetVersion, "Py_GetVersion");
  P.None = reinterpret_cast<PyObject *>(dlsym(RTLD_DEFAULT, "_Py_NoneStruct"));
  P.True_ = reinterpret_cast<PyObject *>(dlsym(RTLD_DEFAULT, "_Py_TrueStruct"));
  P.False_ = reinterpret_cast<PyObject *>(dlsym(RTLD_DEFAULT, "_Py_FalseStruct"));
  P.ValueError = exc("PyExc_ValueError");
  P.TypeError = exc("PyExc_TypeError");
  P.MemoryError = exc("PyExc_MemoryError");
  P.RuntimeError = exc("PyExc_RuntimeError");
  P.AssertionError = exc("PyExc_AssertionError");
  return ok && P.None && P.True_ && P.False_ && P.ValueError && P.TypeError && P.MemoryError && P.RuntimeError &&
         P.AssertionError;
}

// owning reference
struct Ref {
  PyObject *o = nullptr;
  Ref() = default;
  explicit Ref(PyObject *p) : o(p) {}
  Ref(const Ref &) = delete;
  Ref &operator=(const Ref &) = delete;
  ~Ref() { if (o) P.DecRef(o); }
  void reset(PyObject *p) { if (o) P.DecRef(o); o = p; }
  PyObject *release() { PyObject *p = o; o = nullptr; return p; }
};
struct Buf {
  Py_buffer v;
  bool held = false;
  ~Buf() { if (held) P.BufferRelease(&v); }
  bool get(PyObject *o, bool writable) {
    held = P.GetBuffer(o, &v, PyBUF_ND | (writable ? PyBUF_WRITABLE : 0)) == 0;
    return held;
  }
};

bool is_tuple(PyObject *o) { return o && PyTuple_Check(o); }
bool is_str(PyObject *o) { return o && PyUnicode_Check(o); }
bool is_int(PyObject *o) { return o && PyLong_Check(o) && o != P.True_ && o != P.False_; }
// any object with __index__ (python.cc parses these with the "I" format: numpy.int64(k) is as good as k), but
// neither a bool nor a float.  false: no such integer -- a Python error may be pending (cleared by the caller)
bool as_index(PyObject *o, unsigned long long *out) {
  if (!o || o == P.True_ || o == P.False_) return false;   // (a float has no __index__: PyNumber_Index refuses it)
  Ref idx(P.NumberIndex(o));
  if (!idx.o) return false;
  *out = P.LongAsUnsignedLongLong(idx.o);
  return !P.ErrOccurred();
}

PyObject *fail(PyObject *type, const char *msg) {
  P.ErrSetString(type, msg);
  return nullptr;
}

// python.cc:365-409 / :601-631
PyObject *raise_for(int rc, const char *fn) {
  char msg[96];
  switch (rc) {
    case kmcudaInvalidArguments:
      snprintf(msg, sizeof(msg), "Invalid arguments were passed to %s", fn);
      return fail(P.ValueError, msg);
    case kmcudaNoSuchDevice: return fail(P.ValueError, "No such CUDA device exists");
    case kmcudaMemoryAllocationFailure: return fail(P.MemoryError, "Failed to allocate memory on GPU");
    case kmcudaMemoryCopyError: return fail(P.RuntimeError, "cudaMemcpy failed");
    case kmcudaRuntimeError:
      snprintf(msg, sizeof(msg), "%s failure (bug?)", fn);
      return fail(P.AssertionError, msg);
    default:
      snprintf(msg, sizeof(msg), "Unknown error code returned from %s", fn);
      return fail(P.AssertionError, msg);
  }
}

bool get_metric(PyObject *o, KMCUDADistanceMetric *m) {   // kmcuda.h:177-184
  *m = kmcudaDistanceMetricL2;
  if (!o || o == P.None) return true;
  if (!is_str(o)) { fail(P.TypeError, "\"metric\" must be either None or string."); return false; }
  const char *s = P.UnicodeAsUTF8(o);
  if (!s) return false;
  if (!strcmp(s, "euclidean") || !strcmp(s, "L2") || !strcmp(s, "l2")) return true;
  if (!strcmp(s, "cos") || !strcmp(s, "cosine") || !strcmp(s, "angular")) { *m = kmcudaDistanceMetricCosine; return true; }
  fail(P.ValueError, "Unknown metric. Supported values are \"L2\" and \"cos\".");
  return false;
}

bool dtype_is(PyObject *arr, const char *name) {
  Ref dt(P.GetAttrString(arr, "dtype"));
  if (!dt.o) { P.ErrClear(); return false; }
  Ref s(P.ObjStr(dt.o));
  const char *c = s.o ? P.UnicodeAsUTF8(s.o) : nullptr;
  return c && !strcmp(c, name);
}

// python.cc:120-157: a float16 array selects fp16x2, float64 is refused, anything else becomes float32.
// Returns a C-contiguous array (new reference) or null with the error set.
PyObject *get_samples_array(PyObject *samples, bool *fp16x2) {
  *fp16x2 = dtype_is(samples, "float16");
  if (dtype_is(samples, "float64")) return fail(P.TypeError, "\"samples\" must be a 2D float32 or float16 numpy array");
  PyObject *arr = P.CallMethod(P.numpy, "ascontiguousarray", "Os", samples, *fp16x2 ? "float16" : "float32");
  if (!arr) {
    P.ErrClear();
    return fail(P.TypeError, "\"samples\" must be a 2D float32 or float16 numpy array");
  }
  return arr;
}

// (pointer, device, shape[, ...]) tuples of the raw-pointer mode (python.cc:233-262, :470-500)
bool parse_ptr_tuple(PyObject *t, Py_ssize_t lo, Py_ssize_t hi, const char *err_len, uintptr_t *ptr, int *dev,
                     uint32_t *n, uint32_t *d, bool *fp16x2) {
  const Py_ssize_t sz = P.TupleSize(t);
  if (sz != lo && sz != hi) { fail(P.ValueError, err_len); return false; }
  PyObject *p0 = P.TupleGetItem(t, 0), *p1 = P.TupleGetItem(t, 1), *shape = P.TupleGetItem(t, 2);
  if (!is_int(p0)) { fail(P.ValueError, "\"samples\"[0] is not a pointer (integer)"); return false; }
  *ptr = (uintptr_t)P.LongAsUnsignedLongLong(p0);
  if (P.ErrOccurred()) return false;
  if (*ptr == 0) { fail(P.ValueError, "\"samples\"[0] is null"); return false; }
  *dev = (int)P.LongAsLong(p1);
  if (P.ErrOccurred()) return false;
  if (!is_tuple(shape) || (P.TupleSize(shape) != 2 && P.TupleSize(shape) != 3)) {
    fail(P.TypeError, "\"samples\"[2] must be a shape tuple");
    return false;
  }
  *n = (uint32_t)P.LongAsUnsignedLongLong(P.TupleGetItem(shape, 0));
  *d = (uint32_t)P.LongAsUnsignedLongLong(P.TupleGetItem(shape, 1));
  *fp16x2 = P.TupleSize(shape) == 3 && P.IsTrue(P.TupleGetItem(shape, 2)) == 1;
  return !P.ErrOccurred();
}

int copy_to_device(int device, void *dst, const void *src, size_t bytes) {
  if (hipSetDevice(device) != hipSuccess) return kmcudaNoSuchDevice;
  return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? kmcudaSuccess : kmcudaMemoryCopyError;
}

// ------------------------------------------------------------------------------------------------
PyObject *py_kmeans_cuda(PyObject *, PyObject *args, PyObject *kwargs) {
  uint32_t clusters = 0, seed = (uint32_t)time(nullptr), device = 0;
  int verbosity = 0, adflag = 0;
  float tolerance = .01f, yinyang_t = .1f;
  PyObject *samples_obj = nullptr, *init_obj = P.None, *metric_obj = P.None, *clusters_obj = nullptr;
  static const char *kwlist[] = {"samples", "clusters", "tolerance", "init", "yinyang_t", "metric",
                                 "average_distance", "seed", "device", "verbosity", nullptr};
  if (!P.ParseTupleAndKeywords(args, kwargs, "OO|fOfOpIIi", const_cast<char **>(kwlist), &samples_obj, &clusters_obj,
                               &tolerance, &init_obj, &yinyang_t, &metric_obj, &adflag, &seed, &device, &verbosity))
    return nullptr;
  {
    unsigned long long c = 0;
    if (!as_index(clusters_obj, &c)) {
      const bool negative_or_huge = P.ErrOccurred() && is_int(clusters_obj);   // an int, but not a uint64
      P.ErrClear();
      if (!negative_or_huge) return fail(P.TypeError, "\"clusters\" must be an integer");
      c = 0;
    }
    if (c < 2 || c >= 0xFFFFFFFFull)
      return fail(P.ValueError, "\"clusters\" must be greater than 1 and less than (1 << 32) - 1");
    clusters = (uint32_t)c;
  }
  // init: string | (string, m) | array (kmcuda.h:168-174, python.cc:196-217)
  KMCUDAInitMethod init = kmcudaInitMethodPlusPlus;
  uint32_t afkmc2_m = 0;
  PyObject *import_obj = nullptr;
  auto init_from_string = [&](PyObject *s) -> bool {
    const char *c = P.UnicodeAsUTF8(s);
    if (!c) return false;
    if (!strcmp(c, "kmeans++") || !strcmp(c, "k-means++")) init = kmcudaInitMethodPlusPlus;
    else if (!strcmp(c, "afkmc2") || !strcmp(c, "afk-mc2")) init = kmcudaInitMethodAFKMC2;
    else if (!strcmp(c, "random")) init = kmcudaInitMethodRandom;
    else {
      fail(P.ValueError, "Unknown centroids initialization method. Supported values are \"kmeans++\", \"random\" "
                         "and <numpy array>.");
      return false;
    }
    return true;
  };
  if (init_obj == P.None) {
  } else if (is_str(init_obj)) {
    if (!init_from_string(init_obj)) return nullptr;
  } else if (is_tuple(init_obj)) {
    if (P.TupleSize(init_obj) == 0 || P.TupleGetItem(init_obj, 0) == P.None || !is_str(P.TupleGetItem(init_obj, 0)))
      return fail(P.ValueError, "centroid initialization method may not be null.");
    if (!init_from_string(P.TupleGetItem(init_obj, 0))) return nullptr;
    if (init == kmcudaInitMethodAFKMC2 && P.TupleSize(init_obj) > 1) {
      unsigned long long m = 0;
      if (!as_index(P.TupleGetItem(init_obj, 1), &m) || m > 0xFFFFFFFFull) {
        P.ErrClear();
        return fail(P.ValueError, "afkmc2's m must be a non-negative integer below (1 << 32)");
      }
      afkmc2_m = (uint32_t)m;
    }
  } else {
    init = kmcudaInitMethodImport;
    import_obj = init_obj;
  }
  KMCUDADistanceMetric metric;
  if (!get_metric(metric_obj, &metric)) return nullptr;

  const float *samples = nullptr;
  float *centroids = nullptr;
  uint32_t *assignments = nullptr;
  uint32_t n = 0, d = 0;
  int device_ptrs = -1;
  bool fp16x2 = false;
  Ref samples_arr, centroids_arr, assignments_arr;
  Buf sbuf, cbuf, abuf;
  if (is_tuple(samples_obj)) {
    uintptr_t ptr = 0;
    if (!parse_ptr_tuple(samples_obj, 3, 5, "len(\"samples\") must be either 3 or 5", &ptr, &device_ptrs, &n, &d, &fp16x2))
      return nullptr;
    samples = reinterpret_cast<const float *>(ptr);
    if (P.TupleSize(samples_obj) == 5) {
      centroids = reinterpret_cast<float *>((uintptr_t)P.LongAsUnsignedLongLong(P.TupleGetItem(samples_obj, 3)));
      assignments = reinterpret_cast<uint32_t *>((uintptr_t)P.LongAsUnsignedLongLong(P.TupleGetItem(samples_obj, 4)));
      if (P.ErrOccurred()) return nullptr;
    }
  } else {
    samples_arr.reset(get_samples_array(samples_obj, &fp16x2));
    if (!samples_arr.o) return nullptr;
    if (!sbuf.get(samples_arr.o, false)) return nullptr;
    if (sbuf.v.ndim != 2) return fail(P.ValueError, "\"samples\" must be a 2D numpy array");
    n = (uint32_t)sbuf.v.shape[0];
    d = (uint32_t)sbuf.v.shape[1];
    if (fp16x2) {
      if (d % 2) return fail(P.ValueError, "the number of features must be even in fp16 mode");
      d /= 2;
    }
    samples = reinterpret_cast<const float *>(sbuf.v.buf);
  }
  if (d > 0xFFFFu) return fail(P.ValueError, "\"samples\": more than 65535 features is not supported");
  const uint32_t dwide = fp16x2 ? 2 * d : d;
  const size_t elem = fp16x2 ? 2 : 4;
  bool own_device_outputs = false;
  if (device_ptrs < 0) {
    centroids_arr.reset(P.CallMethod(P.numpy, "empty", "(II)s", clusters, dwide, fp16x2 ? "float16" : "float32"));
    assignments_arr.reset(P.CallMethod(P.numpy, "empty", "Is", n, "uint32"));
    if (!centroids_arr.o || !assignments_arr.o) return nullptr;
    if (!cbuf.get(centroids_arr.o, true) || !abuf.get(assignments_arr.o, true)) return nullptr;
    centroids = reinterpret_cast<float *>(cbuf.v.buf);
    assignments = reinterpret_cast<uint32_t *>(abuf.v.buf);
  } else if (!centroids) {
    // python.cc:300-318: the outputs are allocated on the caller's device and handed back as raw pointers
    if (hipSetDevice(device_ptrs) != hipSuccess) return fail(P.ValueError, "No such CUDA device exists");
    if (hipMalloc(reinterpret_cast<void **>(&centroids), (size_t)clusters * dwide * elem) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&assignments), (size_t)n * sizeof(uint32_t)) != hipSuccess) {
      if (centroids) (void)hipFree(centroids);
      return fail(P.MemoryError, "Failed to allocate memory on GPU");
    }
    own_device_outputs = true;
  }
  auto bail = [&](PyObject *r) -> PyObject * {
    if (own_device_outputs) { (void)hipFree(centroids); (void)hipFree(assignments); }
    return r;
  };
  if (import_obj) {   // python.cc:320-345
    Ref imp(P.CallMethod(P.numpy, "ascontiguousarray", "Os", import_obj, fp16x2 ? "float16" : "float32"));
    if (!imp.o) { P.ErrClear(); return bail(fail(P.TypeError, "\"init\" centroids must be a 2D numpy array")); }
    Buf ib;
    if (!ib.get(imp.o, false)) return bail(nullptr);
    if (ib.v.ndim != 2) return bail(fail(P.ValueError, "\"init\" centroids must be a 2D numpy array"));
    if ((uint32_t)ib.v.shape[0] != clusters)
      return bail(fail(P.ValueError, "\"init\" centroids shape[0] does not match the number of clusters"));
    if ((uint32_t)ib.v.shape[1] != dwide)
      return bail(fail(P.ValueError, "\"init\" centroids shape[1] does not match the number of features"));
    const size_t bytes = (size_t)clusters * dwide * elem;
    if (device_ptrs < 0) memcpy(centroids, ib.v.buf, bytes);
    else if (int rc = copy_to_device(device_ptrs, centroids, ib.v.buf, bytes)) return bail(raise_for(rc, "kmeans_cuda"));
  }
  float average_distance = 0;
  int result;
  {
    PyThreadState *ts = P.SaveThread();   // Py_BEGIN_ALLOW_THREADS, python.cc:357
    result = kmeans_cuda(init, &afkmc2_m, tolerance, yinyang_t, metric, n, (uint16_t)d, clusters, seed, device,
                         device_ptrs, fp16x2 ? 1 : 0, verbosity, samples, centroids, assignments,
                         adflag ? &average_distance : nullptr);
    P.RestoreThread(ts);
  }
  if (result != kmcudaSuccess) return bail(raise_for(result, "kmeans_cuda"));
  if (device_ptrs < 0) {
    // the buffers go first: the arrays leave with exactly one reference each
    if (cbuf.held) { P.BufferRelease(&cbuf.v); cbuf.held = false; }
    if (abuf.held) { P.BufferRelease(&abuf.v); abuf.held = false; }
    return adflag ? P.BuildValue("OOf", centroids_arr.o, assignments_arr.o, average_distance)
                  : P.BuildValue("OO", centroids_arr.o, assignments_arr.o);
  }
  const unsigned long long cp = (uintptr_t)centroids, ap = (uintptr_t)assignments;
  return adflag ? P.BuildValue("KKf", cp, ap, average_distance) : P.BuildValue("KK", cp, ap);
}

// ------------------------------------------------------------------------------------------------
PyObject *py_knn_cuda(PyObject *, PyObject *args, PyObject *kwargs) {
  uint32_t device = 0;
  int verbosity = 0;
  long k = 0;
  PyObject *samples_obj = nullptr, *centroids_obj = nullptr, *assignments_obj = nullptr, *metric_obj = P.None;
  static const char *kwlist[] = {"k", "samples", "centroids", "assignments", "metric", "device", "verbosity", nullptr};
  if (!P.ParseTupleAndKeywords(args, kwargs, "lOOO|OIi", const_cast<char **>(kwlist), &k, &samples_obj, &centroids_obj,
                               &assignments_obj, &metric_obj, &device, &verbosity))
    return nullptr;
  if (k <= 0 || k > 0xFFFF) return fail(P.ValueError, "\"k\" must be greater than 0 and less than (1 << 16)");
  KMCUDADistanceMetric metric;
  if (!get_metric(metric_obj, &metric)) return nullptr;
  const float *samples = nullptr, *centroids = nullptr;
  const uint32_t *assignments = nullptr;
  uint32_t *neighbors = nullptr;
  uint32_t n = 0, d = 0, clusters = 0;
  int device_ptrs = -1;
  bool fp16x2 = false;
  Ref samples_arr, centroids_arr, assignments_arr, neighbors_arr;
  Buf sbuf, cbuf, abuf, nbuf;
  bool own_device_output = false;
  if (is_tuple(samples_obj)) {
    uintptr_t ptr = 0;
    if (!parse_ptr_tuple(samples_obj, 3, 4, "len(\"samples\") must be either 3 or 4", &ptr, &device_ptrs, &n, &d, &fp16x2))
      return nullptr;
    samples = reinterpret_cast<const float *>(ptr);
    if (P.TupleSize(samples_obj) == 4) {
      neighbors = reinterpret_cast<uint32_t *>((uintptr_t)P.LongAsUnsignedLongLong(P.TupleGetItem(samples_obj, 3)));
      if (P.ErrOccurred()) return nullptr;
    }
    if (!is_tuple(centroids_obj)) return fail(P.ValueError, "\"centroids\" must be a tuple of length 2");
    if (P.TupleSize(centroids_obj) != 2) return fail(P.ValueError, "len(\"centroids\") must be 2");
    if (!is_int(P.TupleGetItem(centroids_obj, 0))) return fail(P.ValueError, "\"centroids\"[0] is not a pointer (integer)");
    centroids = reinterpret_cast<const float *>((uintptr_t)P.LongAsUnsignedLongLong(P.TupleGetItem(centroids_obj, 0)));
    if (!centroids) return fail(P.ValueError, "\"centroids\"[0] is null");
    clusters = (uint32_t)P.LongAsUnsignedLongLong(P.TupleGetItem(centroids_obj, 1));
    if (!is_int(assignments_obj)) return fail(P.ValueError, "\"assignments\" is not a pointer (integer)");
    assignments = reinterpret_cast<const uint32_t *>((uintptr_t)P.LongAsUnsignedLongLong(assignments_obj));
    if (P.ErrOccurred()) return nullptr;
  } else {
    samples_arr.reset(get_samples_array(samples_obj, &fp16x2));
    if (!samples_arr.o) return nullptr;
    if (!sbuf.get(samples_arr.o, false)) return nullptr;
    if (sbuf.v.ndim != 2) return fail(P.ValueError, "\"samples\" must be a 2D numpy array");
    n = (uint32_t)sbuf.v.shape[0];
    d = (uint32_t)sbuf.v.shape[1];
    if (fp16x2) {
      if (d % 2) return fail(P.ValueError, "the number of features must be even in fp16 mode");
      d /= 2;
    }
    samples = reinterpret_cast<const float *>(sbuf.v.buf);
    centroids_arr.reset(P.CallMethod(P.numpy, "ascontiguousarray", "Os", centroids_obj, fp16x2 ? "float16" : "float32"));
    if (!centroids_arr.o) { P.ErrClear(); return fail(P.TypeError, "\"centroids\" must be a 2D float32 or float16 numpy array"); }
    if (!cbuf.get(centroids_arr.o, false)) return nullptr;
    if (cbuf.v.ndim != 2) return fail(P.ValueError, "\"centroids\" must be a 2D numpy array");
    clusters = (uint32_t)cbuf.v.shape[0];
    if ((uint32_t)cbuf.v.shape[1] != (fp16x2 ? 2 * d : d))
      return fail(P.ValueError, "\"centroids\" must have same number of features as \"samples\" (shape[-1])");
    centroids = reinterpret_cast<const float *>(cbuf.v.buf);
    assignments_arr.reset(P.CallMethod(P.numpy, "ascontiguousarray", "Os", assignments_obj, "uint32"));
    if (!assignments_arr.o) { P.ErrClear(); return fail(P.TypeError, "\"assignments\" must be a 1D uint32 numpy array"); }
    if (!abuf.get(assignments_arr.o, false)) return nullptr;
    if (abuf.v.ndim != 1) return fail(P.ValueError, "\"assignments\" must be a 1D numpy array");
    if ((uint32_t)abuf.v.shape[0] != n) return fail(P.ValueError, "\"assignments\" must be of the same length as \"samples\"");
    assignments = reinterpret_cast<const uint32_t *>(abuf.v.buf);
  }
  if (d > 0xFFFFu) return fail(P.ValueError, "\"samples\": more than 65535 features is not supported");
  if (device_ptrs < 0) {
    neighbors_arr.reset(P.CallMethod(P.numpy, "empty", "(Il)s", n, k, "uint32"));
    if (!neighbors_arr.o || !nbuf.get(neighbors_arr.o, true)) return nullptr;
    neighbors = reinterpret_cast<uint32_t *>(nbuf.v.buf);
  } else if (!neighbors) {
    if (hipSetDevice(device_ptrs) != hipSuccess) return fail(P.ValueError, "No such CUDA device exists");
    if (hipMalloc(reinterpret_cast<void **>(&neighbors), (size_t)n * k * sizeof(uint32_t)) != hipSuccess)
      return fail(P.MemoryError, "Failed to allocate memory on GPU");
    own_device_output = true;
  }
  int result;
  {
    PyThreadState *ts = P.SaveThread();   // python.cc:595-599
    result = knn_cuda((uint16_t)k, metric, n, (uint16_t)d, clusters, device, device_ptrs, fp16x2 ? 1 : 0, verbosity,
                      samples, centroids, assignments, neighbors);
    P.RestoreThread(ts);
  }
  if (result != kmcudaSuccess) {
    if (own_device_output) (void)hipFree(neighbors);
    return raise_for(result, "knn_cuda");
  }
  if (device_ptrs < 0) {
    P.BufferRelease(&nbuf.v);
    nbuf.held = false;
    return neighbors_arr.release();
  }
  return P.BuildValue("K", (unsigned long long)(uintptr_t)neighbors);
}

const char module_doc[] = "MI355X-native K-means and K-nn (drop-in for src-d/kmcuda's libKMCUDA)";
const char kmeans_doc[] =
    "kmeans_cuda(samples, clusters, tolerance=0.01, init=\"k-means++\", yinyang_t=0.1, metric=\"L2\", "
    "average_distance=False, seed=time(), device=0, verbosity=0) -> (centroids, assignments[, average distance])";
const char knn_doc[] =
    "knn_cuda(k, samples, centroids, assignments, metric=\"L2\", device=0, verbosity=0) -> neighbors";

PyMethodDef module_functions[] = {
    {"kmeans_cuda", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(py_kmeans_cuda)),
     METH_VARARGS | METH_KEYWORDS, kmeans_doc},
    {"knn_cuda", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(py_knn_cuda)),
     METH_VARARGS | METH_KEYWORDS, knn_doc},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "libKMCUDA", module_doc, -1, module_functions,
                         nullptr, nullptr, nullptr, nullptr};

}  // namespace

extern "C" __attribute__((visibility("default"))) PyObject *PyInit_libKMCUDA(void) {
  if (!bind_python()) return nullptr;   // not inside a CPython process
  {
    // Python.h's inlined type checks and struct layouts are those of the interpreter this file was compiled
    // against (non-limited API): refuse any other minor version instead of misreading its objects
    const char *v = P.GetVersion();
    char want[16];
    snprintf(want, sizeof(want), "%d.%d.", PY_MAJOR_VERSION, PY_MINOR_VERSION);
    if (!v || strncmp(v, want, strlen(want)) != 0) {
      P.ErrSetString(exc("PyExc_ImportError") ? exc("PyExc_ImportError") : P.RuntimeError,
                     "libKMCUDA was built for CPython " PY_VERSION ": rebuild it (make -C kmcuda_amd/csrc) for this interpreter");
      return nullptr;
    }
  }
  PyObject *m = P.ModuleCreate2(&moduledef, PYTHON_API_VERSION);
  if (!m) return nullptr;
  P.numpy = P.ImportModule("numpy");
  if (!P.numpy) {
    P.DecRef(m);
    return nullptr;
  }
  P.SetAttrString(m, "supports_fp16", P.True_);   // fp16x2 boundary is always built (python.cc:52)
  return m;
}
