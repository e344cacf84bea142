"""`from libKMCUDA import kmeans_cuda, knn_cuda, supports_fp16` -- the reference's module name
(src/python.cc:24-54).  The functions are the NATIVE CPython module that lives inside
kmcuda_amd/libKMCUDA.so (PyInit_libKMCUDA, kmcuda_amd/csrc/pymodule.cpp: GIL released around the C
calls, fresh result arrays); this file only points the import machinery at that .so -- copying or
symlinking kmcuda_amd/libKMCUDA.so onto sys.path does the same without it.  The ctypes mirror
kmcuda_amd.api (same grammar, used by the test-suite) stays available under its own name."""
import importlib.machinery
import importlib.util
import os
import sys

_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kmcuda_amd", "libKMCUDA.so")
if not os.path.exists(_path):
    raise ImportError("libKMCUDA: %s is missing -- build it with `make -C kmcuda_amd/csrc`" % _path)
_loader = importlib.machinery.ExtensionFileLoader("libKMCUDA", _path)
_spec = importlib.util.spec_from_file_location("libKMCUDA", _path, loader=_loader)
_native = importlib.util.module_from_spec(_spec)
_loader.exec_module(_native)
kmeans_cuda = _native.kmeans_cuda
knn_cuda = _native.knn_cuda
supports_fp16 = _native.supports_fp16
