"""`from libKMCUDA import kmeans_cuda, knn_cuda, supports_fp16` -- the reference's module name
(src/python.cc:24-54) over this repository's implementation (kmcuda_amd/api.py -> libKMCUDA.so)."""
from kmcuda_amd.api import kmeans_cuda, knn_cuda, supports_fp16  # noqa: F401
