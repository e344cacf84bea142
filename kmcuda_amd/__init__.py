"""kmcuda_amd -- MI355X (gfx950) native implementation of kmcuda's distance/assignment hot path.

Drop-in surface (mirrors the reference's `libKMCUDA` Python module, src/python.cc):
    from kmcuda_amd import kmeans_cuda, knn_cuda, supports_fp16
Step-level surface for row-sharded multi-process operation: kmcuda_amd.engine.Engine,
kmcuda_amd.distributed.
"""
from .api import kmeans_cuda, knn_cuda, supports_fp16  # noqa: F401
