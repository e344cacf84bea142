#!/usr/bin/env python
"""Per-tile phase times of lloyd_coarse2_kernel from the KMX_TRACE build (scripts/coarse_variants.sh
trace:"-DKMX_TRACE=1"): s_memtime stamps of one block's four waves at
  0 tile start | 1 accumulators seeded (biases arrived) | 2 last MFMA issued | 3 last MFMA's result readable
  | 4 bookkeeping done | 5 (odd tiles) barrier behind the super-tile passed.
Usage: KMCUDA_AMD_LIB=scratch/libs/libtrace.so python scripts/coarse_trace.py [rows]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy
import torch
from kmcuda_amd import _lib
from kmcuda_amd.distributed import HipBackend, ShardedLloyd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000000
d, k = 256, 1024
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
x = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    x[s:s + (1 << 20)].uniform_(0, 1, generator=gen)
b = HipBackend(x, k, "L2", device_index=0, row_cache=True)
loop = ShardedLloyd(b, n)
loop.set_centroids(x[torch.randperm(n, generator=gen, device=dev)[:k]].clone())
for _ in range(8):
    loop.step()
torch.cuda.synchronize()
L = _lib.lib()
words = 4 * 40 * 8
buf = (ctypes.c_ulonglong * words)()
L.kmamd_debug_trace.restype = ctypes.c_int
assert L.kmamd_debug_trace(buf, words) == 0
t = numpy.array(buf, dtype=numpy.uint64).reshape(4, 40, 8).astype(numpy.int64)
tiles = 2 * ((k + 63) // 64)
for w in range(4):
    tw = t[w, :tiles]
    seed = tw[:, 1] - tw[:, 0]
    mfma = tw[:, 2] - tw[:, 1]
    drain = tw[:, 3] - tw[:, 2]
    book = tw[:, 4] - tw[:, 3]
    nxt = numpy.zeros(tiles, numpy.int64)
    nxt[:-1] = tw[1:, 0] - tw[:-1, 4]      # book end -> next tile start (odd tiles: the barrier)
    total = tw[-1, 4] - tw[0, 0]
    print("wave %d: %d tiles, %d cycles total = %.1f per tile | seed %.0f  mfma-issue %.0f  drain %.0f  book %.0f  "
          "gap-even %.0f  gap-odd(barrier) %.0f" %
          (w, tiles, total, total / tiles, seed.mean(), mfma.mean(), drain.mean(), book.mean(),
           nxt[0:-1:2].mean(), nxt[1:-1:2].mean()))
print("per tile, wave 0 (seed, mfma, drain, book, gap):")
tw = t[0, :tiles]
for i in range(tiles):
    g = tw[i + 1, 0] - tw[i, 4] if i + 1 < tiles else 0
    print("  %2d: %5d %5d %5d %5d %5d" % (i, tw[i, 1] - tw[i, 0], tw[i, 2] - tw[i, 1], tw[i, 3] - tw[i, 2],
                                          tw[i, 4] - tw[i, 3], g))
