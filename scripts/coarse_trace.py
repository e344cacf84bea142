#!/usr/bin/env python
"""Timeline of ONE physical CU during lloyd_coarse2_kernel, from the KMX_TRACE build
(scripts/coarse_variants.sh trace:"-DKMX_TRACE=1"): every wave that ran on XCC 0 / SE 0 / SH 0 / CU 0
records s_memtime stamps per tile: 0 start | 1 accumulators seeded | 2 last MFMA issued | 3 bookkeeping
issued | (odd tiles) 4 own DMA landed (vmcnt 0) | 5 barrier passed.
Usage: KMCUDA_AMD_LIB=scratch/libs/libtrace.so python scripts/coarse_trace.py [rows] [out.npy]"""
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
L = _lib.lib()
REC, MAXR = 200, 1024
buf = (ctypes.c_ulonglong * (REC * MAXR))()
nrec = ctypes.c_uint()
L.kmamd_debug_trace.restype = ctypes.c_int
for _ in range(8):
    loop.step()
torch.cuda.synchronize()
L.kmamd_debug_trace(buf, REC * MAXR, ctypes.byref(nrec))     # rearm
loop.step()
torch.cuda.synchronize()
assert L.kmamd_debug_trace(buf, REC * MAXR, ctypes.byref(nrec)) == 0
nr = min(nrec.value, MAXR)
t = numpy.array(buf, dtype=numpy.uint64).reshape(MAXR, REC)[:nr].astype(numpy.int64)
if len(sys.argv) > 2:
    numpy.save(sys.argv[2], t)
tiles = 2 * ((k + 63) // 64)
print("records (waves on the traced CU): %d" % nr)
if nr == 0:
    sys.exit(0)
st = t[:, 2:2 + tiles * 6].reshape(nr, tiles, 6)
t0 = st[:, 0, 0].min()
blk, wave, hwid = t[:, 0], t[:, 1] & 0xFF, t[:, 1] >> 8
simd = (hwid >> 4) & 3
dur = st[:, tiles - 1, 3] - st[:, 0, 0]
print("kernel span on this CU: %d cycles; wave life: mean %d  min %d  max %d" %
      (st[:, tiles - 1, 3].max() - t0, dur.mean(), dur.min(), dur.max()))
seed = st[:, :, 1] - st[:, :, 0]
mfma = st[:, :, 2] - st[:, :, 1]
book = st[:, :, 3] - st[:, :, 2]
odd = numpy.arange(1, tiles, 2)
vm = st[:, odd, 4] - st[:, odd, 3]
bar = st[:, odd, 5] - st[:, odd, 4]
print("means over all waves: seed %.0f | mfma-issue even %.0f odd %.0f | book(+drain) even %.0f odd %.0f | "
      "vmcnt(0) wait %.0f | barrier wait %.0f" %
      (seed.mean(), mfma[:, 0::2].mean(), mfma[:, 1::2].mean(), book[:, 0::2].mean(), book[:, 1::2].mean(),
       vm.mean(), bar.mean()))
late = slice(8, tiles)
print("tiles 8..: seed %.0f | mfma-issue even %.0f odd %.0f | book even %.0f odd %.0f | vmcnt %.0f | barrier %.0f" %
      (seed[:, late].mean(), mfma[:, 8::2].mean(), mfma[:, 9::2].mean(), book[:, 8::2].mean(), book[:, 9::2].mean(),
       vm[:, 4:].mean(), bar[:, 4:].mean()))
# SIMD 0 of the CU: who holds the matrix pipe when (mfma-issue intervals), first 40000 cycles after a mid point
mid = t0 + (st[:, tiles - 1, 3].max() - t0) // 2
sel = numpy.nonzero(simd == 0)[0]
ev = []
for i in sel:
    for tt in range(tiles):
        a, bb = st[i, tt, 1], st[i, tt, 2]
        if bb > mid and a < mid + 30000:
            ev.append((a - mid, bb - mid, blk[i], tt))
ev.sort()
print("SIMD 0, MFMA-issue intervals around the kernel's middle (start, end, block, tile):")
for e in ev[:40]:
    print("  %7d %7d  block %6d tile %2d" % e)
# pipe occupancy estimate on SIMD 0: union length of mfma-issue intervals / span
iv = sorted((st[i, tt, 1], st[i, tt, 2]) for i in sel for tt in range(tiles))
cover, cur_a, cur_b = 0, None, None
for a, bb in iv:
    if cur_b is None or a > cur_b:
        if cur_b is not None:
            cover += cur_b - cur_a
        cur_a, cur_b = a, bb
    else:
        cur_b = max(cur_b, bb)
cover += cur_b - cur_a
span = max(b for _, b in iv) - min(a for a, _ in iv)
print("SIMD 0: some wave in its MFMA-issue phase %.1f %% of the span; both-idle %.1f %%" %
      (100.0 * cover / span, 100.0 - 100.0 * cover / span))
