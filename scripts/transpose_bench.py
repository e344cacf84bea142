#!/usr/bin/env python
"""The supporting transpose (transpose.hip; reference: src/transpose.cu) against its HBM roofline: 2 x rows x cols x 4
bytes per call.   python scripts/transpose_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmcuda_amd.engine import Engine

dev = torch.device("cuda", 0)
for rows, cols in ((8000000, 256), (2000000, 1024), (1000003, 77)):
    src = torch.rand((rows, cols), device=dev)
    dst = torch.empty((cols, rows), device=dev)
    eng = Engine(rows, cols, 8, "L2", device=0)
    eng.transpose(src, rows, cols, dst)
    eng.sync()
    assert torch.equal(dst[:, :4096], src[:4096].t())
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.transpose(src, rows, cols, dst)
    eng.sync()
    dt = (time.perf_counter() - t0) / reps
    gb = 2.0 * rows * cols * 4 / 1e9
    print("transpose %9d x %4d: %.3f ms, %.1f GB per call => %.2f TB/s = %.2f of the 8 TB/s HBM peak" %
          (rows, cols, dt * 1e3, gb, gb / dt / 1e3, gb / dt / 8e3), flush=True)
    eng.close()
    del src, dst
