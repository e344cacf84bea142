#!/usr/bin/env python
"""Whole kmeans_cuda() call with HOST arrays (what a caller of the Python API pays, upload included):
   python scripts/host_call.py [--samples N] [--init random]"""
import argparse
import os
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=8000000)
    ap.add_argument("--features", type=int, default=256)
    ap.add_argument("--clusters", type=int, default=1024)
    ap.add_argument("--init", default="random")
    ap.add_argument("--repeat", type=int, default=2)
    args = ap.parse_args()
    import torch
    from kmcuda_amd import kmeans_cuda
    g = torch.Generator()
    g.manual_seed(1)
    x = torch.rand((args.samples, args.features), generator=g, dtype=torch.float32).numpy()
    print("rows: %.2f GB of pageable host memory" % (x.nbytes / 1e9), flush=True)
    # the raw copy, for scale
    dev = torch.device("cuda", 0)
    buf = torch.empty((args.samples, args.features), dtype=torch.float32, device=dev)
    for what, src in (("pageable", torch.from_numpy(x)),):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        buf.copy_(src)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("torch copy_ of the %s array: %.3f s = %.1f GB/s" % (what, dt, x.nbytes / dt / 1e9), flush=True)
    del buf
    torch.cuda.empty_cache()
    for r in range(args.repeat):
        t0 = time.perf_counter()
        c, a = kmeans_cuda(x, args.clusters, init=args.init, seed=777, tolerance=0.01, yinyang_t=0, device=1, verbosity=0)
        dt = time.perf_counter() - t0
        print("kmeans_cuda(host array) call %d: %.3f s" % (r, dt), flush=True)


if __name__ == "__main__":
    main()
