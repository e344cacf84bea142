#!/usr/bin/env python
"""BASELINE config B/A runs through the drop-in boundary with device-resident inputs:
   python scripts/config_b.py [--samples N] [--yinyang T] [--tolerance X] [--metric cos] [--dtype f16]
Prints wall time of the whole kmeans_cuda() call and the iteration lines."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=8000000)
    ap.add_argument("--features", type=int, default=256)
    ap.add_argument("--clusters", type=int, default=1024)
    ap.add_argument("--yinyang", type=float, default=0.1)
    ap.add_argument("--tolerance", type=float, default=0.01)
    ap.add_argument("--init", default="random")
    ap.add_argument("--verbosity", type=int, default=1)
    ap.add_argument("--metric", default="L2", choices=["L2", "cos"], help="cos: rows are normalised to unit length")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"], help="f16: the fp16x2 path (rows as halves)")
    ap.add_argument("--data", default="uniform", choices=["uniform", "gaussian"],
                    help="uniform [0,1) rows (README.md:206-207) or a mixture of `clusters` unit Gaussians "
                         "with centres uniform in [0,10)^D (SURVEY 8d: pruning is meaningful)")
    args = ap.parse_args()
    import torch
    from kmcuda_amd import kmeans_cuda
    from kmcuda_amd.api import _DEVICE_ALLOCS
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    x = torch.empty((args.samples, args.features), dtype=torch.float32, device=dev)
    centres = torch.rand((args.clusters, args.features), device=dev, generator=gen) * 10.0
    for s in range(0, args.samples, 1 << 20):
        e = min(args.samples, s + (1 << 20))
        if args.data == "uniform":
            x[s:e].uniform_(0.0, 1.0, generator=gen)
        else:
            lab = torch.randint(0, args.clusters, (e - s,), device=dev, generator=gen)
            x[s:e].normal_(0.0, 1.0, generator=gen)
            x[s:e] += centres[lab]
    if args.metric == "cos":
        for s in range(0, args.samples, 1 << 20):
            e = min(args.samples, s + (1 << 20))
            x[s:e] /= x[s:e].norm(dim=1, keepdim=True)
    shape = (args.samples, args.features)
    if args.dtype == "f16":
        x = x.to(torch.float16)
        shape = (args.samples, args.features // 2, True)   # fp16x2: features counted in half2 pairs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cptr, aptr = kmeans_cuda((x.data_ptr(), 0, shape), args.clusters, init=args.init,
                             seed=777, tolerance=args.tolerance, yinyang_t=args.yinyang, metric=args.metric,
                             device=1, verbosity=args.verbosity)
    dt = time.perf_counter() - t0
    asg = _DEVICE_ALLOCS[aptr]
    import ctypes
    from kmcuda_amd import _lib
    it, loop_s = ctypes.c_uint32(), ctypes.c_double()
    _lib.lib().kmamd_last_run_stats(ctypes.byref(it), ctypes.byref(loop_s), None, None, None)
    print("kmeans_cuda wall: %.3f s (%d iterations, %.3f s in the iteration loop); clusters used: %d" %
          (dt, it.value, loop_s.value, int(torch.unique(asg).numel())), flush=True)


if __name__ == "__main__":
    main()
