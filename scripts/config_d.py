#!/usr/bin/env python
"""BASELINE config D shape: k-NN (k=10) over N x 256 fp32 with precomputed K=1024 clusters, through
knn_cuda() with device-resident inputs.  Data: mixture of `clusters` unit Gaussians with centres
uniform in [0,10)^D so that the cluster pruning is meaningful, or uniform rows."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1000000)
    ap.add_argument("--features", type=int, default=256)
    ap.add_argument("--clusters", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--data", default="gaussian", choices=["uniform", "gaussian"])
    ap.add_argument("--sigma", type=float, default=1.0)
    ap.add_argument("--check", type=int, default=0, help="verify this many rows against a brute-force torch search")
    ap.add_argument("--shard", default="", help="i/n: what rank i of an n-GPU search does -- the whole corpus resident, "
                                                 "1/n of the queries (KMCUDA_AMD_KNN_SHARD); BASELINE config D is "
                                                 "--samples 8000000 --shard 0/8")
    ap.add_argument("--init", default="random")
    args = ap.parse_args()
    if args.shard:
        os.environ["KMCUDA_AMD_KNN_SHARD"] = args.shard
    import torch
    from kmcuda_amd import kmeans_cuda, knn_cuda
    from kmcuda_amd.api import _DEVICE_ALLOCS
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    n, d, K = args.samples, args.features, args.clusters
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    centres = torch.rand((K, d), device=dev, generator=gen) * 10.0
    for s in range(0, n, 1 << 20):
        e = min(n, s + (1 << 20))
        if args.data == "uniform":
            x[s:e].uniform_(0.0, 1.0, generator=gen)
        else:
            lab = torch.randint(0, K, (e - s,), device=dev, generator=gen)
            x[s:e].normal_(0.0, args.sigma, generator=gen)
            x[s:e] += centres[lab]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cptr, aptr = kmeans_cuda((x.data_ptr(), 0, (n, d)), K, init=args.init, seed=777, tolerance=0.01, yinyang_t=0,
                             device=1, verbosity=0)
    t1 = time.perf_counter()
    nbuf = torch.full((n, args.k), -1, dtype=torch.int32, device=dev)   # rows outside the shard stay 0xFFFFFFFF
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    knn_cuda(args.k, (x.data_ptr(), 0, (n, d), nbuf.data_ptr()), (cptr, K), aptr, device=1, verbosity=1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    done = int((nbuf[:, 0] != -1).sum().item())
    print("kmeans_cuda %.3f s; knn_cuda %.3f s for %d of %d queries => %.3e neighbour lists/s" %
          (t1 - t0, t2 - t1, done, n, done / (t2 - t1)), flush=True)
    if args.check:
        nb = nbuf
        have = torch.nonzero(nbuf[:, 0] != -1).ravel()
        rows = have[torch.randint(0, have.numel(), (args.check,), device=dev, generator=gen)]
        bad = 0
        for r in rows.tolist():
            dist = ((x - x[r]) ** 2).sum(1)
            dist[r] = float("inf")
            best = torch.topk(dist, args.k, largest=False).indices
            got = nb[r].to(torch.int64)
            if set(best.tolist()) != set(got.tolist()):
                bad += 1
        print("brute-force check: %d of %d rows differ" % (bad, args.check), flush=True)


if __name__ == "__main__":
    main()
