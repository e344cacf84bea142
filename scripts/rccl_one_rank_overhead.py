#!/usr/bin/env python
"""What the collective's plumbing costs per iteration on a 1M-row shard, as far as ONE GPU can tell: the Python
loop (bench.py's N > 1 path) with a one-rank RCCL group reducing its fp64 buffer every iteration against the same
loop without a group.  (The wire time of a real 8-rank all-reduce is not in here.)
   python scripts/rccl_one_rank_overhead.py [--samples 1000000] [--steps 200]"""
import argparse
import datetime
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1000000)
    ap.add_argument("--features", type=int, default=256)
    ap.add_argument("--clusters", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    x = torch.rand((args.samples, args.features), device=dev, generator=gen)
    init = x[torch.randperm(args.samples, device=dev, generator=gen)[:args.clusters]].clone()

    def run(reduce_always):
        loop = ShardedLloyd(HipBackend(x, args.clusters, "L2", device_index=0), args.samples, reduce_always=reduce_always)
        loop.set_centroids(init.clone())
        for _ in range(args.warmup):
            loop.step(0.0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.step(0.0)
        loop.drain()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / args.steps * 1e3

    plain = run(False)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
    with_group = run(True)
    plain2 = run(False)
    dist.destroy_process_group()
    print("ms per iteration, %d x %d rows, K = %d: no group %.4f / %.4f; one-rank RCCL all-reduce of the %d-byte buffer "
          "every iteration %.4f  => +%.1f us" % (args.samples, args.features, args.clusters, plain, plain2,
                                                8 * (args.clusters * args.features + args.clusters + 4), with_group,
                                                (with_group - 0.5 * (plain + plain2)) * 1e3))


if __name__ == "__main__":
    main()
