"""scripts/scale_parity.py N [--one-gpu] -- kmeans_cuda() and knn_cuda() over a device mask of N GPUs against device 1.

What must hold when the rows are sharded over N devices (reference: the multi-device loops of src/kmeans.cu:1014-1024 /
:1251-1261 and knn.cu):
  * Lloyd (yinyang_t = 0), the default fp64 update: the same number of iterations, assignments identical, centroids
    within rtol 2e-5 (the fp64 partial sums of N shards add up in another order than one shard's; the strict update --
    the reference's global serial order, KMCUDA_AMD_EXACT_UPDATE=1 -- is refused over more than one shard by design);
  * k-means++ seeds identical (the sharded chooser), knn_cuda neighbour lists identical (indices and order).
--one-gpu: the N shards live on GPU 0 (KMCUDA_AMD_VIRTUAL_SHARDS=N): every line of the sharded host code without N GPUs.
Exit code 0 = all of it held; anything else prints what differed."""
import os
import sys

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_dev = int(sys.argv[1])
    one_gpu = "--one-gpu" in sys.argv
    import torch
    from kmcuda_amd import kmeans_cuda, knn_cuda
    if one_gpu:
        mask = 1
    else:
        if torch.cuda.device_count() < n_dev:
            raise SystemExit("scale_parity: %d GPUs asked, %d visible" % (n_dev, torch.cuda.device_count()))
        mask = (1 << n_dev) - 1

    def sharded(fn):
        if one_gpu:
            os.environ["KMCUDA_AMD_VIRTUAL_SHARDS"] = str(n_dev)
        try:
            return fn(mask)
        finally:
            os.environ.pop("KMCUDA_AMD_VIRTUAL_SHARDS", None)

    rs = numpy.random.RandomState(5)
    n, d, k = 400000, 64, 256
    centres = rs.rand(k, d).astype(numpy.float32)
    x = (centres[rs.randint(0, k, n)] + 0.08 * rs.randn(n, d)).astype(numpy.float32)
    bad = []

    def run(dev, init):
        return kmeans_cuda(x, k, init=init, seed=9, tolerance=0.0005, yinyang_t=0, device=dev, verbosity=0)

    for init in ("random", "k-means++"):
        c1, a1 = run(1, init)
        cN, aN = sharded(lambda m: run(m, init))
        if not (a1 == aN).all():
            bad.append("default update, init=%s: %d assignments differ" % (init, int((a1 != aN).sum())))
        if not numpy.allclose(c1, cN, rtol=2e-5, atol=1e-7, equal_nan=True):
            bad.append("default update, init=%s: centroids beyond rtol 2e-5" % init)
    nb1 = knn_cuda(10, x, c1, a1, metric="L2", device=1, verbosity=0)
    nbN = sharded(lambda m: knn_cuda(10, x, c1, a1, metric="L2", device=m, verbosity=0))
    if not (nb1 == nbN).all():
        bad.append("knn_cuda: %d neighbour entries differ" % int((nb1 != nbN).sum()))
    if bad:
        print("scale_parity FAILED over %d %s:" % (n_dev, "virtual shards" if one_gpu else "GPUs"))
        for b in bad:
            print("  " + b)
        sys.exit(1)
    print("scale_parity ok: %d %s == one device (Lloyd from random and k-means++ seeds, k-NN lists)" % (
        n_dev, "virtual shards on GPU 0" if one_gpu else "GPUs"))


if __name__ == "__main__":
    main()
