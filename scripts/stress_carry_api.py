"""Randomised whole kmeans_cuda() calls (GPU): the default schedule (bounds carried after the hand-over point) against
KMCUDA_AMD_CARRY=0, same seeds: centroids and assignments must be bit-identical -- shapes, metrics, row types, shard
counts, tolerances drawn at random.   python scripts/stress_carry_api.py [seconds] [seed]"""
import os, sys, time, traceback
import numpy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))
from stress_carry import make
from kmcuda_amd import kmeans_cuda


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = numpy.random.RandomState(seed)
    t0 = time.time()
    trials = failures = 0
    while time.time() - t0 < budget:
        n = int(rs.choice([3000, 20000, 90000, 300000]))
        d = int(rs.choice([16, 32, 64, 100, 128, 256, 300, 512]))
        k = int(rs.choice([20, 64, 130, 300]))
        k = min(k, n // 20)
        metric = str(rs.choice(["L2", "cos"]))
        half = bool(rs.rand() < 0.25)
        shards = int(rs.choice([1, 1, 2, 3]))
        tol = float(rs.choice([0.01, 0.001, 0.0001, 0.00002]))   # (never 0: fp16 runs end in limit cycles of one reassignment -- in the reference too -- and never return)
        init = str(rs.choice(["random", "kmeans++"])) if n <= 90000 else "random"
        kind, x = make(rs, n, d, k, metric)
        if kind == "nan":
            x = numpy.nan_to_num(x, nan=0.5)   # (the API rejects nothing, but k-means++ over NaN rows is its own subject)
            if metric == "cos":
                x /= numpy.maximum(numpy.linalg.norm(x, axis=1, keepdims=True), 1e-12)
        if half:
            x = x.astype(numpy.float16)
        sd = int(rs.randint(1, 1000))
        desc = "%dx%d@%d %s %s %s shards=%d tol=%g init=%s" % (n, d, k, metric, "fp16" if half else "fp32", kind, shards, tol, init)
        if shards > 1:
            os.environ["KMCUDA_AMD_VIRTUAL_SHARDS"] = str(shards)
        else:
            os.environ.pop("KMCUDA_AMD_VIRTUAL_SHARDS", None)
        try:
            res = []
            for carry in ("1", "0"):
                os.environ["KMCUDA_AMD_CARRY"] = carry
                res.append(kmeans_cuda(x, k, init=init, seed=sd, tolerance=tol, yinyang_t=0.1, metric=metric, device=1,
                                       verbosity=0))
            same_a = bool((res[0][1] == res[1][1]).all())
            same_c = bool((res[0][0].view(numpy.uint16 if half else numpy.uint32) ==
                           res[1][0].view(numpy.uint16 if half else numpy.uint32)).all())
            if same_a and same_c:
                print("ok   %s" % desc, flush=True)
            else:
                failures += 1
                print("FAIL %s: assignments %s, centroids %s" % (desc, same_a, same_c), flush=True)
        except Exception:
            failures += 1
            print("ERR  %s" % desc, flush=True)
            traceback.print_exc()
        trials += 1
    os.environ.pop("KMCUDA_AMD_CARRY", None)
    print("%d trials, %d failures, %.0f s" % (trials, failures, time.time() - t0), flush=True)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
