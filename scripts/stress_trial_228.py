"""The trial of scripts/stress_carry_api.py (seed 2, as first committed: tolerances drawn from 0.01 / 0.001 / 0.0001 / 0) that
did not return: 90000x16 fp16 rows, K = 130, three virtual shards, tolerance 0, seed 370 -- a limit cycle of ONE
reassignment per iteration under every schedule (KMCUDA_AMD_CARRY=0|1, YY=0: plain Lloyd), as in the reference, whose
loop has no other way out than `changed <= tolerance * N` either (kmeans.cu:697-717).  The rows are the 229th draw of
that script's generator: replayed here (half a minute of numpy) unless scratch/trial228.npy holds them already.
KMCUDA_AMD_CARRY=0|1 [YY=0] [SHARDS=1] timeout 15 python scripts/stress_trial_228.py"""
import os, sys, types
import numpy
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, ".."))
sys.path.insert(0, here)
cache = os.path.join(here, "..", "scratch", "trial228.npy")
if os.path.exists(cache):
    x = numpy.load(cache)
else:
    sys.modules.setdefault("test_gpu_carry", types.ModuleType("test_gpu_carry"))   # (stress_carry imports it for _run_pair only)
    from stress_carry import make
    rs = numpy.random.RandomState(2)
    for t in range(229):
        n = int(rs.choice([3000, 20000, 90000, 300000]))
        d = int(rs.choice([16, 32, 64, 100, 128, 256, 300, 512]))
        k = min(int(rs.choice([20, 64, 130, 300])), n // 20)
        metric = str(rs.choice(["L2", "cos"]))
        half = bool(rs.rand() < 0.25)
        shards = int(rs.choice([1, 1, 2, 3]))
        tol = float(rs.choice([0.01, 0.001, 0.0001, 0.0]))
        init = str(rs.choice(["random", "kmeans++"])) if n <= 90000 else "random"
        kind, x = make(rs, n, d, k, metric)
        sd = int(rs.randint(1, 1000))
    assert (n, d, k, metric, half, shards, tol, init, sd) == (90000, 16, 130, "L2", True, 3, 0.0, "random", 370)
    x = x.astype(numpy.float16)
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    numpy.save(cache, x)
from kmcuda_amd import kmeans_cuda
os.environ["KMCUDA_AMD_VIRTUAL_SHARDS"] = os.environ.get("SHARDS", "3")
print("start", x.shape, x.dtype, flush=True)
kmeans_cuda(x, 130, init="random", seed=370, tolerance=0.0, yinyang_t=float(os.environ.get("YY", "0.1")), metric="L2", device=1, verbosity=1)
print("returned", flush=True)
