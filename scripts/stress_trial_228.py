"""The trial of scripts/stress_carry_api.py (seed 2) that did not return: 90000x16 fp16 rows (scratch/trial228.npy, written
by replaying the script's draws), K = 130, three virtual shards, tolerance 0, seed 370.
KMCUDA_AMD_CARRY=0|1 [YY=0] [SHARDS=1] python scripts/stress_trial_228.py"""
import os, sys
import numpy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from kmcuda_amd import kmeans_cuda
x = numpy.load(os.path.join(os.path.dirname(__file__), "..", "scratch", "trial228.npy"))
os.environ["KMCUDA_AMD_VIRTUAL_SHARDS"] = os.environ.get("SHARDS", "3")
print("start", x.shape, x.dtype, flush=True)
kmeans_cuda(x, 130, init="random", seed=370, tolerance=0.0, yinyang_t=float(os.environ.get("YY", "0.1")), metric="L2", device=1, verbosity=1)
print("returned", flush=True)
