#!/usr/bin/env python
"""Stage-1 kernel A/B on a FIXED steady state: `--save f` runs 8 Lloyd iterations with the current
library and saves centroids + assignments; `--load f` (with KMCUDA_AMD_LIB pointing at a variant)
times lloyd_assign passes on exactly that state (HIP events around the coarse kernel).  Variants whose
results are wrong on purpose (KMX_ABL) can be timed this way without wrecking the trajectory."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy
import torch
from kmcuda_amd.distributed import HipBackend, ShardedLloyd

mode, path = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8000000
d, k = 256, 1024
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
x = torch.empty((n, d), dtype=torch.float32, device=dev)
for s in range(0, n, 1 << 20):
    x[s:s + (1 << 20)].uniform_(0, 1, generator=gen)
b = HipBackend(x, k, "L2", device_index=0, row_cache=True)
if mode == "--save":
    loop = ShardedLloyd(b, n)
    loop.set_centroids(x[torch.randperm(n, generator=gen, device=dev)[:k]].clone())
    for _ in range(8):
        loop.step()
    torch.cuda.synchronize()
    numpy.savez(path, c=b.centroids.cpu().numpy(), a=b.assignments.cpu().numpy())
    sys.exit(0)
st = numpy.load(path)
b.centroids.copy_(torch.from_numpy(st["c"]).to(dev))
asg = torch.from_numpy(st["a"]).to(dev)
for _ in range(2):
    b.assignments.copy_(asg)
    b.assign()
b.engine.profile(True)
for _ in range(6):
    b.assignments.copy_(asg)
    b.assign()
torch.cuda.synchronize()
p = b.engine.profile_read()
print("%-12s coarse %.3f ms   filter stage %.3f ms" % (os.path.basename(os.environ.get("KMCUDA_AMD_LIB", "default")),
                                                     p["coarse_ms"] / p["filter_launches"], p["filter_ms"] / p["filter_launches"]))
