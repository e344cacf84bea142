"""GPU-vs-GPU stress: the default filter chain (cached / uncached) against the exact-only kernels
(every pair in reference arithmetic) on states the CPU oracle is too slow for."""
import sys, time, numpy, torch
sys.path.insert(0, ".")
from kmcuda_amd.engine import Engine
dev = torch.device("cuda", 0)
shapes = [(300000, 256, 1024), (200000, 256, 4000), (500000, 128, 512), (400000, 64, 2000), (250000, 200, 777),
          (100000, 32, 100), (150001, 255, 1023), (120000, 16, 5000), (50000, 256, 20000)]
total_bad = 0
import os
METRIC = os.environ.get("STRESS_METRIC", "L2")
for (n, d, k) in shapes:
    for data in ("uniform", "gauss"):
        g = torch.Generator(device=dev); g.manual_seed(n + d + k)
        if data == "uniform":
            x = torch.rand((n, d), device=dev, generator=g)
        else:
            cen = torch.rand((64, d), device=dev, generator=g) * 8
            x = torch.randn((n, d), device=dev, generator=g) + cen[torch.randint(0, 64, (n,), device=dev, generator=g)]
        if METRIC != "L2":   # angular: unit rows (kmcuda.cc:232-252 rejects anything else)
            x = x - x.mean(0, keepdim=True) if data == "gauss" else x
            x = (x / x.norm(dim=1, keepdim=True)).contiguous()
        c = x[torch.randperm(n, device=dev, generator=g)[:k]].clone()
        engs = {}
        for name in ("cached", "uncached", "exact"):
            e = Engine(n, d, k, METRIC, device=0)
            if name == "cached":
                e.set_row_cache(True)
            engs[name] = (e, torch.full((n,), -1, dtype=torch.int32, device=dev), torch.full((n,), -1, dtype=torch.int32, device=dev))
        for it in range(4):
            res = {}
            for name, (e, asg, prev) in engs.items():
                e.reset_counters(-1)
                e.lloyd_assign(x, c, asg, prev, exact=(name == "exact"))
                cnt = e.counters()
                res[name] = (asg.clone(), prev.clone(), cnt[0])
            for name in ("cached", "uncached"):
                bad = int((res[name][0] != res["exact"][0]).sum()) + int((res[name][1] != res["exact"][1]).sum())
                if bad or res[name][2] != res["exact"][2]:
                    print("MISMATCH", (n, d, k), data, name, "iter", it, bad, res[name][2], res["exact"][2], flush=True)
                    total_bad += 1
            # plain mean update (any centroids will do for the comparison)
            a = res["exact"][0].long()
            sums = torch.zeros((k, d), device=dev).index_add_(0, a.clamp(max=k - 1), x)
            cnts = torch.bincount(a.clamp(max=k - 1), minlength=k).clamp(min=1).unsqueeze(1)
            c = (sums / cnts).contiguous()
            if METRIC != "L2":
                c = (c / c.norm(dim=1, keepdim=True).clamp(min=1e-20)).contiguous()
        for e, _, _ in engs.values():
            e.close()
        print("ok", (n, d, k), data, flush=True)
print("TOTAL MISMATCHING CASES", total_bad)
