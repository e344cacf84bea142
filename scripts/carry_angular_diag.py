"""Diagnosis: where do carried angular passes leave the plain schedule?  (GPU)"""
import numpy, torch, sys
sys.path.insert(0, ".")
from kmcuda_amd.distributed import HipBackend, ShardedLloyd

rs = numpy.random.RandomState(3)
cen = rs.randn(40, 64)
x = cen[rs.randint(0, 40, 60000)] + 0.15 * rs.randn(60000, 64)
x = (x / numpy.linalg.norm(x, axis=1, keepdims=True)).astype(numpy.float32)
k = 40
dev = torch.device("cuda", 0)
rs = numpy.random.RandomState(5)
init = x[rs.choice(len(x), k, replace=False)].copy()
xs = torch.from_numpy(x).to(dev)
loops = []
for which in range(2):
    b = HipBackend(xs, k, "cos", device_index=0)
    loop = ShardedLloyd(b, len(x))
    loop.set_centroids(torch.from_numpy(init).to(dev))
    loops.append(loop)
plain, carry = loops
cs = []
x64 = x.astype(numpy.float64)
for it in range(8):
    if it == 2:
        carry.b.engine.set_carry(True)
    cs.append(plain.b.centroids.cpu().numpy().astype(numpy.float64))
    cc = carry.b.centroids.cpu().numpy().astype(numpy.float64)
    print("it", it, "centroids equal before the pass:", (cs[-1] == cc).all(), "norms", numpy.linalg.norm(cs[-1], axis=1)[:3])
    for loop in loops:
        loop.step(tolerance=0.0)
    for loop in loops:
        loop.b.synchronize()
    a0, a1 = plain.b.assignments.cpu().numpy(), carry.b.assignments.cpu().numpy()
    bad = numpy.nonzero(a0 != a1)[0]
    print("it", it, "differ", len(bad), "carry stats", carry.b.engine.carry_stats())
    if len(bad):
        cold, cnew = cs[-2], cs[-1]
        dr = numpy.linalg.norm(cnew - cold, axis=1)
        print("max drift", dr.max())
        for s in bad[:10]:
            so, sn = cold @ x64[s], cnew @ x64[s]
            a = a1[s]
            oth = numpy.delete(so, a).max()
            print("row", s, "plain", a0[s], "carry", a1[s], "old best", int(so.argmax()), "old gap of carry's", so[a] - oth,
                  "shrink", numpy.linalg.norm(x64[s]) * (dr[a] + dr.max()), "new gap", sn[a] - numpy.delete(sn, a).max(),
                  "new best", int(sn.argmax()))
        break
