"""One trial of scripts/stress_carry_api.py again: replays the script's random draws for `seed` up to the trial whose
description contains `needle`, then runs that call with and without the carried bounds under a few environments.
    python scripts/stress_replay_api.py <seed> "<needle>" """
import os, sys
import numpy
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))
from stress_carry import make


def main():
    seed, needle = int(sys.argv[1]), sys.argv[2]
    rs = numpy.random.RandomState(seed)
    for trial in range(100000):
        n = int(rs.choice([3000, 20000, 90000, 300000]))
        d = int(rs.choice([16, 32, 64, 100, 128, 256, 300, 512]))
        k = int(rs.choice([20, 64, 130, 300]))
        k = min(k, n // 20)
        metric = str(rs.choice(["L2", "cos"]))
        half = bool(rs.rand() < 0.25)
        shards = int(rs.choice([1, 1, 2, 3]))
        tol = float(rs.choice([0.01, 0.001, 0.0001, 0.00002]))
        init = str(rs.choice(["random", "kmeans++"])) if n <= 90000 else "random"
        kind, x = make(rs, n, d, k, metric)
        if kind == "nan":
            x = numpy.nan_to_num(x, nan=0.5)
            if metric == "cos":
                x /= numpy.maximum(numpy.linalg.norm(x, axis=1, keepdims=True), 1e-12)
        if half:
            x = x.astype(numpy.float16)
        sd = int(rs.randint(1, 1000))
        desc = "%dx%d@%d %s %s %s shards=%d tol=%g init=%s" % (n, d, k, metric, "fp16" if half else "fp32", kind, shards, tol, init)
        if needle in desc:
            break
    else:
        raise SystemExit("no such trial")
    print("trial %d: %s (seed of the call %d)" % (trial, desc, sd), flush=True)
    if os.environ.get("REPLAY_HALF") == "0":     # variations of the trial: which ingredient matters
        x, half = x.astype(numpy.float32), False
    if os.environ.get("REPLAY_METRIC"):
        metric = os.environ["REPLAY_METRIC"]
    print("running as: %s, %s" % (metric, "fp16" if half else "fp32"), flush=True)
    import ctypes
    libs = {}

    def call(env, path=None):
        """the C ABI directly (an older build of the library may lack newer kmamd_* symbols the binding wants)"""
        for key in ("KMCUDA_AMD_CARRY", "KMCUDA_AMD_CARRY_PAIRS", "KMCUDA_AMD_CARRY_MAX", "KMCUDA_AMD_VIRTUAL_SHARDS",
                    "KMCUDA_AMD_CARRY_TRACE"):
            os.environ.pop(key, None)
        if shards > 1:
            os.environ["KMCUDA_AMD_VIRTUAL_SHARDS"] = str(shards)
        os.environ.update(env)
        path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kmcuda_amd", "libKMCUDA.so")
        if path not in libs:
            libs[path] = ctypes.CDLL(path)
        L = libs[path]
        cen = numpy.zeros((k, d), numpy.float16 if half else numpy.float32)
        asg = numpy.zeros(n, numpy.uint32)
        xs = numpy.ascontiguousarray(x)
        L.kmeans_cuda.restype = ctypes.c_int
        rc = L.kmeans_cuda(ctypes.c_int(0 if init == "random" else 1), None, ctypes.c_float(tol), ctypes.c_float(0.1),
                           ctypes.c_int(0 if metric == "L2" else 1), ctypes.c_uint32(n),
                           ctypes.c_uint16(d // 2 if half else d), ctypes.c_uint32(k), ctypes.c_uint32(sd), ctypes.c_uint32(1),
                           ctypes.c_int32(-1), ctypes.c_int32(1 if half else 0), ctypes.c_int32(0),
                           xs.ctypes.data_as(ctypes.c_void_p), cen.ctypes.data_as(ctypes.c_void_p),
                           asg.ctypes.data_as(ctypes.c_void_p), None)
        assert rc == 0, rc
        return cen, asg

    view = numpy.uint16 if half else numpy.uint32
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    for name in sys.argv[3:] or [None]:
        path = os.path.join(root, "scratch", name) if name else None
        ref = call({"KMCUDA_AMD_CARRY": "0"}, path)
        for env in ({"KMCUDA_AMD_CARRY": "0"}, {"KMCUDA_AMD_CARRY": "1"}, {"KMCUDA_AMD_CARRY": "1", "KMCUDA_AMD_CARRY_PAIRS": "0"}, {}):
            got = call(env, path)
            print(name or "built", env, "assignments differ:", int((ref[1] != got[1]).sum()), "centroid words differ:",
                  int((ref[0].view(view) != got[0].view(view)).sum()), flush=True)


if __name__ == "__main__":
    main()
