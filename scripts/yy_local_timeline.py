import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["config_b.py"] + sys.argv[1:]
src = open(os.path.join(ROOT, "scripts", "config_b.py")).read().replace('if __name__ == "__main__":\n    main()', '')
src = src.replace("sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))", "")
exec(src)
main()
from kmcuda_amd import _lib
L = _lib.lib()
buf = (ctypes.c_ulonglong * 64)()
assert L.kmamd_stamps(buf) == 0
t = [int(buf[i]) for i in range(64)]
t0 = t[0]
names = {0: "kernel entry", 1: "row index known", 2: "operands + norms in registers", 3: "prologue done (bounds fold, cut-off)", 4: "sweep done",
         5: "last flush starts", 6: "replay operands requested", 40: "chains done", 41: "replay done", 42: "flush returned"}
print("\nyy_local_hint_kernel, wave 0 of block 20000, last launch: cycles since kernel entry (nq of its last flush = %d)" % t[43])
prev = t0
for k in list(range(0, 7)) + list(range(8, 40)) + [40, 41, 42]:
    if t[k] == 0: continue
    print("  %-40s %8d  (+%d)" % (names.get(k, "chain batch %d %s" % ((k - 8) // 2, "landed" if (k - 8) % 2 == 0 else "computed")), t[k] - t0, t[k] - prev))
    prev = t[k]
