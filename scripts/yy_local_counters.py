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
buf = (ctypes.c_ulonglong * 8)()
L.kmamd_yyl_debug(buf)
wc, pro, nf, fc, ns, sc, wt, epi = [buf[i] for i in range(8)]
print("\nyy_local_hint_kernel, all launches: wave cycles %.3g; prologue %.1f %%, waiting (vmcnt + barrier) %.1f %%, slow path %.1f %% (%.1f entries per wave-launch... %d total, %.0f cycles each), flushes %.1f %% (%d, %.0f cycles each), epilogue incl. final flush %.1f %%" % (wc, 100.0 * pro / wc, 100.0 * wt / wc, 100.0 * sc / wc, 0.0, ns, sc / max(ns, 1), 100.0 * fc / wc, nf, fc / max(nf, 1), 100.0 * epi / wc))
