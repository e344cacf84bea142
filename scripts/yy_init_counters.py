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
L.kmamd_yyi_debug(buf)
f, ent, fc, wc, mc, sc, scans, forced = [buf[i] for i in range(8)]
waves = 2 * (8000000 + 31) // 32
print("\nper wave: %.1f flushes (%.1f forced), %.1f entries per flush of 128, %.0f cycles per flush; scans %.2f" % (f / waves, forced / waves, ent / max(f, 1), fc / max(f, 1), scans / waves))
print("wave cycles %.3g: flushes %.1f %%, matrix-core loop %.1f %%, stage+barriers %.1f %%" % (wc / waves, 100.0 * fc / wc, 100.0 * mc / wc, 100.0 * sc / wc))
