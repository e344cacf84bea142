import sys, numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 3).astype(np.float64)
s, e, tiles = t[:, 0], t[:, 1], t[:, 2]
t0 = s.min()
s = (s - t0) / 100e6 * 1e3; e = (e - t0) / 100e6 * 1e3   # wall_clock64: 100 MHz -> ms
dur = e - s
print("blocks %d; kernel span %.1f ms; block duration: mean %.1f ms, min %.1f, median %.1f, 90%% %.1f, max %.1f" %
      (len(s), e.max(), dur.mean(), dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
print("tiles per block: mean %.0f, min %.0f, max %.0f; ms per 1000 tiles: median %.3f" % (tiles.mean(), tiles.min(), tiles.max(), np.median(dur / tiles * 1e3)))
slots = 512
busy = dur.sum() / slots
print("sum of block durations / %d slots = %.1f ms => %.1f %% of the span is tail / imbalance" % (slots, busy, 100 * (1 - busy / e.max())))
# when does the last block START, and how many slots are busy over time
for q in (0.8, 0.9, 0.95, 0.99):
    tq = e.max() * q
    print("  at %.0f %% of the span: %d blocks running" % (100 * q, int(((s <= tq) & (e > tq)).sum())))
order = np.argsort(s)
print("corr(duration, tiles) = %.3f; corr(duration, start order) = %.3f" % (np.corrcoef(dur, tiles)[0, 1], np.corrcoef(dur[order], np.arange(len(order)))[0, 1]))
