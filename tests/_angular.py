"""What may differ between the device and the oracle under the angular metric, and what may not.

`acosf` is libm in the oracle, ocml on the GPU and CUDA's in the reference (parity-unpinned), so a row whose
two nearest centroids are within a last place of each other in the oracle's OWN arithmetic may land on either.  Anything
else is a defect -- in particular a row whose candidates both sit at distance 0: the clamp `p >= 1 ? 0 : acos(p)`
(metric_abstraction.h:171-177) involves no acos, the lowest index wins (tests/test_gpu_angular_clamp.py)."""
import numpy

import oracle


def assert_only_acos_matters(x, c, got, ref, what="", max_fraction=2e-3):
    """got / ref: assignments of the rows x against the centroids c (device / oracle).  Returns the number of rows that
    differ -- every one of them an acos last-place matter, never a clamp tie, and no more than max_fraction of the rows."""
    got, ref = numpy.asarray(got), numpy.asarray(ref)
    bad = numpy.nonzero(got != ref)[0]
    assert bad.size <= max_fraction * len(ref), "%s: %d of %d rows differ" % (what, bad.size, len(ref))
    k = c.shape[0]
    for i in bad:
        assert got[i] < k and ref[i] < k, (what, int(i), int(got[i]), int(ref[i]))
        dg = oracle.distance(x[i], c[got[i]], metric=oracle.COS)
        dr = oracle.distance(x[i], c[ref[i]], metric=oracle.COS)
        assert not (dg == 0.0 and dr == 0.0), "%s: row %d: a tie at the clamp went to %d, not %d" % (what, i, got[i], ref[i])
        assert abs(dg - dr) <= 2 * numpy.spacing(numpy.float32(max(dg, dr))), \
            "%s: row %d: distances %r (device's %d) and %r (oracle's %d) are not a last-place matter" % (
                what, i, dg, got[i], dr, ref[i])
    return int(bad.size)
