"""The reference's angular distance is `p >= 1 ? 0 : p <= -1 ? pi : acos(p)` (metric_abstraction.h:171-177; oracle:
kmcuda_oracle.c, exact.hpp:112 on the device): every centroid whose product with a row reaches 1 sits at distance 0, and
the ascending strict-`<` scan of kmeans_assign_lloyd (kmeans.cu:293-364) keeps the LOWEST index among them -- not the
largest product.  Round 5 found the device filters deciding such rows by the larger product (DESIGN_LOG
13.13): this file pins what the right answer is, on the CPU, for whoever makes the filters leave those rows to the exact
kernels."""
import numpy

import oracle


def test_products_at_or_beyond_one_tie_and_the_lowest_index_wins():
    # half-precision "unit" rows: norms off 1 by up to 1e-3, as a float16 cast of normalised rows leaves them
    x = numpy.zeros((3, 16), numpy.float32)
    x[0, 0] = 1.0009766          # ||x|| slightly above 1
    x[1, 1] = 1.0
    x[2, 2] = 0.99902344
    c = numpy.zeros((4, 16), numpy.float32)
    c[0, 0] = 0.9995117          # product with row 0: 1.00049 >= 1 -> distance 0
    c[1, 0] = 1.0009766          # product with row 0: 1.00195 >= 1 -> distance 0 too, LARGER product, higher index
    c[2, 1] = 1.0                # product with row 1: exactly 1
    c[3, 2] = 1.0
    asg, prev, changed = oracle.lloyd_assign(x, c, metric=oracle.COS)
    assert asg[0] == 0           # not 1: both are at distance 0, the scan keeps the first
    assert asg[1] == 2 and asg[2] == 3
    # the same two centroids in the other order: now the larger product IS the first
    c2 = c.copy()
    c2[[0, 1]] = c2[[1, 0]]
    asg2, _, _ = oracle.lloyd_assign(x, c2, metric=oracle.COS)
    assert asg2[0] == 0


def test_a_single_product_beyond_one_wins_over_every_smaller_one():
    rs = numpy.random.RandomState(3)
    x = rs.randn(50, 24).astype(numpy.float32)
    x /= numpy.linalg.norm(x, axis=1, keepdims=True)
    c = rs.randn(9, 24).astype(numpy.float32)
    c /= numpy.linalg.norm(c, axis=1, keepdims=True)
    c[7] = x[11] * 1.001         # one centroid past the clamp for row 11 only
    asg, _, _ = oracle.lloyd_assign(x, c, metric=oracle.COS)
    assert asg[11] == 7
    prods = x @ c.T
    others = numpy.delete(numpy.arange(50), 11)
    assert (asg[others] == prods[others].argmax(axis=1)).all()   # below the clamp the largest product is the nearest
