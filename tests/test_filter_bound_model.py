"""Model check of the coarse filter's error bound (kmcuda_amd/csrc/lloyd_f16.hip: e_c in
lloyd_coarse2_kernel / lloyd_refine_kernel; yinyang_hint.hip: e_mfma of the f16 candidate sweep).

The matrix cores see hi(x') . hi(c') + bias, halves rounded to nearest, products exact, sums in fp32 in
an order the hardware does not document.  Every decision the filter takes rests on
    | acc16 - (x'.c' + bias) |  <=  E
with E built from norms: 2 eps (||x'|| C + B) for the fp32 accumulation, Cauchy-Schwarz on the MEASURED
residuals ||x' - hi(x')||, ||c' - hi(c')|| for the dropped parts, an absolute term for halves below the
normal range, 2e-6 (...) for the register number packed into the low 4 mantissa bits.  Here the left side
is computed in float64 for random and for adversarially aligned operands, with the fp32 sums taken in
several orders, and compared with E exactly as the kernels form it."""
import numpy

F = numpy.float32


def _bound(x, c, bias, D, packed):
    eps = F(1.02 * (D + 12.0) * 2.0 ** -24)                       # engine.cpp: eps_
    hx = x.astype(numpy.float16).astype(numpy.float32)
    hc = c.astype(numpy.float16).astype(numpy.float32)
    xn = F(numpy.sqrt(numpy.sum(x.astype(numpy.float64) ** 2))) * F(1.0001)
    cm = F(numpy.sqrt(numpy.sum(c.astype(numpy.float64) ** 2))) * F(1.000001)
    dx = F(numpy.sqrt(numpy.sum((x - hx).astype(numpy.float64) ** 2))) * F(1.0001)
    dc = F(numpy.sqrt(numpy.sum((c - hc).astype(numpy.float64) ** 2))) * F(1.0001)
    b = F(abs(bias))
    e = F(2.0) * eps * (xn * cm + b) + (xn * dc + dx * cm + dx * dc) * F(1.001) + F(6e-8) * F(numpy.sqrt(D)) * (xn + cm)
    if packed:
        e = e + F(2e-6) * (F(1.001) * xn * cm + b)
    return hx, hc, float(e)


def _sums(prod, bias):
    """fp32 accumulations of the (exact) products in different orders, all starting from the bias."""
    outs = []
    acc = F(bias)
    for p in prod:                                   # sequential
        acc = F(acc + p)
    outs.append(acc)
    acc = F(bias)
    for k in range(0, len(prod), 16):                # blocks of 16 summed pairwise first (one MFMA k-step)
        blk = list(prod[k:k + 16])
        while len(blk) > 1:
            blk = [F(blk[i] + blk[i + 1]) if i + 1 < len(blk) else blk[i] for i in range(0, len(blk), 2)]
        acc = F(acc + blk[0])
    outs.append(acc)
    acc = F(bias)
    for p in prod[::-1]:                             # reversed
        acc = F(acc + p)
    outs.append(acc)
    return outs


def _check(x, c, bias, D):
    exact = float(numpy.sum(x.astype(numpy.float64) * c.astype(numpy.float64)) + float(bias))
    for packed in (False, True):
        hx, hc, e = _bound(x, c, bias, D, packed)
        prod = (hx * hc).astype(numpy.float32)       # exact: 11 x 11 significant bits
        assert numpy.array_equal(prod.astype(numpy.float64), hx.astype(numpy.float64) * hc.astype(numpy.float64))
        for acc in _sums(prod, bias):
            if packed:
                bits = numpy.array([acc], numpy.float32).view(numpy.uint32)
                for r in (0, 15):
                    v = ((bits & numpy.uint32(0xFFFFFFF0)) | numpy.uint32(r)).view(numpy.float32)[0]
                    assert abs(float(v) - exact) <= e, (D, packed, float(v), exact, e)
            else:
                assert abs(float(acc) - exact) <= e, (D, packed, float(acc), exact, e)


def test_random_operands():
    rs = numpy.random.RandomState(1)
    for trial in range(400):
        D = int(rs.choice([16, 32, 64, 128, 256]))
        scale_x = 10.0 ** rs.uniform(-3, 3)
        scale_c = 10.0 ** rs.uniform(-3, 3)
        x = (rs.randn(D) * scale_x).astype(numpy.float32)
        c = (rs.randn(D) * scale_c).astype(numpy.float32)
        bias = F(-0.5 * numpy.sum(c.astype(numpy.float64) ** 2)) if trial % 2 else F(rs.randn() * scale_x * scale_c)
        _check(x, c, bias, D)


def test_aligned_rounding_errors():
    """every dropped part pushes the same way: x_i > 0 with residual +0.49 ulp(half), c likewise"""
    rs = numpy.random.RandomState(2)
    for trial in range(200):
        D = int(rs.choice([16, 64, 256]))
        base = (rs.rand(D) + 1.0) * 10.0 ** rs.uniform(-2, 2)
        h = base.astype(numpy.float16).astype(numpy.float64)
        ulp = numpy.spacing(h.astype(numpy.float16)).astype(numpy.float64)
        x = (h + 0.49 * ulp).astype(numpy.float32)
        hc = ((rs.rand(D) + 1.0) * 10.0 ** rs.uniform(-2, 2)).astype(numpy.float16).astype(numpy.float64)
        c = (hc + 0.49 * numpy.spacing(hc.astype(numpy.float16)).astype(numpy.float64)).astype(numpy.float32)
        if trial % 2:
            c = -c
        _check(x, c, F(-0.5 * numpy.sum(c.astype(numpy.float64) ** 2)), D)


def test_below_the_normal_range_of_halves():
    """values whose halves are subnormal or flush to zero: the absolute term has to carry them"""
    rs = numpy.random.RandomState(3)
    for trial in range(200):
        D = int(rs.choice([16, 64, 256]))
        x = (rs.randn(D) * 10.0 ** rs.uniform(-8, -4)).astype(numpy.float32)
        c = (rs.randn(D) * 10.0 ** rs.uniform(-8, 0)).astype(numpy.float32)
        if trial % 3 == 0:
            x[: D // 2] = (rs.randn(D // 2) * 100).astype(numpy.float32)     # mixed scales in one row
        _check(x, c, F(-0.5 * numpy.sum(c.astype(numpy.float64) ** 2)), D)
