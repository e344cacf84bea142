"""AFK-MC2's random numbers (reference: kmeans.cu:107-116, curand_init(seed, thread, step) +
curand_uniform; host chain kmcuda.cc:337-396).

cuRAND is not in /root/reference and not on this box, so the generator is a restatement (oracle/kmcuda_oracle.c has the
piece-by-piece account: cuRAND and rocRAND share the generator and the jumps and differ in the seed scrambling; the
reference's four iteration pins are met by rocRAND's constants -- the default -- and three of them by cuRAND's as
quoted there; `KMCUDA_AMD_AFKMC2_SEEDING=curand` / `oracle.set_afkmc2_seeding(True)` select those).  What CAN be pinned,
and is here:
  1. the restatement's generator, its 2^67-draw subsequence jumps, its offset jumps and its Weyl bookkeeping equal a
     second, independent implementation -- rocRAND's HOST generator (librocrand through ctypes, ROCRAND_RNG_PSEUDO_XORWOW);
  2. the DEVICE generator of the seeding kernels (csrc/seeding.hip: rocRAND's engine, the state seeded either way) equals
     the restatement, stream by stream, at the (seed, thread, step) triples the reference uses, under BOTH seedings;
  3. the two seedings do give different streams (so the distinction is not academic);
  4. whole AFK-MC2 calls equal the oracle's seed for seed under both.
What cannot: which stream CUDA's library produced for the reference's authors (parity unpinned)."""
import ctypes

import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

ROCRAND_RNG_PSEUDO_XORWOW = 401


def _rocrand_host(seed, offset, n):
    """n outputs of rocRAND's host XORWOW generator (legacy ordering: its documented thread layout)."""
    L = ctypes.CDLL("/opt/rocm/lib/librocrand.so")
    gen = ctypes.c_void_p()
    assert L.rocrand_create_generator_host(ctypes.byref(gen), ROCRAND_RNG_PSEUDO_XORWOW) == 0
    try:
        assert L.rocrand_set_seed(gen, ctypes.c_ulonglong(seed)) == 0
        assert L.rocrand_set_offset(gen, ctypes.c_ulonglong(offset)) == 0
        out = numpy.empty(n, numpy.uint32)
        assert L.rocrand_generate(gen, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n)) == 0
        return out
    finally:
        L.rocrand_destroy_generator(gen)


def _layout(out, seed):
    """How the host generator lays streams out at offset 0: out[i] = draw (i // T) of subsequence (i % T) for its thread
    count T.  Found from the data (T is the library's launch configuration), then asserted over the whole buffer."""
    assert int(out[0]) == int(oracle.xorwow_draws(seed, 0, 0, 1, curand_seeding=False)[0])
    second = int(oracle.xorwow_draws(seed, 0, 0, 2, curand_seeding=False)[1])
    hits = numpy.nonzero(out == numpy.uint32(second))[0]
    assert hits.size >= 1, "subsequence 0's second draw is nowhere in the host generator's output"
    return int(hits[0])


@pytest.mark.parametrize("seed,offset", [(0, 0), (3, 0), (1234, 5), (0xDEADBEEFCAFE, 1000003), (2**63 + 11, 2**33 + 7)])
def test_restatement_equals_rocrand_host_generator_under_rocrand_seeding(seed, offset):
    """offset: per STREAM (the reference's `step`).  The host generator's own offset counts outputs of the interleaved
    sequence, so offset x T there moves every stream by `offset`."""
    n = 1 << 20
    T = _layout(_rocrand_host(seed, 0, n), seed)
    assert T >= 64 and n // T >= 2, T
    out = _rocrand_host(seed, offset * T, n)
    # every 97th stream (and the first / last), all of its draws that the buffer holds
    for t in sorted(set(list(range(0, T, 97)) + [1, T - 1])):
        mine = out[t::T]
        ref = oracle.xorwow_draws(seed, t, offset, len(mine), curand_seeding=False)
        assert (mine == ref).all(), (seed, offset, t, T)


@pytest.mark.parametrize("curand", [False, True])
@pytest.mark.parametrize("seed,step", [(3, 0), (3, 1), (3, 49), (777, 199), (0xFFFFFFFF, 12345), (2**40 + 5, 7)])
def test_device_draws_equal_the_restatement(seed, step, curand, monkeypatch):
    """The triples of the reference's call: seed = the API's 32-bit seed (widened), subsequence = thread < m (m up to
    N / 2), offset = the seeding step."""
    from kmcuda_amd import _lib
    from kmcuda_amd.engine import Engine
    if curand:
        monkeypatch.setenv("KMCUDA_AMD_AFKMC2_SEEDING", "curand")
    else:
        monkeypatch.delenv("KMCUDA_AMD_AFKMC2_SEEDING", raising=False)
    dev = torch.device("cuda", 0)
    threads, n = 4096, 6
    out = torch.zeros(threads * n, dtype=torch.int32, device=dev)
    eng = Engine(1024, 8, 4, "L2", device=0)
    _lib.check(eng.lib.kmamd_afkmc2_draws(eng.h, seed, step, threads, n, ctypes.c_void_p(out.data_ptr())), "kmamd_afkmc2_draws")
    eng.sync()
    got = out.cpu().numpy().view(numpy.uint32).reshape(threads, n)
    eng.close()
    for t in list(range(0, 64)) + list(range(64, threads, 61)) + [threads - 1]:
        ref = oracle.xorwow_draws(seed, t, step, n, curand_seeding=curand)
        assert (got[t] == ref).all(), (seed, step, t)
    other = oracle.xorwow_draws(seed, 0, step, n, curand_seeding=not curand)
    assert not (got[0] == other).all()   # the two libraries' seedings are different streams


@pytest.mark.parametrize("curand", [False, True])
def test_afkmc2_whole_call_equals_the_oracle_under_both_seedings(fixture13k, curand, monkeypatch):
    """kmeans_cuda(init=afkmc2) seed for seed against the oracle; the reference's pin of 4 iterations (test.py:248-262)
    holds under both (under the default seeding so do the other three pins: tests/test_gpu_kmeans.py, test_gpu_fp16.py)."""
    from kmcuda_amd import kmeans_cuda
    if curand:
        monkeypatch.setenv("KMCUDA_AMD_AFKMC2_SEEDING", "curand")
    else:
        monkeypatch.delenv("KMCUDA_AMD_AFKMC2_SEEDING", raising=False)
    oracle.set_afkmc2_seeding(curand)
    try:
        cen, asg = kmeans_cuda(fixture13k, 50, init=("afkmc2", 200), seed=3, tolerance=0.05, yinyang_t=0, device=1, verbosity=0)
        ocen, oasg, log = oracle.kmeans(fixture13k, 50, init=("afkmc2", 200), seed=3, tolerance=0.05, yinyang_t=0)
    finally:
        oracle.set_afkmc2_seeding(False)
    assert len(log) == 4
    assert (asg == oasg).mean() > 0.999
    numpy.testing.assert_allclose(cen, ocen, rtol=2e-4, atol=1e-5)
