"""Real96 high-precision anticyclic FFT (high-precision-anticyclic-fft/src/code.cpp = HP).

PARITY UNPINNED by reference object code: HP/code.cpp needs NTL, which this image lacks, so the C
restatement in oracle/ cannot be compared with the reference binary.  It is held instead by
  * the reference's OWN assertions on the tables (HP:531-541 unit circle, HP:563-567 omega*ombar = 1),
  * two independent twiddle builders that must agree entry for entry (oracle: libquadmath; product:
    124-bit fixed-point Taylor series), plus mpmath when it is importable,
  * the defining property: the transform pair multiplies polynomials mod X^N + 1 (checked against
    exact integer products), and the round trip the reference prints (HP:583-586).
The GPU kernels are then bit-compared with the restatement (integer arithmetic: equality)."""
import importlib

import numpy as np
import pytest

import oracle_py as O

T = importlib.import_module("experimental-tfhe_amd")


def signed128(a):
    """[..., 2] uint64 (lo, hi) -> python ints (two's complement)"""
    a = np.asarray(a, np.uint64)
    lo, hi = a[..., 0].astype(object), a[..., 1].astype(object)
    v = lo + (hi << 64)
    return np.where(v >= 2 ** 127, v - 2 ** 128, v)


def to_u128(v):
    v = int(v) % 2 ** 128
    return [v & (2 ** 64 - 1), v >> 64]


@pytest.mark.parametrize("n", [64, 2048, 4096])
def test_twiddle_builders_agree(n):
    oa, ob = O.hp_twiddles(n)
    pa, pb = T.hp_twiddles(n)
    assert np.array_equal(oa, pa), "powomega: quadmath vs fixed-point Taylor"
    assert np.array_equal(ob, pb), "powombar"


def test_twiddles_match_mpmath():
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 300
    n = 4096
    pa, _ = T.hp_twiddles(n)
    c, s = signed128(pa[:, 0:2]), signed128(pa[:, 2:4])
    for i in list(range(0, 40)) + list(range(1000, 1050)) + [2047, 2048, 2049, 3071, 3072, 3073, 4095]:
        wc = int(mp.nint(mp.cos(2 * mp.pi * i / n) * 2 ** 64))
        ws = int(mp.nint(mp.sin(2 * mp.pi * i / n) * 2 ** 64))
        wc = 2 ** 64 - 1 if wc == 2 ** 64 else wc   # HP:248
        ws = 2 ** 64 - 1 if ws == 2 ** 64 else ws   # HP:265
        assert (int(c[i]), int(s[i])) == (wc, ws), i


def test_reference_table_assertions():
    """HP:531-541 (c*c + s*s very_close to 1) and HP:563-567 (powomega[i] * powombar[i] very_close to 1),
    with the reference's own tolerance |difference| < 10000 units of 2^-64 (HP:206-212)"""
    n = 4096
    pa, pb = T.hp_twiddles(n)
    c, s, sb = signed128(pa[:, 0:2]), signed128(pa[:, 2:4]), signed128(pb[:, 2:4])
    one = 2 ** 64
    for i in range(n):
        ci, si, bi = int(c[i]), int(s[i]), int(sb[i])
        assert abs((ci * ci >> 64) + (si * si >> 64) - one) < 10000, i
        re = (ci * ci >> 64) - (si * bi >> 64)      # (c + i s)(c + i sb), sb = sin(n - i) = -s
        im = (ci * bi >> 64) + (si * ci >> 64)
        assert abs(re - one) < 10000 and abs(im) < 10000, i


@pytest.mark.parametrize("N", [64, 2048])
def test_round_trip(N):
    """FFT(iFFT(in)) == in up to the low bits lost to the fixed-point products (HP:583-586 prints the xor)"""
    rs = np.random.RandomState(3)
    pa, pb = O.hp_twiddles(2 * N)
    x = rs.randint(-2 ** 63, 2 ** 63 - 1, size=N, dtype=np.int64)
    back = O.hp_fft(O.hp_ifft(x, pa), pb)
    err = np.abs((back.astype(object) - x.astype(object)))
    assert int(err.max()) <= 16, int(err.max())   # 6 observed
    z = np.zeros(N, np.int64)
    assert np.array_equal(O.hp_fft(O.hp_ifft(z, pa), pb), z)


def test_transform_pair_multiplies_polynomials():
    """iFFT both factors, multiply the spectra point by point with exact integers, FFT back: the
    negacyclic product (the property that makes this THE anticyclic transform, bit-reversed order
    and all), compared with the exact integer product."""
    N = 64
    rs = np.random.RandomState(4)
    pa, pb = O.hp_twiddles(2 * N)
    a = rs.randint(-2 ** 40, 2 ** 40, size=N, dtype=np.int64)
    b = rs.randint(-2 ** 40, 2 ** 40, size=N, dtype=np.int64)
    A, B = O.hp_ifft(a, pa), O.hp_ifft(b, pa)
    ar, ai = signed128(A[:, 0:2]), signed128(A[:, 2:4])
    br, bi = signed128(B[:, 0:2]), signed128(B[:, 2:4])
    sh = 30                                   # keep the products inside 96 bits
    spec = np.array([to_u128((int(ar[j]) * int(br[j]) - int(ai[j]) * int(bi[j])) >> sh) +
                     to_u128((int(ar[j]) * int(bi[j]) + int(ai[j]) * int(br[j])) >> sh) for j in range(N // 2)], np.uint64)
    got = O.hp_fft(spec, pb)
    exact = [0] * N
    for i in range(N):
        for j in range(N):
            k, v = i + j, int(a[i]) * int(b[j])
            if k >= N:
                k, v = k - N, -v
            exact[k] += v
    for k in range(N):
        want = exact[k] >> sh
        want = (want + 2 ** 63) % 2 ** 64 - 2 ** 63
        assert abs(int(got[k]) - want) <= 2 ** 16, (k, int(got[k]), want)


# ------------------------------------------------------------------ kernels vs restatement
def check_hp_kernels_wide(lib_path, N, B):
    """a batch of many polynomials (one workgroup each): a subset against the restatement, the duplicated first rows at the far
    end of the batch, and the round trip on all of them"""
    rs = np.random.RandomState(60 + N)
    pa, pb = O.hp_twiddles(2 * N)
    x = np.frombuffer(rs.bytes(B * N * 8), dtype=np.int64).reshape(B, N).copy()
    x[B - 5:] = x[:5]
    e = T.Engine(torus_bits=64, n=1, N=N, l=2, Bgbit=8, ks_t=0, lib_path=lib_path)
    try:
        spec = e.hp_ifft(x)
        sub = sorted(set([0, 1, B // 2, B - 6, B - 1]) | set(rs.choice(B, 6, replace=False).tolist()))
        for b in sub:
            assert np.array_equal(spec[b], O.hp_ifft(x[b], pa)), f"Real96 iFFT, polynomial {b} of {B}"
        assert np.array_equal(spec[B - 5:], spec[:5])
        back = e.hp_fft(spec)
        for b in sub:
            assert np.array_equal(back[b], O.hp_fft(spec[b], pb)), f"Real96 FFT, polynomial {b} of {B}"
        assert np.array_equal(back[B - 5:], back[:5])
        # FFT(iFFT(x)) returns x up to the transforms' truncations (HP's own round-trip check, very_close)
        assert np.abs((back - x).astype(np.int64)).max() < 2 ** 16
    finally:
        e.close()


def check_hp_kernels(lib_path, N, B):
    rs = np.random.RandomState(6)
    pa, pb = O.hp_twiddles(2 * N)
    x = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(B, N), dtype=np.int64)
    x[0] = 0
    x[1] = -2 ** 63
    x[2] = 2 ** 63 - 1
    e = T.Engine(torus_bits=64, n=1, N=N, l=2, Bgbit=8, ks_t=0, lib_path=lib_path)
    try:
        spec = e.hp_ifft(x)
        want = np.stack([O.hp_ifft(x[b], pa) for b in range(B)])
        assert np.array_equal(spec, want), "Real96 iFFT"
        back = e.hp_fft(spec)
        assert np.array_equal(back, np.stack([O.hp_fft(want[b], pb) for b in range(B)])), "Real96 FFT"
        rnd = np.frombuffer(rs.bytes(B * (N // 2) * 32), np.uint64).reshape(B, N // 2, 4)  # arbitrary 128-bit inputs
        assert np.array_equal(e.hp_fft(rnd), np.stack([O.hp_fft(rnd[b], pb) for b in range(B)])), "Real96 FFT, random input"
    finally:
        e.close()


@pytest.mark.parametrize("N", [1024, 2048])
def test_hp_kernels_emu(emu_lib, N):
    check_hp_kernels(emu_lib, N, B=3)


def test_hp_kernels_wider_batch_emu(emu_lib):
    check_hp_kernels_wide(emu_lib, 2048, B=11)
    check_hp_kernels_wide(emu_lib, 1024, B=10)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1024, 2048])
def test_hp_kernels_gpu(gpu_lib, N):
    check_hp_kernels(gpu_lib, N, B=9)
    check_hp_kernels_wide(gpu_lib, N, B=1301)
