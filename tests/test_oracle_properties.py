"""Domain properties of the oracle's composed path (the library-form functions the
reference cannot compile): decrypt-correctness, ring identities, FFT error vs exact product."""
import numpy as np

import oracle_py as O

SEED = 0x5446484500000001


def test_rotation_identities():
    rs = np.random.RandomState(1)
    N = 1024
    p = rs.randint(-2 ** 31, 2 ** 31, size=N).astype(np.int32)
    for a in (0, 1, 5, N - 1, N, N + 3, 2 * N - 1):
        rot = O.mul_xai32(a, p)
        want = ((rot.astype(np.int64) - p) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        assert np.array_equal(O.mul_xai_minus_one32(a, p), want)
        # X^a * X^(2N-a) = 1
        assert np.array_equal(O.mul_xai32((2 * N - a) % (2 * N), rot), p)
    assert np.array_equal(O.mul_xai32(N, p), (-p.astype(np.int64)).astype(np.int32))  # X^N = -1
    p64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=N, dtype=np.int64)
    assert np.array_equal(O.mul_xai64(3, O.mul_xai64(2 * N - 3, p64)), p64)


def test_decomposition_recomposes():
    rs = np.random.RandomState(2)
    N, l, Bgbit = 1024, 2, 10
    p = rs.randint(-2 ** 31, 2 ** 31, size=N).astype(np.int32)
    d = O.decomp32(p, l, Bgbit).astype(np.int64)
    assert d.min() >= -(1 << (Bgbit - 1)) and d.max() < (1 << (Bgbit - 1))
    rec = sum(d[i] << (32 - (i + 1) * Bgbit) for i in range(l))
    err = ((rec - p.astype(np.int64) + 2 ** 31) % 2 ** 32) - 2 ** 31
    assert np.abs(err).max() <= 1 << (32 - l * Bgbit)  # truncation error below the last digit (no rounding bit)
    l2, bg2 = 4, 9
    p64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=N, dtype=np.int64)
    d = O.decomp64(p64, l2, bg2)
    rec = np.zeros(N, dtype=object)
    for i in range(l2):
        rec = rec + d[i].astype(object) * (1 << (64 - (i + 1) * bg2))
    err = np.array([((int(r) - int(v) + 2 ** 63) % 2 ** 64) - 2 ** 63 for r, v in zip(rec, p64)], dtype=object)
    assert max(abs(int(e)) for e in err) <= 1 << (63 - l2 * bg2)  # rounded to nearest (rounding bit in the offset)


def test_fft_product_close_to_exact_product():
    """|FFT product - exact negacyclic product| stays far below the noise budget (the reference
    only checks |naive - karatsuba| <= 1: PAR/test_parallel_multiplications.cpp:62-143)."""
    rs = np.random.RandomState(3)
    for N, lim in ((1024, 512), (2048, 256)):
        d = rs.randint(-lim, lim, size=N).astype(np.int32)
        t = rs.randint(-2 ** 31, 2 ** 31, size=N).astype(np.int32)
        got = O.execute_direct_torus32(N, O.lagrange_addmul(N, np.zeros(N), O.execute_reverse_int(N, d),
                                                            O.execute_reverse_int(N, t)))
        exact = O.negacyclic_mul32(d, t)
        err = ((got.astype(np.int64) - exact + 2 ** 31) % 2 ** 32) - 2 ** 31
        assert np.abs(err).max() <= 2, np.abs(err).max()


def test_gate_bootstrap_decrypts():
    N, n, l, Bgbit, t, bb = 1024, 40, 2, 10, 8, 2
    lwe_key, tkey = O.keygen_binary(n, SEED, 1), O.keygen_binary(N, SEED, 2)
    bk = O.bk_create32(N, lwe_key, tkey, l, Bgbit, 2.0 ** -25, SEED, 1000)
    ks = O.ks_create32(tkey, lwe_key, t, bb, 2.0 ** -15, SEED, 100000)
    mu = 1 << 29
    for i in range(8):
        m = mu if i % 2 else -mu
        ct = O.lwe_encrypt32(m, 2.0 ** -15, lwe_key, O.rng(SEED, 50 + i))
        assert abs(O.lwe_phase32(ct, lwe_key) - m) < 2 ** 24
        out = O.bootstrap32(N, bk, ks, mu, ct, l, Bgbit, t, bb)
        ph = O.lwe_phase32(out, lwe_key)
        assert (ph > 0) == (m > 0) and abs(abs(ph) - mu) < mu // 2
        # woKS output decrypts under the extracted (TLWE) key
        u = O.bootstrap_woks32(N, bk, mu, ct, l, Bgbit)
        ph = O.lwe_phase32(u, tkey)
        assert (ph > 0) == (m > 0) and abs(abs(ph) - mu) < mu // 4


def test_extprod_is_cmux_selector():
    """external product with a TGSW encryption of 1 / 0 returns the TLWE message / zero"""
    N, l, Bgbit = 1024, 2, 10
    tkey = O.keygen_binary(N, SEED, 2)
    rs = np.random.RandomState(5)
    msg = (rs.randint(-4, 4, size=N).astype(np.int64) << 28).astype(np.int32)
    tl = np.zeros(2 * N, np.int32)
    tl[N:] = msg  # noiseless trivial TLWE of msg
    for bit in (0, 1):
        bk = O.bk_create32(N, np.array([bit], np.int32), tkey, l, Bgbit, 2.0 ** -30, SEED, 77)
        out = O.extprod32(N, tl, bk[0], l, Bgbit)
        ph = O.tlwe_phase32(out, tkey).astype(np.int64)
        err = ((ph - bit * msg.astype(np.int64) + 2 ** 31) % 2 ** 32) - 2 ** 31
        assert np.abs(err).max() < 2 ** 24


def test_circuit_bootstrap_small():
    """tfhe_CircuitBootstrapFFT (library rotation semantics) yields TGSW rows whose phases are
    bit * 2^(32-(w+1)Bgbit1) on a[u] * key -- reduced ring sizes (the oracle is generic in N) so
    that generating the private key-switch key takes milliseconds instead of minutes."""
    n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21 = 8, 64, 128, 2, 8, 4, 9, 6, 2, 10, 3
    key0, key1 = O.keygen_binary(n0, SEED, 21), O.keygen_binary(N1, SEED, 22)
    key2 = O.keygen_binary(N2, SEED, 23)
    bk = O.bk_create64(N2, key0, key2, l2, bg2, 2.0 ** -44, SEED, 3000)
    preks = O.ks_create32(key1, key0, t10, bb10, 2.0 ** -20, SEED, 4000)
    privks = O.privks_create(key2, key1, t21, bb21, 2.0 ** -31, SEED, 5000)
    for bit in (0, 1):
        x = O.lwe_encrypt32(bit << 31, 2.0 ** -20, key1, O.rng(SEED, 60 + bit))  # message in {0, 1/2}
        out = O.circuit_bootstrap(x, preks, bk, privks, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21)
        for w in range(l1):
            h = 1 << (32 - (w + 1) * bg1)
            # u = 1: plain TLWE of bit*h (constant polynomial); u = 0: TLWE of -bit*h*key1
            ph1 = O.tlwe_phase32(out[1, w].ravel(), key1).astype(np.int64)
            want1 = np.zeros(N1, np.int64)
            want1[0] = bit * h
            err = ((ph1 - want1 + 2 ** 31) % 2 ** 32) - 2 ** 31
            assert np.abs(err).max() < h // 4, (bit, w, np.abs(err).max(), h)
            ph0 = O.tlwe_phase32(out[0, w].ravel(), key1).astype(np.int64)
            want0 = -bit * h * key1.astype(np.int64)
            err = ((ph0 - want0 + 2 ** 31) % 2 ** 32) - 2 ** 31
            assert np.abs(err).max() < h // 4, (bit, w, np.abs(err).max(), h)


def test_circuit_bootstrap_feeds_lut_small():
    """The chain BASELINE config 3 names: LWE-encrypted bits -> tfhe_CircuitBootstrapFFT -> TGSW32
    -> LUT evaluation by vertical packing -> LWE of f(bits).  Reduced rings as above (N1 = 64, so 8
    bits = 6 rotation steps + a 2-level CMux tree); decrypts to the table entry for every input."""
    n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21 = 8, 64, 128, 2, 8, 4, 9, 6, 2, 10, 3
    d = 8
    key0, key1 = O.keygen_binary(n0, SEED, 21), O.keygen_binary(N1, SEED, 22)
    key2 = O.keygen_binary(N2, SEED, 23)
    bk = O.bk_create64(N2, key0, key2, l2, bg2, 2.0 ** -44, SEED, 3000)
    preks = O.ks_create32(key1, key0, t10, bb10, 2.0 ** -20, SEED, 4000)
    privks = O.privks_create(key2, key1, t21, bb21, 2.0 ** -31, SEED, 5000)
    rs = np.random.RandomState(5)
    table = (rs.randint(-4, 4, size=1 << d).astype(np.int64) << 28).astype(np.int32)
    for x in (0, (1 << d) - 1, 0b10110010):
        sel = []
        for i in range(d):
            bit = (x >> i) & 1
            ct = O.lwe_encrypt32(bit << 31, 2.0 ** -20, key1, O.rng(SEED, 700 + 16 * x + i))
            tgsw = O.circuit_bootstrap(ct, preks, bk, privks, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21)
            sel.append(O.execute_reverse_int(N1, np.asarray(tgsw, np.int32).reshape(-1, N1)))  # tGswToFFTConvert
        out = O.lut_eval32(N1, np.stack(sel), d, table, l1, bg1)
        err = (O.lwe_phase32(out, key1) - int(table[x]) + 2 ** 31) % 2 ** 32 - 2 ** 31
        assert abs(err) < 2 ** 26, (x, err)
