"""Parity checks shared by the GPU tests (real library, -m gpu) and the CPU-side kernel
emulation tests (tests/emu build of the same sources).  Every check drives the C ABI
(include/tfhe_amd.h) through the ctypes binding and compares with the oracle bit for bit:
integer torus outputs with array_equal, Lagrange-domain doubles on their raw 64-bit
patterns (+0.0 / -0.0 are the one tolerated difference: a sign of zero never reaches a
torus value)."""
import importlib

import numpy as np

import oracle_py as O

T = importlib.import_module("experimental-tfhe_amd")

SEED = 0x5446484500000001


def same_doubles(a, b):
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    if a.shape != b.shape:
        return False
    ua, ub = a.view(np.uint64).copy(), b.view(np.uint64).copy()
    z = np.uint64(0x8000000000000000)
    ua[ua == z] = 0  # -0.0 -> +0.0
    ub[ub == z] = 0
    return bool(np.array_equal(ua, ub))


class GateSetup:
    """keys + engine for one Torus32 parameter set; keys come from the ORACLE's generator and
    are cross-checked against the library's own generator (same PRNG specification)."""

    def __init__(self, lib_path, N, n, l, Bgbit, ks_t, ks_bb, bk_stdev=2.0 ** -25, ks_stdev=2.0 ** -15, seed=SEED, device=0):
        self.N, self.n, self.l, self.Bgbit, self.ks_t, self.ks_bb = N, n, l, Bgbit, ks_t, ks_bb
        self.lib_path, self.seed = lib_path, seed
        self.lwe_key = O.keygen_binary(n, seed, 1)
        self.tkey = O.keygen_binary(N, seed, 2)
        self.bk = O.bk_create32(N, self.lwe_key, self.tkey, l, Bgbit, bk_stdev, seed, 1000)
        self.ks = O.ks_create32(self.tkey, self.lwe_key, ks_t, ks_bb, ks_stdev, seed, 100000)
        self.bk_stdev, self.ks_stdev = bk_stdev, ks_stdev
        self.eng = T.Engine(torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=ks_t, ks_basebit=ks_bb, device=device, lib_path=lib_path)
        self.gsw = self.eng.gsw_from_fft(self.bk)
        self.eng.set_bootstrap_key(self.gsw)
        self.eng.load_keyswitch_key(self.ks)

    def close(self):
        self.eng.close()

    def encrypt(self, messages, stdev=2.0 ** -15, stream0=5000):
        return np.stack([O.lwe_encrypt32(int(m), stdev, self.lwe_key, O.rng(self.seed, stream0 + i))
                         for i, m in enumerate(messages)])


# ---------------------------------------------------------------- L1 plugin
def check_fft_plugin(lib_path, N, count=4, seed=11):
    rs = np.random.RandomState(seed)
    e = T.Engine(torus_bits=32, n=1, N=N, l=2, Bgbit=10, ks_t=0, lib_path=lib_path)
    try:
        f, r = e.tables()
        fo, ro = O.table_arrays(N)
        assert np.array_equal(f.view(np.uint64), fo.view(np.uint64)), "fft twiddle table"
        assert np.array_equal(r.view(np.uint64), ro.view(np.uint64)), "ifft twiddle table"
        a32 = rs.randint(-2 ** 31, 2 ** 31, size=(count, N)).astype(np.int32)
        dig = rs.randint(-512, 512, size=(count, N)).astype(np.int32)
        a64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(count, N), dtype=np.int64)
        edge = np.zeros((4, N), np.int32)  # empty / extreme inputs
        edge[1] = 2 ** 31 - 1
        edge[2] = -2 ** 31
        edge[3, 0] = 1
        for x in (a32, dig, edge):
            assert same_doubles(e.ifft_int32(x), O.execute_reverse_int(N, x)), "execute_reverse_int"
        assert same_doubles(e.ifft_torus64(a64), O.execute_reverse_torus64(N, a64)), "execute_reverse_torus64"
        lag32 = O.lagrange_addmul(N, np.zeros((count, N)), O.execute_reverse_int(N, dig), O.execute_reverse_int(N, a32))
        assert np.array_equal(e.fft_torus32(lag32), O.execute_direct_torus32(N, lag32)), "execute_direct_torus32"
        lag64 = O.lagrange_addmul(N, np.zeros((count, N)), O.execute_reverse_int(N, dig),
                                  O.execute_reverse_torus64(N, a64))
        assert np.array_equal(e.fft_torus64(lag64), O.execute_direct_torus64(N, lag64)), "execute_direct_torus64"
        # execute_direct_torus64's shift ranges: results far below 1 (right shifts up to and beyond 63),
        # around 2^64 (wrap) and far above (left shifts of 64 or more: defined as 0)
        for sc in (2.0 ** -70, 2.0 ** -52, 2.0 ** 12, 2.0 ** 30, 2.0 ** 70):
            assert np.array_equal(e.fft_torus64(lag64 * sc), O.execute_direct_torus64(N, lag64 * sc)), f"direct_torus64 x {sc}"
        # execute_direct_torus32 around and beyond the 2^51 bound of the short rounding sequence (per-wave fallback)
        for sc in (2.0 ** -40, 2.0 ** 8, 2.0 ** 11, 2.0 ** 14, 2.0 ** 40):
            assert np.array_equal(e.fft_torus32(lag32 * sc), O.execute_direct_torus32(N, lag32 * sc)), f"direct_torus32 x {sc}"
        # the bare core transforms (spqlios-fft.h `ifft` / `fft`: doubles in, doubles out, no scale, no rounding)
        raw = rs.standard_normal((count, N)) * 2.0 ** 20
        assert same_doubles(e.ifft_f64(raw), O.ifft(N, raw)), "ifft (C core)"
        assert same_doubles(e.fft_f64(raw), O.fft(N, raw)), "fft (C core)"
        # spqlios-bench.cpp:76-77: fft(ifft(x)) = N/2 x (up to fp64 rounding)
        assert np.abs(e.fft_f64(e.ifft_f64(dig.astype(np.float64))) - dig.astype(np.float64) * (N / 2)).max() < 1e-6
        # host-only table builder == the context's tables
        fb, rb = T.build_tables(N, lib_path=lib_path)
        assert np.array_equal(fb.view(np.uint64), fo.view(np.uint64)) and np.array_equal(rb.view(np.uint64), ro.view(np.uint64))
        # Lagrange-domain inputs are read with 16-byte loads: a misaligned device pointer is refused, not mis-read
        d_in, d_out = e.to_device(np.zeros((2, N))), e.alloc(N * 8)
        assert e.lib.tfhe_amd_fft_torus64(e.ctx, d_out.ptr, d_in.ptr + 8, 1) == T.ERR_PARAM
        assert e.lib.tfhe_amd_fft_torus32(e.ctx, d_out.ptr, d_in.ptr + 8, 1) == T.ERR_PARAM
        assert e.lib.tfhe_amd_fft_f64(e.ctx, d_out.ptr, d_in.ptr + 8, 1) == T.ERR_PARAM
        d_in.free()
        d_out.free()
        assert np.array_equal(e.fft_torus32(np.zeros((1, N))), np.zeros((1, N), np.int32))
        assert np.array_equal(e.fft_torus64(np.zeros((1, N))), np.zeros((1, N), np.int64))
        got = e.lagrange_addmul(lag32, O.execute_reverse_int(N, dig), O.execute_reverse_int(N, a32))
        want = O.lagrange_addmul(N, lag32, O.execute_reverse_int(N, dig), O.execute_reverse_int(N, a32))
        assert same_doubles(got, want), "LagrangeHalfCPolynomialAddMul"
        got = e.lagrange_addmul(lag32, O.execute_reverse_int(N, dig), O.execute_reverse_int(N, a32[0]), b_shared=True)
        want = O.lagrange_addmul(N, lag32, O.execute_reverse_int(N, dig),
                                 np.repeat(O.execute_reverse_int(N, a32[:1]), count, axis=0))
        assert same_doubles(got, want), "AddMul with shared b"
        # round trip fft(ifft(a)) (spqlios-bench.cpp:76-77 identity with 2/N folded in): the
        # truncating conversion may lose one unit, and must lose it exactly where the oracle does
        rt = e.fft_torus32(e.ifft_int32(a32))
        assert np.array_equal(rt, O.execute_direct_torus32(N, O.execute_reverse_int(N, a32))), "round trip"
        assert np.abs(rt.astype(np.int64) - a32).max() <= 1
    finally:
        e.close()


# ------------------------------------------------------------ Torus32 path
def check_gate_path(lib_path, N, n, l, Bgbit, ks_t, ks_bb, B, seed=21, check_export=True, br_split=None):
    """br_split: None = the library's own choice of blind-rotation kernel; 0 = one wave per ciphertext
    (k_blind_rotate) whatever the batch; a large value = the latency-shaped kernel (k_blind_rotate_split)
    wherever it applies (Torus32, N = 1024, l = 2)."""
    rs = np.random.RandomState(seed)
    s = GateSetup(lib_path, N, n, l, Bgbit, ks_t, ks_bb)
    e = s.eng
    try:
        if br_split is not None:
            e.set_option(T.OPT_BR_SPLIT, br_split)
        # harness parity: the library's key generator == the oracle's (same PRNG spec)
        assert np.array_equal(T.keygen_binary(n, s.seed, 1, lib_path=lib_path), s.lwe_key)
        bkt = T.keygen_bk_torus(32, s.lwe_key, s.tkey, l, Bgbit, s.bk_stdev, s.seed, 1000, lib_path=lib_path)
        assert np.array_equal(T.keygen_ks32(s.tkey, s.lwe_key, ks_t, ks_bb, s.ks_stdev, s.seed, 100000,
                                            lib_path=lib_path), s.ks)
        if check_export:
            # tGswToFFTConvert on the GPU == oracle's execute_reverse_torus32 of every polynomial
            g2 = e.gsw_from_torus(bkt)
            for idx in (0, n - 1):
                assert same_doubles(e.gsw_export_fft(g2, idx), s.bk[idx]), "bk conversion"
                assert same_doubles(e.gsw_export_fft(s.gsw, idx), s.bk[idx]), "bk upload/export"
        acc = rs.randint(-2 ** 31, 2 ** 31, size=(B, 2, N)).astype(np.int32)
        # tGswFFTExternMulToTLwe
        want = np.stack([O.extprod32(N, acc[b], s.bk[n // 2], l, Bgbit) for b in range(B)]).reshape(B, 2, N)
        assert np.array_equal(e.extern_mul(acc, s.gsw, n // 2), want), "external product"
        # tfhe_MuxRotate_FFT (rotation 0 = identity)
        ba = rs.randint(1, 2 * N, size=B).astype(np.int32)
        ba[0] = 0
        want = np.stack([O.mux_rotate32(N, acc[b], s.bk[1 % n], ba[b], l, Bgbit) if ba[b] else acc[b]
                         for b in range(B)]).reshape(B, 2, N)
        assert np.array_equal(e.mux_rotate(acc, s.gsw, 1 % n, ba), want), "CMux"
        # tfhe_blindRotate_FFT, including skipped (zero) rotations and both halves of [0,2N)
        bara = rs.randint(0, 2 * N, size=(B, n)).astype(np.int32)
        bara[0, 0] = 0
        bara[B - 1, n - 1] = 0
        bara[0, 1 % n] = N
        bara[0, 2 % n] = 2 * N - 1
        want = np.stack([O.blind_rotate32(N, acc[b], s.bk, bara[b], l, Bgbit) for b in range(B)]).reshape(B, 2, N)
        assert np.array_equal(e.blind_rotate(acc, bara), want), "blind rotation"
        # tfhe_blindRotateAndExtract_FFT, shared and per-sample test vectors, barb = 0 edge
        rot = rs.randint(0, 2 * N, size=(B, n + 1)).astype(np.int32)
        rot[0, n] = 0
        v = rs.randint(-2 ** 31, 2 ** 31, size=N).astype(np.int32)
        want = np.stack([O.blind_rotate_extract32(N, v, s.bk, rot[b, n], rot[b, :n], l, Bgbit) for b in range(B)])
        assert np.array_equal(e.blind_rotate_extract(v, rot), want), "blind rotate + extract"
        vs = rs.randint(-2 ** 31, 2 ** 31, size=(B, N)).astype(np.int32)
        want = np.stack([O.blind_rotate_extract32(N, vs[b], s.bk, rot[b, n], rot[b, :n], l, Bgbit) for b in range(B)])
        assert np.array_equal(e.blind_rotate_extract(vs, rot), want), "per-sample test vectors"
        # tfhe_bootstrap_woKS_FFT / lweKeySwitch / tfhe_bootstrap_FFT on real ciphertexts
        mu = 1 << 29
        msgs = [mu if b % 2 else -mu for b in range(B)]
        x = s.encrypt(msgs)
        assert np.array_equal(T.lwe_encrypt32(msgs[0], 2.0 ** -15, s.lwe_key, s.seed, 5000, lib_path=lib_path), x[0])
        assert np.array_equal(e.modswitch(x), O.modswitch32(x, 2 * N)), "modSwitchFromTorus32"
        woks = np.stack([O.bootstrap_woks32(N, s.bk, mu, x[b], l, Bgbit) for b in range(B)])
        assert np.array_equal(e.bootstrap_woks(mu, x), woks), "bootstrap without key switch"
        ksw = np.stack([O.keyswitch32(s.ks, woks[b], N, n, ks_t, ks_bb) for b in range(B)])
        assert np.array_equal(e.keyswitch(woks), ksw), "key switch"
        full = np.stack([O.bootstrap32(N, s.bk, s.ks, mu, x[b], l, Bgbit, ks_t, ks_bb) for b in range(B)])
        assert np.array_equal(full, ksw)
        assert np.array_equal(e.bootstrap(mu, x), full), "bootstrap"
        assert np.array_equal(e.bootstrap(mu, x, streamed=True), full), "bootstrap, one launch per CMux"
        assert np.array_equal(e.bootstrap_host(mu, x), full), "bootstrap, host pointers"
        # uniformly random 'ciphertexts' (throughput workload): still bit-identical
        xr = rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32)
        want = np.stack([O.bootstrap32(N, s.bk, s.ks, mu, xr[b], l, Bgbit, ks_t, ks_bb) for b in range(B)])
        assert np.array_equal(e.bootstrap(mu, xr), want), "bootstrap on random samples"
        # empty batch is a no-op
        assert e.bootstrap(mu, np.zeros((0, n + 1), np.int32)).shape == (0, n + 1)
        return s, x, msgs, full
    finally:
        s.close()


def check_gate_wide_batch(lib_path, l, Bgbit, B=1031, N=1024, n=6, ks_t=8, ks_bb=2, seed=31):
    """batches above 1024 run k_blind_rotate in 8-wave workgroups (two waves per SIMD, the partner balance): bootstrap and
    in-place blind rotation of B samples against the oracle on a subset, and against the 4-wave form of the same kernel
    (a small batch of the same inputs) on the first rows"""
    s = GateSetup(lib_path, N, n, l, Bgbit, ks_t, ks_bb)
    try:
        rs = np.random.RandomState(seed)
        s.eng.set_option(T.OPT_BR_SPLIT, 0)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32)
        got = s.eng.bootstrap(1 << 29, x)
        sub = sorted(set([0, 7, 8, B - 8, B - 7, B - 1] + [int(v) for v in rs.choice(B, 4, replace=False)]))
        want = np.stack([O.bootstrap32(N, s.bk, s.ks, 1 << 29, x[i], l, Bgbit, ks_t, ks_bb) for i in sub])
        assert np.array_equal(got[sub], want), (l, Bgbit)
        assert np.array_equal(got[:24], s.eng.bootstrap(1 << 29, x[:24])), (l, Bgbit, "8-wave vs 4-wave workgroups")
        # the one-launch-per-CMux schedule at this batch: for l = 2 its launches take k_cmux_stream (accumulators in place in global
        # memory, persistent waves: several ciphertexts per wave when the batch exceeds what the chip holds), plain and as a graph
        assert np.array_equal(s.eng.bootstrap(1 << 29, x, streamed=True), got), (l, Bgbit, "one launch per CMux")
        # ... and one CMux step on its own (tfhe_MuxRotate_FFT), rotation 0 (skipped) included
        accm = rs.randint(-2 ** 31, 2 ** 31, size=(B, 2, N)).astype(np.int32)
        ba = rs.randint(0, 2 * N, size=B).astype(np.int32)
        ba[B - 1] = 0
        mr = s.eng.mux_rotate(accm, s.gsw, n - 1, ba)
        for i in (0, 5, B - 2, B - 1):
            w = O.mux_rotate32(N, accm[i], s.bk[n - 1], ba[i], l, Bgbit).reshape(2, N) if ba[i] else accm[i]
            assert np.array_equal(mr[i], w), (l, Bgbit, i, "CMux step")
        acc = rs.randint(-2 ** 31, 2 ** 31, size=(B, 2, N)).astype(np.int32)
        bara = rs.randint(0, 2 * N, size=(B, n)).astype(np.int32)
        bara[B - 1, 0] = 0
        rot = s.eng.blind_rotate(acc, bara)
        for i in (0, B - 2, B - 1):
            assert np.array_equal(rot[i], O.blind_rotate32(N, acc[i], s.bk, bara[i], l, Bgbit).reshape(2, N)), (l, Bgbit, i)
    finally:
        s.close()


# ------------------------------------------------------------ Torus64 path
def check_torus64_path(lib_path, N, n, l, Bgbit, B, seed=31):
    rs = np.random.RandomState(seed)
    key0 = O.keygen_binary(n, SEED, 11)
    tkey = O.keygen_binary(N, SEED, 12)
    bk = O.bk_create64(N, key0, tkey, l, Bgbit, 2.0 ** -44, SEED, 2000)
    e = T.Engine(torus_bits=64, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=0, lib_path=lib_path)
    try:
        bkt = T.keygen_bk_torus(64, key0, tkey, l, Bgbit, 2.0 ** -44, SEED, 2000, lib_path=lib_path)
        g = e.gsw_from_torus(bkt)
        assert same_doubles(e.gsw_export_fft(g, n - 1), bk[n - 1]), "Torus64 bk conversion"
        e.set_bootstrap_key(g)
        acc = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(B, 2, N), dtype=np.int64)
        want = np.stack([O.extprod64(N, acc[b], bk[0], l, Bgbit) for b in range(B)]).reshape(B, 2, N)
        assert np.array_equal(e.extern_mul(acc, g, 0), want), "Torus64 external product"
        bara = rs.randint(0, 2 * N, size=(B, n)).astype(np.int32)
        bara[0, 0] = 0
        bara[:, n // 2] = 0          # a step every sample skips
        bara[B - 1, n - 1] = 0       # skipped steps at both ends of the batch
        bara[0, n - 1] = N
        want = np.stack([O.blind_rotate64(N, acc[b], bk, bara[b], l, Bgbit) for b in range(B)]).reshape(B, 2, N)
        assert np.array_equal(e.blind_rotate(acc, bara), want), "Torus64 blind rotation"
        abar = rs.randint(0, 2 * N, size=(B, n + 1)).astype(np.int32)
        mu = 1 << 56
        want = np.stack([O.cb_bootstrap_woks64(N, mu, abar[b], bk, l, Bgbit) for b in range(B)])
        assert np.array_equal(e.cb_bootstrap_woks(mu, abar), want), "circuitBootstrapWoKS"
    finally:
        e.close()


# ------------------------------------------------------------------ CMux on data
def check_cmux_data(lib_path, N=1024, l=2, Bgbit=8, B=11, seed=71):
    """out = gsw[sel] (x) (d1 - d0) + d0 with a per-sample TGSW selector (one level of a
    vertical-packing tree over circuit-bootstrap outputs, gadget l1=2, Bgbit1=8 as in poc:70-85),
    against the oracle's external product, plus the functional property: TGSW(1) selects d1,
    TGSW(0) selects d0."""
    rs = np.random.RandomState(seed)
    tkey = O.keygen_binary(N, SEED, 2)
    bits = np.array([0, 1, 1, 0], np.int32)
    gsw_lag = O.bk_create32(N, bits, tkey, l, Bgbit, 2.0 ** -30, SEED, 9000)   # TGSW encryptions of the bits
    e = T.Engine(torus_bits=32, n=4, N=N, l=l, Bgbit=Bgbit, ks_t=0, lib_path=lib_path)
    try:
        g = e.gsw_from_fft(gsw_lag)
        d0 = rs.randint(-2 ** 31, 2 ** 31, size=(B, 2, N)).astype(np.int32)
        d1 = rs.randint(-2 ** 31, 2 ** 31, size=(B, 2, N)).astype(np.int32)
        sel = rs.randint(0, 4, size=B).astype(np.int32)

        def want_for(s_, a0, a1):
            diff = ((a1.astype(np.int64) - a0) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
            prod = O.extprod32(N, diff, gsw_lag[s_], l, Bgbit).reshape(2, N)
            return ((prod.astype(np.int64) + a0) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)

        want = np.stack([want_for(sel[b], d0[b], d1[b]) for b in range(B)])
        assert np.array_equal(e.cmux(g, sel, d0, d1), want), "CMux on data"
        want0 = np.stack([want_for(0, d0[b], d1[b]) for b in range(B)])
        assert np.array_equal(e.cmux(g, None, d0, d1), want0), "CMux on data, default selector"
        # functional check on noiseless trivial TLWE samples (a = 0, b = message polynomial)
        m0 = (rs.randint(-4, 4, size=(B, N)).astype(np.int64) << 28).astype(np.int32)
        m1 = (rs.randint(-4, 4, size=(B, N)).astype(np.int64) << 28).astype(np.int32)
        t0, t1 = np.zeros((B, 2, N), np.int32), np.zeros((B, 2, N), np.int32)
        t0[:, 1], t1[:, 1] = m0, m1
        out = e.cmux(g, sel, t0, t1)
        for b in range(B):
            ph = O.tlwe_phase32(out[b].ravel(), tkey).astype(np.int64)
            chosen = (m1 if bits[sel[b]] else m0)[b].astype(np.int64)
            err = ((ph - chosen + 2 ** 31) % 2 ** 32) - 2 ** 31
            assert np.abs(err).max() < 2 ** 24, (b, np.abs(err).max())
    finally:
        e.close()


def check_lut_eval(lib_path, N=1024, l=3, Bgbit=8, d=12, B=3, seed=81, decrypt_tol=2 ** 25):
    """LUT evaluation by vertical packing (tfhe_amd_lut_eval): d TGSW-encrypted bits per item select
    f(x) out of a 2^d-entry table.  Bit-compared with the oracle's composition of the reference's
    external product / MuxRotate / sample extraction, and decrypt-checked: the phase of the result
    is the table entry of the item's bits.  The library's Torus32 decomposition truncates (offset
    without a rounding bit, tgsw_functions.cpp:24-36), so every CMux adds a bias of about
    N/2 * 2^(32 - l*Bgbit - 1) to the phase: with the PoC's l1=2, Bgbit1=8 that is 2^24 per level and
    the decrypt check is only meaningful at l=3 (pass decrypt_tol=None to bit-compare only)."""
    rs = np.random.RandomState(seed)
    logn = N.bit_length() - 1
    tkey = O.keygen_binary(N, SEED, 2)
    x = rs.randint(0, 1 << d, size=B)
    x[0] = (1 << d) - 1  # all bits set: every level takes the "1" branch, every rotation applies
    bits = np.array([[(int(v) >> i) & 1 for i in range(d)] for v in x], np.int32)
    gsw_lag = O.bk_create32(N, bits.ravel(), tkey, l, Bgbit, 2.0 ** -30, SEED, 9100)  # [B*d][2l][2][N]
    npoly = 1 << max(0, d - logn)
    table = (rs.randint(-4, 4, size=npoly * N).astype(np.int64) << 28).astype(np.int32)  # f(x), x < npoly*N
    e = T.Engine(torus_bits=32, n=1, N=N, l=l, Bgbit=Bgbit, ks_t=0, lib_path=lib_path)
    try:
        g = e.gsw_from_fft(gsw_lag)
        got = e.lut_eval(g, d, table.reshape(npoly, N), B)
        per_item = gsw_lag.reshape(B, d, 2 * l, 2, N)
        want = np.stack([O.lut_eval32(N, per_item[b], d, table, l, Bgbit) for b in range(B)])
        assert np.array_equal(got, want), f"LUT evaluation d={d}"
        for b in range(B if decrypt_tol else 0):
            err = (O.lwe_phase32(got[b], tkey) - int(table[x[b]]) + 2 ** 31) % 2 ** 32 - 2 ** 31
            assert abs(err) < decrypt_tol, (b, x[b], err)
    finally:
        e.close()


def check_streamed_graph(lib_path, N=1024, n=12, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=5, seed=91):
    """tfhe_amd_bootstrap_streamed under TFHE_AMD_OPT_STREAMED_GRAPH: call 1 runs plain launches, call 2
    captures the n+3 launches into a hipGraph, call 3 replays it; new data written into the SAME device
    buffers must be picked up by the replay; a different batch size must re-capture.  (The CPU emulator
    build records and replays the graph too -- EmuGraph in tests/emu/emu_runtime.cpp -- so the capture and
    replay logic of the host code runs there as well.)"""
    s = GateSetup(lib_path, N, n, l, Bgbit, ks_t, ks_bb)
    rs = np.random.RandomState(seed)
    e = s.eng
    try:
        e.set_option(T.OPT_STREAMED_GRAPH, 1)
        mu = 1 << 29
        xa = rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32)
        xb = rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32)
        want_a, want_b = e.bootstrap(mu, xa), e.bootstrap(mu, xb)   # persistent kernel (parity-checked elsewhere)
        assert np.array_equal(want_a[0], O.bootstrap32(N, s.bk, s.ks, mu, xa[0], l, Bgbit, ks_t, ks_bb))
        x_d, out_d = e.to_device(xa), e.alloc(xa.nbytes)
        for call in range(3):
            e._chk(e.lib.tfhe_amd_bootstrap_streamed(e.ctx, out_d.ptr, mu, x_d.ptr, B))
            assert np.array_equal(out_d.download(np.int32, xa.shape), want_a), f"streamed call {call}"
        x_d.upload(xb)
        e._chk(e.lib.tfhe_amd_bootstrap_streamed(e.ctx, out_d.ptr, mu, x_d.ptr, B))
        assert np.array_equal(out_d.download(np.int32, xb.shape), want_b), "replay on new data in the same buffers"
        e._chk(e.lib.tfhe_amd_bootstrap_streamed(e.ctx, out_d.ptr, mu, x_d.ptr, B - 2))   # other batch: re-capture
        assert np.array_equal(out_d.download(np.int32, xb.shape)[:B - 2], want_b[:B - 2]), "re-captured for a smaller batch"
        e._chk(e.lib.tfhe_amd_bootstrap_streamed(e.ctx, out_d.ptr, -mu, x_d.ptr, B - 2))  # other mu: re-capture
        assert np.array_equal(out_d.download(np.int32, xb.shape)[:B - 2], e.bootstrap(-mu, xb[:B - 2])), "re-captured for another mu"
    finally:
        s.close()


def check_streamed_graph_batch_classes(lib_path, N=1024, n=6, l=2, Bgbit=10, ks_t=8, ks_bb=2, big=1031, small=8, seed=92, br_split=None,
                                       order=None):
    """graph mode with NO prior plain bootstrap on the context, at a batch served by the 8-wave kernel and then at one
    served by the latency-shaped kernel (another kernel class: its first launch sets an LDS attribute, which must not
    happen inside the capture -- the warm-up is per class), each: warm-up call, capturing call, replay.
    br_split / order: the latency-shaped kernel for EVERY batch, small batch first -- the schedule's two 0-step launches then
    change kernel (4-wave -> 8-wave form) while the 1-step class stays: the warm-up is keyed on both (the emulator aborts on a
    hipFuncSetAttribute inside a capture)"""
    s = GateSetup(lib_path, N, n, l, Bgbit, ks_t, ks_bb)
    rs = np.random.RandomState(seed)
    e = s.eng
    try:
        if br_split is not None:
            e.set_option(T.OPT_BR_SPLIT, br_split)
        e.set_option(T.OPT_STREAMED_GRAPH, 1)
        mu = 1 << 29
        x = rs.randint(-2 ** 31, 2 ** 31, size=(big, n + 1)).astype(np.int32)
        x_d, out_d = e.to_device(x), e.alloc(x.nbytes)
        got = {}
        for B in (order or (big, small)):
            for call in range(3):
                e._chk(e.lib.tfhe_amd_bootstrap_streamed(e.ctx, out_d.ptr, mu, x_d.ptr, B))
                got[(B, call)] = out_d.download(np.int32, x.shape)[:B].copy()
        e.set_option(T.OPT_STREAMED_GRAPH, 0)
        want = e.bootstrap(mu, x)
        for (B, call), g in got.items():
            assert np.array_equal(g, want[:B]), (B, call)
        for i in (0, big - 1):
            assert np.array_equal(want[i], O.bootstrap32(N, s.bk, s.ks, mu, x[i], l, Bgbit, ks_t, ks_bb))
    finally:
        s.close()


def check_exact_extprod(lib_path, torus_bits=32, N=1024, l=2, Bgbit=10, B=3, seed=95, fft_bound=None):
    """tfhe_amd_extern_mul_exact (the reference's FFT-free backend, poc:285-316) against the oracle's exact
    negacyclic products, bit for bit; and the fp64 external product against it: the difference is the
    rounding noise of the transforms, bounded by `fft_bound` torus units."""
    rs = np.random.RandomState(seed)
    tkey = O.keygen_binary(N, SEED, 2)
    stdev = 2.0 ** -25 if torus_bits == 32 else 2.0 ** -44
    gsw_t = T.keygen_bk_torus(torus_bits, np.array([1], np.int32), tkey, l, Bgbit, stdev, SEED, 9500, lib_path=lib_path)[0]
    e = T.Engine(torus_bits=torus_bits, n=1, N=N, l=l, Bgbit=Bgbit, ks_t=0, lib_path=lib_path)
    try:
        if torus_bits == 32:
            acc = rs.randint(-2 ** 31, 2 ** 31, size=(B, 2, N)).astype(np.int32)
            want = np.stack([O.extprod_exact32(N, acc[b].ravel(), gsw_t, l, Bgbit) for b in range(B)]).reshape(B, 2, N)
        else:
            acc = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(B, 2, N), dtype=np.int64)
            want = np.stack([O.extprod_exact64(N, acc[b].ravel(), gsw_t, l, Bgbit) for b in range(B)]).reshape(B, 2, N)
        got = e.extern_mul_exact(acc, gsw_t)
        assert np.array_equal(got, want), "exact external product"
        approx = e.extern_mul(acc, e.gsw_from_torus(gsw_t[None]), 0)
        diff = (approx.astype(object) - got.astype(object) + 2 ** (torus_bits - 1)) % 2 ** torus_bits - 2 ** (torus_bits - 1)
        worst = max(abs(int(v)) for v in diff.ravel())
        if fft_bound is not None:
            assert worst <= fft_bound, f"fp64 external product is {worst} torus units from the exact one (bound {fft_bound})"
        return worst
    finally:
        e.close()


# ------------------------------------------------------- rounding range extremes
def check_rounding_extremes(lib_path):
    """Torus32 rounding has a short sequence valid for |x| < 2^51 and an exact fallback (DESIGN.md,
    bit-exactness rules).  Drive the external product to its worst-case magnitude 2*l*N*(Bg/2)*2^31 =
    2^52: all digits -Bg/2, all key coefficients -2^31, which no random input
    reaches; plus mixed samples around the 2^51 threshold, in one batch (the choice is per wave)."""
    N, l, Bgbit = 1024, 2, 10
    halfBg = 1 << (Bgbit - 1)
    offset = (halfBg * sum(1 << (32 - (i + 1) * Bgbit) for i in range(l))) & 0xFFFFFFFF
    rs = np.random.RandomState(77)
    e = T.Engine(torus_bits=32, n=3, N=N, l=l, Bgbit=Bgbit, ks_t=0, lib_path=lib_path)
    try:
        coef = np.full((3, 2 * l, 2, N), -2 ** 31, np.int64)
        coef[1] = rs.randint(-2 ** 31, 2 ** 31, size=(2 * l, 2, N))           # ordinary key
        coef[2, :, :, ::2] = 2 ** 31 - 1                                        # alternating extremes
        gsw_lag = O.execute_reverse_int(N, coef.astype(np.int32).reshape(-1, N)).reshape(3, 2 * l, 2, N)
        g = e.gsw_from_fft(gsw_lag)
        allneg = np.full(2 * N, (-offset) & 0xFFFFFFFF, np.uint32).view(np.int32)  # every digit = -Bg/2
        acc = np.stack([allneg,
                        rs.randint(-2 ** 31, 2 ** 31, size=2 * N).astype(np.int32),
                        np.where(np.arange(2 * N) % 3 == 0, allneg, 0).astype(np.int32),
                        allneg, allneg, rs.randint(-2 ** 31, 2 ** 31, size=2 * N).astype(np.int32),
                        allneg, allneg, allneg]).reshape(9, 2, N)   # 9 samples: two workgroups / waves differ
        for idx in range(3):
            want = np.stack([O.extprod32(N, acc[b], gsw_lag[idx], l, Bgbit) for b in range(9)]).reshape(9, 2, N)
            assert np.array_equal(e.extern_mul(acc, g, idx), want), f"external product at extreme magnitude (key {idx})"
        ba = np.array([1, 5, N, 2 * N - 1, 7, 9, 11, 13, 1000], np.int32)
        want = np.stack([O.mux_rotate32(N, acc[b], gsw_lag[0], ba[b], l, Bgbit) for b in range(9)]).reshape(9, 2, N)
        assert np.array_equal(e.mux_rotate(acc, g, 0, ba), want), "CMux at extreme magnitude"
    finally:
        e.close()


def check_rounding_extremes64(lib_path, N=1024, l=4, Bgbit=9):
    """Torus64 rounding: the short sequence holds for |x| < 2^83, beyond it the kernel falls back to the
    reference's bit-field form (per wave).  All digits -Bg/2 against key coefficients -2^63 reach
    2*l*N*(Bg/2)*2^63 >= 2^84; mixed with ordinary samples and keys in one batch, plus values straddling the
    2^32 split (ties) that the short sequence treats specially."""
    rs = np.random.RandomState(78)
    offset = 0
    for i in range(l + 1):
        offset |= 1 << (63 - i * Bgbit)                      # poc:349-350 (with the rounding bit)
    e = T.Engine(torus_bits=64, n=3, N=N, l=l, Bgbit=Bgbit, ks_t=0, lib_path=lib_path)
    try:
        coef = np.full((3, 2 * l, 2, N), -2 ** 63, np.int64)
        coef[1] = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(2 * l, 2, N), dtype=np.int64)   # ordinary key
        coef[2, :, :, ::2] = 2 ** 63 - 1                                                  # alternating extremes
        gsw_lag = O.execute_reverse_torus64(N, coef.reshape(-1, N)).reshape(3, 2 * l, 2, N)
        g = e.gsw_from_fft(gsw_lag)
        allneg = np.full(2 * N, (-offset) & (2 ** 64 - 1), np.uint64).view(np.int64)       # every digit = -Bg/2
        rnd = lambda: rs.randint(-2 ** 63, 2 ** 63 - 1, size=2 * N, dtype=np.int64)
        acc = np.stack([allneg, rnd(), np.where(np.arange(2 * N) % 3 == 0, allneg, 0).astype(np.int64), allneg, rnd(),
                        allneg]).reshape(6, 2, N)   # more samples than one workgroup has waves: the choice is per wave
        for idx in range(3):
            want = np.stack([O.extprod64(N, acc[b], gsw_lag[idx], l, Bgbit) for b in range(6)]).reshape(6, 2, N)
            assert np.array_equal(e.extern_mul(acc, g, idx), want), f"Torus64 external product at extreme magnitude (key {idx})"
        # a key that is the constant polynomial 1 in one slot: the product is the digit polynomial itself, small
        # integers whose roundings sit exactly on / next to integers and the 2^32 split of the short sequence
        one = np.zeros((1, 2 * l, 2, N), np.int64)
        one[0, :, :, 0] = np.array([1, 2 ** 31, 2 ** 32, -2 ** 31, 2 ** 32 + 1, -(2 ** 32), 2 ** 33 - 1, -(2 ** 31) - 1][:2 * l]).reshape(2 * l, 1)
        lag1 = O.execute_reverse_torus64(N, one.reshape(-1, N)).reshape(1, 2 * l, 2, N)
        g1 = e.gsw_from_fft(lag1)
        want = np.stack([O.extprod64(N, acc[b], lag1[0], l, Bgbit) for b in range(6)]).reshape(6, 2, N)
        assert np.array_equal(e.extern_mul(acc, g1, 0), want), "Torus64 external product, products on the 2^31 / 2^32 boundaries"
    finally:
        e.close()


# ------------------------------------------------------------ circuit bootstrap
def check_circuit_bootstrap(lib_path, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, B, seed=61):
    """circuitPrivKS and the whole tfhe_CircuitBootstrapFFT pipeline (preKeySwitch, preModSwitch,
    Torus64 blind rotation, private key switch) against the oracle.  The bootstrapping key is a
    real TGSW key; preKS / privKS are synthetic (uniformly random tables): parity does not need
    them to decrypt, and generating a real privKS costs minutes of CPU."""
    rs = np.random.RandomState(seed)
    key0, key2 = O.keygen_binary(n0, SEED, 21), O.keygen_binary(N2, SEED, 23)
    bk = O.bk_create64(N2, key0, key2, l2, bg2, 2.0 ** -44, SEED, 3000)
    preks = O.fill32(101, N1 * t10 * (1 << bb10) * (n0 + 1)).reshape(N1, t10, 1 << bb10, n0 + 1)
    privks = O.fill32(202, 2 * (N2 + 1) * t21 * (1 << bb21) * 2 * N1).reshape(2, N2 + 1, t21, 1 << bb21, 2, N1)
    cb = T.CircuitBootstrap(n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, lib_path=lib_path)
    try:
        cb.load_preks(preks)
        cb.load_bk_fft(bk)
        cb.load_privks(privks)
        x64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(B, N2 + 1), dtype=np.int64)
        x64[0, :3] = [0, -1, 1 << (63 - t21 * bb21)]  # digit rounding boundary
        for u in (0, 1):
            want = np.stack([O.privks(privks[u], x64[b], N2, N1, t21, bb21) for b in range(B)]).reshape(B, 2, N1)
            assert np.array_equal(cb.privks(u, x64), want), f"circuitPrivKS u={u}"
        x = rs.randint(-2 ** 31, 2 ** 31, size=(B, N1 + 1)).astype(np.int32)
        want = np.stack([O.circuit_bootstrap(x[b], preks, bk, privks, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10,
                                             t21, bb21) for b in range(B)])
        assert np.array_equal(cb.circuit_bootstrap(x), want), "tfhe_CircuitBootstrapFFT"
        # the chain of BASELINE config 3, device-resident end to end: the TGSW32 outputs become the
        # selectors of a LUT evaluation (d = 2 items x ... bits taken from the same inputs)
        d = min(B, 11)
        table = O.fill32(77, (1 << max(0, d - (N1.bit_length() - 1))) * N1)
        tgsw, lwe = cb.circuit_bootstrap_lut(x[:d], d, table)
        assert np.array_equal(tgsw, want[:d]), "circuit bootstrap inside the chain"
        sel = O.execute_reverse_int(N1, want[:d].reshape(-1, N1)).reshape(d, 2 * l1, 2, N1)
        assert np.array_equal(lwe[0], O.lut_eval32(N1, sel, d, table, l1, bg1)), "circuit bootstrap -> LUT evaluation"
    finally:
        cb.close()


def check_privks_wide(lib_path, N1=1024, N2=2048, t21=10, bb21=3, counts=(300, 1031), pipeline_B=600, n0=4, l1=2, bg1=8, l2=4, bg2=9,
                      t10=6, bb10=2, seed=65, planes=(0, 1)):
    """circuitPrivKS (poc:667-698) on MANY samples per launch: the int64 instantiation of the matrix-core key switch with
    several 256-sample tiles and every wave of a tile live -- the launch shape BASELINE config 3's number comes from (1024
    circuit bootstraps = 2048 samples per plane).  For each count: 8 scattered rows (+ first / last row of every tile boundary)
    against the oracle, the first 3 rows against a count-3 launch of the same inputs, and every shorter count a prefix of the
    longest.  Then the whole pipeline at `pipeline_B` inputs (l1 * pipeline_B samples per plane, in the grouped output layout):
    scattered rows against the oracle and against a 3-input launch of the same rows."""
    rs = np.random.RandomState(seed)
    key0, key2 = O.keygen_binary(n0, SEED, 21), O.keygen_binary(N2, SEED, 23)
    bk = O.bk_create64(N2, key0, key2, l2, bg2, 2.0 ** -44, SEED, 3000)
    preks = O.fill32(101, N1 * t10 * (1 << bb10) * (n0 + 1)).reshape(N1, t10, 1 << bb10, n0 + 1)
    privks = O.fill32(202, 2 * (N2 + 1) * t21 * (1 << bb21) * 2 * N1).reshape(2, N2 + 1, t21, 1 << bb21, 2, N1)
    cb = T.CircuitBootstrap(n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, lib_path=lib_path)
    try:
        cb.load_preks(preks)
        cb.load_bk_fft(bk)
        cb.load_privks(privks)
        cmax = max(counts)
        x64 = np.frombuffer(rs.bytes(cmax * (N2 + 1) * 8), dtype=np.int64).reshape(cmax, N2 + 1).copy()
        x64[0, :3] = [0, -1, 1 << (63 - t21 * bb21)]  # digit rounding boundary
        for u in planes:
            small = cb.privks(u, x64[:3])
            for b in range(3):
                assert np.array_equal(small[b].ravel(), O.privks(privks[u], x64[b], N2, N1, t21, bb21)), f"u={u} row {b} of 3"
            full = None
            for count in sorted(counts, reverse=True):
                got = cb.privks(u, x64[:count])
                assert np.array_equal(got[:3], small), f"circuitPrivKS u={u} count={count}: first rows vs the 3-sample launch"
                rows = sorted(set([r for r in (0, 255, 256, 257, 511, 512, 767, 768, 1023, 1024, count - 1) if r < count]
                                  + list(rs.choice(count, min(8, count), replace=False))))
                for r in rows:
                    assert np.array_equal(got[r].ravel(), O.privks(privks[u], x64[r], N2, N1, t21, bb21)), f"circuitPrivKS u={u} count={count} row {r}"
                if full is None:
                    full = got
                else:
                    assert np.array_equal(got, full[:count]), f"u={u}: count={count} is not a prefix of the longest launch"
        B = pipeline_B
        if B < 3:
            return
        x = rs.randint(-2 ** 31, 2 ** 31, size=(B, N1 + 1)).astype(np.int32)
        got = cb.circuit_bootstrap(x)
        rows = sorted(set([0, 127, 128, 255, 256, B // 2, B - 1]) & set(range(B)))
        for r in rows[:5] + rows[-1:]:
            want = O.circuit_bootstrap(x[r], preks, bk, privks, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21)
            assert np.array_equal(got[r], want), f"tfhe_CircuitBootstrapFFT, input {r} of {B}"
        pick = [0, B // 2, B - 1]
        assert np.array_equal(cb.circuit_bootstrap(x[pick]), got[pick]), "the same inputs in a 3-input launch"
    finally:
        cb.close()


def check_pool(lib_path, devices, count, chunk, n=4, l=2, Bgbit=10, ks_t=8, ks_bb=2, seed=71):
    """tfhe_amd_pool_*: host arrays through a pool of `devices` (entries may repeat) with the pipeline chunk `chunk`, against the
    single-context engine on every row and the oracle on a few; keys handed over as device-layout bytes to a third context too"""
    N = 1024
    s = GateSetup(lib_path, N, n, l, Bgbit, ks_t, ks_bb, seed=SEED + seed)
    pool = T.Pool(devices, torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=ks_t, ks_basebit=ks_bb, lib_path=lib_path)
    other = T.Engine(torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=ks_t, ks_basebit=ks_bb, device=devices[-1], lib_path=lib_path)
    try:
        pool.load_keys(s.bk, s.ks)
        pool.set_chunk_rows(chunk)
        rs = np.random.RandomState(seed)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(count, n + 1)).astype(np.int32)
        mu = 1 << 29
        single = s.eng.bootstrap(mu, x)
        assert np.array_equal(pool.bootstrap(mu, x), single), "tfhe_amd_pool_bootstrap_host"
        for i in sorted(set([0, count // 2, count - 1])):
            assert np.array_equal(single[i], O.bootstrap32(N, s.bk, s.ks, mu, x[i], l, Bgbit, ks_t, ks_bb)), i
        u = pool.bootstrap_woks(mu, x)
        assert np.array_equal(u, s.eng.bootstrap_woks(mu, x)), "tfhe_amd_pool_bootstrap_woks_host"
        assert np.array_equal(pool.keyswitch(u), single), "tfhe_amd_pool_keyswitch_host"
        base, rem = divmod(count, len(devices))
        assert pool.last_split()[0] == [base + (1 if r < rem else 0) for r in range(len(devices))]
        bk_bytes, ks_bytes = np.empty(s.eng.gsw_packed_bytes(n), np.uint8), np.empty(s.eng.keyswitch_key_bytes(), np.uint8)
        s.eng.gsw_export_packed(s.gsw, T._np_ptr(bk_bytes))
        s.eng.keyswitch_key_export(T._np_ptr(ks_bytes))
        other.set_bootstrap_key(other.gsw_from_packed(T._np_ptr(bk_bytes), n))
        other.load_keyswitch_key_d(T._np_ptr(ks_bytes))
        assert np.array_equal(other.bootstrap(mu, x[:min(count, 70)]), single[:70]), "keys handed over as device-layout bytes"
    finally:
        other.close()
        pool.close()
        s.close()


# ------------------------------------------------- empty batches, state and parameter errors
def check_abi_edges(lib_path):
    """Empty batches are no-ops that touch nothing; calls made in the wrong state or with bad arguments fail
    with the documented status (include/tfhe_amd.h) instead of computing -- the reference asserts/aborts in
    the same situations (lwe_functions.cpp:480-481)."""
    import ctypes as C
    N, n = 1024, 3
    e = T.Engine(torus_bits=32, n=n, N=N, l=2, Bgbit=10, ks_t=8, ks_basebit=2, lib_path=lib_path)
    lib = e.lib
    try:
        sentinel = np.full((2, n + 1), 0x5A5A5A5A, np.int32)
        d_out, d_in = e.to_device(sentinel), e.to_device(np.zeros((2, N + 1), np.int32))
        # nothing loaded yet: bootstrap and key switch must refuse
        assert lib.tfhe_amd_bootstrap(e.ctx, d_out.ptr, 1 << 29, d_in.ptr, 1) == T.ERR_STATE
        assert lib.tfhe_amd_keyswitch(e.ctx, d_out.ptr, d_in.ptr, 1) == T.ERR_STATE
        assert b"key" in lib.tfhe_amd_last_error(e.ctx)
        lk, tk = O.keygen_binary(n, SEED, 1), O.keygen_binary(N, SEED, 2)
        bk = O.bk_create32(N, lk, tk, 2, 10, 2.0 ** -25, SEED, 1000)
        g = e.gsw_from_fft(bk)
        # a bootstrapping key of the wrong length is a parameter error
        g1 = e.gsw_from_fft(bk[:1])
        assert lib.tfhe_amd_set_bootstrap_key(e.ctx, g1) == T.ERR_PARAM
        e.set_bootstrap_key(g)
        e.load_keyswitch_key(O.ks_create32(tk, lk, 8, 2, 2.0 ** -15, SEED, 100000))
        # empty batches: OK, outputs untouched
        for call in (lambda: lib.tfhe_amd_bootstrap(e.ctx, d_out.ptr, 1 << 29, d_in.ptr, 0),
                     lambda: lib.tfhe_amd_bootstrap_woks(e.ctx, d_out.ptr, 1 << 29, d_in.ptr, 0),
                     lambda: lib.tfhe_amd_keyswitch(e.ctx, d_out.ptr, d_in.ptr, 0),
                     lambda: lib.tfhe_amd_bootstrap_streamed(e.ctx, d_out.ptr, 1 << 29, d_in.ptr, 0),
                     lambda: lib.tfhe_amd_extern_mul(e.ctx, d_in.ptr, g, 0, 0),
                     lambda: lib.tfhe_amd_blind_rotate(e.ctx, d_in.ptr, d_out.ptr, 0),
                     lambda: lib.tfhe_amd_modswitch(e.ctx, d_out.ptr, d_in.ptr, 0),
                     lambda: lib.tfhe_amd_ifft_int32(e.ctx, d_in.ptr, d_out.ptr, 0),
                     lambda: lib.tfhe_amd_fft_torus32(e.ctx, d_out.ptr, d_in.ptr, 0)):
            assert call() == T.OK
        e.sync()
        assert np.array_equal(d_out.download(np.int32, sentinel.shape), sentinel), "an empty batch wrote to its output"
        # bad arguments
        assert lib.tfhe_amd_bootstrap(e.ctx, None, 1 << 29, d_in.ptr, 1) == T.ERR_PARAM
        assert lib.tfhe_amd_bootstrap(e.ctx, d_out.ptr, 1 << 29, d_in.ptr, -1) == T.ERR_PARAM
        assert lib.tfhe_amd_extern_mul(e.ctx, d_in.ptr, g, n, 1) == T.ERR_PARAM  # TGSW index out of range
        assert lib.tfhe_amd_set_option(e.ctx, 99, 0) == T.ERR_PARAM
        # the bare transforms are out of place and their operands share an element type: ANY overlap of the two ranges is
        # refused (the same pointer, or an output that starts inside the input batch), disjoint neighbours are accepted
        d_lag = e.alloc(4 * N * 8)
        for fn in (lib.tfhe_amd_ifft_f64, lib.tfhe_amd_fft_f64):
            assert fn(e.ctx, d_lag.ptr, d_lag.ptr, 1) == T.ERR_PARAM
            assert fn(e.ctx, d_lag.ptr + N * 8, d_lag.ptr, 2) == T.ERR_PARAM
            assert fn(e.ctx, d_lag.ptr, d_lag.ptr + N * 8, 2) == T.ERR_PARAM
            assert b"overlap" in lib.tfhe_amd_last_error(e.ctx)
            assert fn(e.ctx, d_lag.ptr + 2 * N * 8, d_lag.ptr, 2) == T.OK
        e.sync()
        # a batch of one still works after all of that (state intact)
        x = O.lwe_encrypt32(1 << 29, 2.0 ** -15, lk, O.rng(SEED, 77))
        got = e.bootstrap(1 << 29, x[None, :])
        assert np.array_equal(got[0], O.bootstrap32(N, bk, O.ks_create32(tk, lk, 8, 2, 2.0 ** -15, SEED, 100000), 1 << 29, x, 2, 10, 8, 2))
    finally:
        e.close()


# --------------------------------------------------------------- key switch only
def check_keyswitch_shapes(lib_path, N, n_out, ks_t, ks_bb, B, seed=51):
    """lweKeySwitch / preKeySwitch on a synthetic (uniformly random) key: exercises the
    matrix-core kernel's column blocks / K padding for the real output sizes, and the gather kernel beside it."""
    import os
    rs = np.random.RandomState(seed)
    ks = rs.randint(-2 ** 31, 2 ** 31, size=(N, ks_t, 1 << ks_bb, n_out + 1)).astype(np.int32)
    x = rs.randint(-2 ** 31, 2 ** 31, size=(B, N + 1)).astype(np.int32)
    x[0, :4] = [0, -1, 1 << 31 - ks_t * ks_bb, -(1 << 31 - ks_t * ks_bb)]  # rounding-boundary digits
    want = np.stack([O.keyswitch32(ks, x[b], N, n_out, ks_t, ks_bb) for b in range(B)])
    e = T.Engine(torus_bits=32, n=n_out, N=N, l=2, Bgbit=10, ks_t=ks_t, ks_basebit=ks_bb, lib_path=lib_path)
    try:
        e.load_keyswitch_key(ks)
        for force_gather in (0, 1):
            e.set_option(T.OPT_KS_GATHER, force_gather)
            assert np.array_equal(e.keyswitch(x), want), f"key switch (gather={force_gather})"
        e.set_option(T.OPT_KS_GATHER, 0)
        assert np.array_equal(e.keyswitch(x[:1]), want[:1]), "key switch, one sample"
    finally:
        e.close()
