"""Ring degrees other than 1024 / 2048.  The reference's FFT plugin is generic in N: new_fft_table / new_ifft_table /
FFT_Processor_Spqlios(N) accept every power of two >= 16 (CB/spqlios/spqlios-fft-impl.cpp:157-160,400-403,
fft_processor_spqlios.cpp:18-25); so is the library-form bootstrap (parameters are run-time values, lwe_functions.cpp:399-446).
The engine serves them through tfhe_kernels_generic.h (a team of work-items per polynomial instead of a wave).

Pinned by tests/golden/ref_ringdeg.npz: outputs of the COMPILED REFERENCE (`oracle/_ref/ref_driver`, which constructs
FFT_Processor_Spqlios(N)) for N in {16, 64, 512, 4096, 8192}, table SHA-256s included, and gate bootstraps composed from the
reference's object code at N = 512, 4096 and 16 (generator: tests/golden/make_golden_ringdeg.py).

  not gpu:  the oracle against those goldens; the kernels on the CPU emulator against the oracle and the goldens
  gpu:      the HIP library against the goldens (no oracle in the loop) and against the oracle (parity_checks)"""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

import oracle_py as O
import parity_checks as P

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_ringdeg.npz"))
META = json.load(open(os.path.join(HERE, "golden", "ref_ringdeg.json")))
RING_DEGREES = tuple(META["ring_degrees"])
HASH_ONLY = tuple(META["hash_only"])
GATE_SETS = [tuple(g) for g in META["gate_sets"]]
T = importlib.import_module("experimental-tfhe_amd")


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def inputs(N):
    """the generator's inputs of ring degree N: stored, or (hash-only degrees) redrawn from the recorded seed"""
    if N in HASH_ONLY:
        import golden.make_golden_ringdeg as M
        return M.draw(np.random.RandomState(META["numpy_seed"] + N), N, 1)
    return G[f"a32_{N}"], G[f"dig_{N}"], G[f"a64_{N}"], G[f"raw_in_{N}"]


def expect(N, key, got, doubles):
    """`got` == the compiled reference's output `key` at ring degree N (stored array, or its SHA-256)"""
    if N in HASH_ONLY:
        if doubles:  # a sign of zero is the one tolerated difference: normalise before hashing (the reference's outputs hold no -0.0 here)
            got = np.ascontiguousarray(got, np.float64).copy()
            got[got == 0.0] = 0.0
        assert sha(got) == META["output_sha256"][str(N)][key], (N, key)
    elif doubles:
        assert P.same_doubles(got, G[f"{key}_{N}"]), (N, key)
    else:
        assert np.array_equal(got, G[f"{key}_{N}"]), (N, key)


class OracleFace:
    """the oracle behind the method names of the engine binding, so one golden walk serves both"""

    def __init__(self, N):
        self.N = N

    def tables(self):
        return O.table_arrays(self.N)

    def ifft_int32(self, a):
        return O.execute_reverse_int(self.N, a)

    def ifft_torus64(self, a):
        return O.execute_reverse_torus64(self.N, a)

    def fft_torus32(self, a):
        return O.execute_direct_torus32(self.N, a)

    def fft_torus64(self, a):
        return O.execute_direct_torus64(self.N, a)

    def ifft_f64(self, a):
        return O.ifft(self.N, a)

    def fft_f64(self, a):
        return O.fft(self.N, a)

    def lagrange_addmul(self, res, a, b):
        return O.lagrange_addmul(self.N, res, a, b)


def walk_goldens(face, N):
    f, r = face.tables()
    assert sha(f) == META["table_sha256"][f"fft_trig_{N}"] and sha(r) == META["table_sha256"][f"ifft_trig_{N}"]
    a32, dig, a64, raw = inputs(N)
    rev_a32, rev_dig, rev_t64 = face.ifft_int32(a32), face.ifft_int32(dig), face.ifft_torus64(a64)
    expect(N, "rev_int_a32", rev_a32, True)
    expect(N, "rev_int_dig", rev_dig, True)
    expect(N, "rev_t64", rev_t64, True)
    z = np.zeros((a32.shape[0], N))
    # (inputs of the next stage are the REFERENCE's outputs where stored, so one failure does not cascade)
    ref_or = lambda key, mine: mine if N in HASH_ONLY else G[f"{key}_{N}"]
    am32 = face.lagrange_addmul(z, ref_or("rev_int_dig", rev_dig), ref_or("rev_int_a32", rev_a32))
    am64 = face.lagrange_addmul(z, ref_or("rev_int_dig", rev_dig), ref_or("rev_t64", rev_t64))
    expect(N, "addmul32", am32, True)
    expect(N, "addmul64", am64, True)
    expect(N, "dir_t32", face.fft_torus32(ref_or("addmul32", am32)), False)
    expect(N, "dir_t64", face.fft_torus64(ref_or("addmul64", am64)), False)
    expect(N, "raw_ifft", face.ifft_f64(raw), True)
    expect(N, "raw_fft", face.fft_f64(raw), True)


def gate_keys(N, n, l, Bgbit, t, bb):
    import golden.make_golden_ringdeg as M
    return M.gate_keys(N, n, l, Bgbit, t, bb)


# ------------------------------------------------------------------ the oracle against the compiled reference
@pytest.mark.parametrize("N", RING_DEGREES)
def test_oracle_vs_reference_golden(N):
    walk_goldens(OracleFace(N), N)


@pytest.mark.parametrize("gs", GATE_SETS, ids=lambda g: f"N{g[0]}")
def test_oracle_gate_bootstrap_vs_reference_object_code(gs):
    N, n, l, Bgbit, t, bb, count = gs
    lk, bk, ks = gate_keys(N, n, l, Bgbit, t, bb)
    x = G[f"boot32_x_{N}"]
    got = np.stack([O.bootstrap32(N, bk, ks, META["mu"], x[i], l, Bgbit, t, bb) for i in range(count)])
    assert np.array_equal(got, G[f"boot32_out_{N}"])
    if N == 512:  # the one set whose key switch (4 x 2 bits) is precise enough to decrypt: the others are parity shapes only
        for i in range(count // 2):
            assert (O.lwe_phase32(got[i], lk) > 0) == bool(i % 2)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/ref_driver not present")
@pytest.mark.parametrize("N", [32, 128, 256, 16384])
def test_oracle_vs_live_reference_other_degrees(N):
    """ring degrees without stored vectors, against the reference run here (FFT_Processor_Spqlios(N))"""
    rs = np.random.RandomState(N)
    a32 = rs.randint(-2 ** 31, 2 ** 31, size=(2, N)).astype(np.int32)
    a64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(2, N), dtype=np.int64)
    t = O.ref("tables", b"", np.float64, N)
    f, r = O.table_arrays(N)
    assert np.array_equal(bits(t[:2 * N - 8]), bits(f)) and np.array_equal(bits(t[2 * N - 8:]), bits(r))
    rev = O.execute_reverse_int(N, a32)
    assert np.array_equal(bits(O.ref("rev_int", a32, np.float64, N).reshape(2, N)), bits(rev))
    rev64 = O.execute_reverse_torus64(N, a64)
    assert np.array_equal(bits(O.ref("rev_t64", a64, np.float64, N).reshape(2, N)), bits(rev64))
    lag = O.lagrange_addmul(N, np.zeros((2, N)), O.execute_reverse_int(N, a32 >> 22), rev)
    assert np.array_equal(O.ref("dir_t32", lag, np.int32, N).reshape(2, N), O.execute_direct_torus32(N, lag))
    lag64 = O.lagrange_addmul(N, np.zeros((2, N)), O.execute_reverse_int(N, a32 >> 22), rev64)
    assert np.array_equal(O.ref("dir_t64", lag64, np.int64, N).reshape(2, N), O.execute_direct_torus64(N, lag64))


# ------------------------------------------------------------------ the kernels on the CPU emulator
@pytest.mark.parametrize("N", RING_DEGREES)
def test_engine_vs_reference_golden_emu(emu_lib, N):
    e = T.Engine(torus_bits=32, n=1, N=N, l=2, Bgbit=10, ks_t=0, lib_path=emu_lib)
    try:
        walk_goldens(e, N)
    finally:
        e.close()


@pytest.mark.parametrize("N", [16, 32, 512, 4096, 32768, 1 << 20])
def test_fft_plugin_emu(emu_lib, N):
    """every execute_*, AddMul and the C core against the oracle, ragged batches; 32768 and the largest degree served, 2^20:
    transform buffers in global scratch"""
    P.check_fft_plugin(emu_lib, N, count=1 if N > 32768 else (3 if N >= 4096 else 67))


@pytest.mark.parametrize("N,n,l,Bgbit,ks_t,ks_bb,B", [(512, 4, 2, 10, 4, 2, 3), (4096, 3, 2, 10, 3, 2, 2), (16, 5, 3, 6, 5, 2, 5),
                                                      (8192, 2, 2, 10, 2, 2, 2)])
def test_gate_path_emu(emu_lib, N, n, l, Bgbit, ks_t, ks_bb, B):
    """key conversion, external product, CMux, blind rotation (zero rotations, both halves of [0, 2N)), extraction, mod switch,
    bootstrap with and without key switch, the streamed schedule -- N = 8192: every work area in global scratch"""
    P.check_gate_path(emu_lib, N=N, n=n, l=l, Bgbit=Bgbit, ks_t=ks_t, ks_bb=ks_bb, B=B)


@pytest.mark.parametrize("gs", GATE_SETS, ids=lambda g: f"N{g[0]}")
def test_gate_bootstrap_vs_reference_object_code_emu(emu_lib, gs):
    check_boot32_golden(emu_lib, gs)


def check_boot32_golden(lib, gs):
    N, n, l, Bgbit, t, bb, count = gs
    lk, bk, ks = gate_keys(N, n, l, Bgbit, t, bb)
    e = T.Engine(torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=lib)
    try:
        e.set_bootstrap_key(e.gsw_from_fft(bk))
        e.load_keyswitch_key(ks)
        assert np.array_equal(e.bootstrap(META["mu"], G[f"boot32_x_{N}"]), G[f"boot32_out_{N}"])
        assert np.array_equal(e.bootstrap(META["mu"], G[f"boot32_x_{N}"], streamed=True), G[f"boot32_out_{N}"])
    finally:
        e.close()


@pytest.mark.parametrize("N,l,Bgbit,B", [(512, 4, 9, 3), (4096, 3, 10, 2)])
def test_torus64_path_emu(emu_lib, N, l, Bgbit, B):
    """4096 / Torus64: the three work areas are exactly the 160 KB of LDS"""
    P.check_torus64_path(emu_lib, N=N, n=3, l=l, Bgbit=Bgbit, B=B)


def test_consumers_emu(emu_lib):
    """the callers either side of the path at another ring degree: exact backend, CMux on data, LUT evaluation, circuit bootstrap"""
    P.check_exact_extprod(emu_lib, 32, 512, 2, 10, B=2, fft_bound=4)
    P.check_exact_extprod(emu_lib, 64, 512, 3, 9, B=2, fft_bound=2 ** 34)
    P.check_cmux_data(emu_lib, N=512, B=5)
    P.check_lut_eval(emu_lib, N=512, d=11, B=2)
    P.check_lut_eval(emu_lib, N=64, d=3, B=2, decrypt_tol=None)
    P.check_circuit_bootstrap(emu_lib, n0=2, N1=512, N2=4096, l1=2, bg1=8, l2=3, bg2=9, t10=3, bb10=2, t21=2, bb21=3, B=2)


def test_ring_degree_validation_emu(emu_lib):
    """what new_fft_table's require() rejects is rejected (spqlios-fft-impl.cpp:158-159): below 16, not a power of two"""
    for N in (8, 24, 1000, 3 << 10, 1 << 21):
        with pytest.raises(T.TfheAmdError):
            T.Engine(torus_bits=32, n=1, N=N, l=2, Bgbit=10, ks_t=0, lib_path=emu_lib)
        with pytest.raises(T.TfheAmdError):
            T.build_tables(N, lib_path=emu_lib)
    e = T.Engine(torus_bits=64, n=1, N=512, l=2, Bgbit=10, ks_t=0, lib_path=emu_lib)
    try:  # the Real96 transforms stay at the reference's own sizes
        with pytest.raises(T.TfheAmdError):
            e.hp_ifft(np.zeros((1, 512), np.int64))
    finally:
        e.close()


# ------------------------------------------------------------------ the HIP library
@pytest.mark.gpu
@pytest.mark.parametrize("N", RING_DEGREES)
def test_engine_vs_reference_golden_gpu(gpu_lib, N):
    """no oracle in the loop: GPU outputs == outputs of the reference's own object code at this ring degree"""
    e = T.Engine(torus_bits=32, n=1, N=N, l=2, Bgbit=10, ks_t=0, lib_path=gpu_lib)
    try:
        walk_goldens(e, N)
    finally:
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [16, 64, 512, 4096, 8192, 32768, 1 << 20])
def test_fft_plugin_gpu(gpu_lib, N):
    P.check_fft_plugin(gpu_lib, N, count=2 if N > 32768 else (5 if N >= 4096 else 131))


@pytest.mark.gpu
def test_gate_path_n65536_gpu(gpu_lib):
    """far beyond what the LDS holds: accumulator, digits and Fourier accumulator (3.5 MB per ciphertext) in global scratch"""
    P.check_gate_path(gpu_lib, N=1 << 16, n=2, l=2, Bgbit=10, ks_t=2, ks_bb=2, B=3, check_export=False)


@pytest.mark.gpu
@pytest.mark.parametrize("N,n,l,Bgbit,ks_t,ks_bb,B", [(512, 9, 2, 10, 8, 2, 19), (4096, 5, 2, 10, 4, 2, 5), (16, 7, 3, 6, 5, 2, 9),
                                                      (8192, 3, 2, 10, 2, 2, 3)])
def test_gate_path_gpu(gpu_lib, N, n, l, Bgbit, ks_t, ks_bb, B):
    P.check_gate_path(gpu_lib, N=N, n=n, l=l, Bgbit=Bgbit, ks_t=ks_t, ks_bb=ks_bb, B=B)


@pytest.mark.gpu
@pytest.mark.parametrize("gs", GATE_SETS, ids=lambda g: f"N{g[0]}")
def test_gate_bootstrap_vs_reference_object_code_gpu(gpu_lib, gs):
    check_boot32_golden(gpu_lib, gs)


@pytest.mark.gpu
@pytest.mark.parametrize("N,l,Bgbit,B", [(512, 4, 9, 7), (4096, 3, 10, 3)])
def test_torus64_path_gpu(gpu_lib, N, l, Bgbit, B):
    P.check_torus64_path(gpu_lib, N=N, n=4, l=l, Bgbit=Bgbit, B=B)


@pytest.mark.gpu
def test_consumers_gpu(gpu_lib):
    P.check_exact_extprod(gpu_lib, 32, 512, 2, 10, B=3, fft_bound=4)
    P.check_exact_extprod(gpu_lib, 64, 512, 3, 9, B=2, fft_bound=2 ** 34)
    P.check_cmux_data(gpu_lib, N=512, B=11)
    P.check_lut_eval(gpu_lib, N=512, d=11, B=3)
    P.check_circuit_bootstrap(gpu_lib, n0=4, N1=512, N2=4096, l1=2, bg1=8, l2=3, bg2=9, t10=3, bb10=2, t21=4, bb21=3, B=3)


@pytest.mark.gpu
def test_wide_batch_n512_gpu(gpu_lib):
    """a batch larger than the persistent grid (workgroups walk several ciphertexts each) at N = 512: first, last and a few
    rows against the oracle, every row against a second run in two halves"""
    N, n, l, Bgbit, t, bb, B = 512, 6, 2, 10, 4, 2, 5000
    s = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    try:
        rs = np.random.RandomState(512)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1), dtype=np.int64).astype(np.int32)
        got = s.eng.bootstrap(1 << 29, x)
        for i in (0, 1, 2047, 2048, 4097, B - 1):
            assert np.array_equal(got[i], O.bootstrap32(N, s.bk, s.ks, 1 << 29, x[i], l, Bgbit, t, bb)), i
        assert np.array_equal(np.concatenate([s.eng.bootstrap(1 << 29, x[:2500]), s.eng.bootstrap(1 << 29, x[2500:])]), got)
    finally:
        s.close()
