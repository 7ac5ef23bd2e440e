"""The oracle (oracle/tfhe_oracle.c) against golden vectors produced by the COMPILED
REFERENCE (tests/golden/ref_vectors.npz, generator: tests/golden/make_golden.py), and --
where oracle/_ref/ref_driver exists -- against the reference live on fresh inputs."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_py as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ref_vectors.npz"))
META = json.load(open(os.path.join(HERE, "golden", "ref_vectors.json")))
n0, n1, n2, bg1, l1, bg2, l2, t10, bb10, t21, bb21 = [int(v) for v in G["poc_params"]]


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.mark.parametrize("N", [1024, 2048])
def test_twiddle_tables_sha256(N):
    # SURVEY App. A.1 hashes == hashes of the compiled reference's tables == oracle tables
    f, r = O.table_arrays(N)
    assert hashlib.sha256(f.tobytes()).hexdigest() == META["table_sha256"][f"fft_trig_{N}"]
    assert hashlib.sha256(r.tobytes()).hexdigest() == META["table_sha256"][f"ifft_trig_{N}"]
    survey = {1024: ("42044d755ef66b39097fd80762663ebbc8285b2a482956ee2dc74e75ecfd87c0",
                     "388a3fd58ba229a7fddd66645e30c12587c5ebe2a2eee833c119b211420d28a5"),
              2048: ("ee02048210c18afeac62d64ad2965995888d262b35ee2a43c221721e59124355",
                     "581fb6f7c7d54f2172c800428f13e553793c0af1ea2f2e064d589202748a7002")}[N]
    assert (META["table_sha256"][f"fft_trig_{N}"], META["table_sha256"][f"ifft_trig_{N}"]) == survey


@pytest.mark.parametrize("N", [1024, 2048])
def test_fft_plugin_golden(N):
    assert np.array_equal(bits(O.execute_reverse_int(N, G[f"a32_{N}"])), bits(G[f"rev_int_a32_{N}"]))
    assert np.array_equal(bits(O.execute_reverse_int(N, G[f"dig_{N}"])), bits(G[f"rev_int_dig_{N}"]))
    assert np.array_equal(bits(O.execute_reverse_torus64(N, G[f"a64_{N}"])), bits(G[f"rev_t64_{N}"]))
    z = np.zeros_like(G[f"addmul32_{N}"])
    assert np.array_equal(bits(O.lagrange_addmul(N, z, G[f"rev_int_dig_{N}"], G[f"rev_int_a32_{N}"])),
                          bits(G[f"addmul32_{N}"]))
    assert np.array_equal(bits(O.lagrange_addmul(N, z, G[f"rev_int_dig_{N}"], G[f"rev_t64_{N}"])),
                          bits(G[f"addmul64_{N}"]))
    assert np.array_equal(
        bits(O.lagrange_addmul(N, G[f"addmul32_{N}"], G[f"rev_int_dig_{N}"][::-1].copy(), G[f"rev_int_a32_{N}"])),
        bits(G[f"addmul32b_{N}"]))
    assert np.array_equal(O.execute_direct_torus32(N, G[f"addmul32_{N}"]), G[f"dir_t32_{N}"])
    assert np.array_equal(O.execute_direct_torus64(N, G[f"addmul64_{N}"]), G[f"dir_t64_{N}"])
    assert np.array_equal(bits(O.ifft(N, G[f"raw_in_{N}"])), bits(G[f"raw_ifft_{N}"]))
    assert np.array_equal(bits(O.fft(N, G[f"raw_in_{N}"])), bits(G[f"raw_fft_{N}"]))


def test_decomp64_golden():
    got = np.stack([O.decomp64(x, l2, bg2) for x in G["decomp64_in"]])
    assert np.array_equal(got, G["decomp64_out"])


def test_premodswitch_golden():
    got = np.stack([O.pre_modswitch(x, n2) for x in G["premodswitch_in"]])
    assert np.array_equal(got, G["premodswitch_out"])


def test_prekeyswitch_golden():
    tab = O.fill32(META["preks_seed"], n1 * t10 * (1 << bb10) * (n0 + 1))
    assert np.array_equal(tab[:4096], O.fill32_numpy(META["preks_seed"], 4096))  # C filler == numpy spec
    got = np.stack([O.keyswitch32(tab, x, n1, n0, t10, bb10) for x in G["preks_in"]])
    assert np.array_equal(got, G["preks_out"])


def test_privks_golden():
    # 1.34 GB synthetic table regenerated from the seed (same table for u = 0 and u = 1:
    # ref_driver builds one plane and aliases it)
    tab = O.fill32(META["privks_seed"], (n2 + 1) * t21 * (1 << bb21) * 2 * n1)
    got = O.privks(tab, G["privks_in"][0], n2, n1, t21, bb21).reshape(2, n1)
    assert np.array_equal(got, G["privks_out_u0"])
    assert np.array_equal(got, G["privks_out_u1"])


def test_karatsuba_golden():
    assert np.array_equal(O.negacyclic_mul32(G["karat32_int"][0], G["karat32_torus"][0]), G["karat32_out"][0])
    assert np.array_equal(O.negacyclic_mul64(G["karat64_int"][0], G["karat64_torus"][0]), G["karat64_out"][0])


def test_poc_blind_rotation_composition_golden():
    """500 CMux steps of the PoC loop (with its quirks, on the subset where it is defined)
    reproduce the reference binary bit for bit: pins decomposition -> ifft -> MAC order (p outer,
    q inner) -> fft -> accumulate -> extraction (+mu/2), i.e. everything the library-form blind
    rotation shares with it."""
    got = O.cb_bootstrap_woks64_poc_quirks(n2, int(G["cbwoks_mu"]), G["cbwoks_abar"], G["cbwoks_bk0"], l2, bg2)
    assert np.array_equal(got, G["cbwoks_out"])


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/ref_driver not built here")
@pytest.mark.parametrize("N", [1024, 2048])
def test_live_reference_fresh_inputs(N):
    rs = np.random.RandomState(N + 7)
    cnt = 8
    a = rs.randint(-2 ** 31, 2 ** 31, size=(cnt, N)).astype(np.int32)
    d = rs.randint(-512, 512, size=(cnt, N)).astype(np.int32)
    a64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(cnt, N), dtype=np.int64)
    la, ld, l64 = O.execute_reverse_int(N, a), O.execute_reverse_int(N, d), O.execute_reverse_torus64(N, a64)
    assert np.array_equal(bits(la), bits(O.ref("rev_int", a, np.float64, N).reshape(cnt, N)))
    assert np.array_equal(bits(l64), bits(O.ref("rev_t64", a64, np.float64, N).reshape(cnt, N)))
    acc = O.lagrange_addmul(N, np.zeros((cnt, N)), ld, la)
    assert np.array_equal(bits(acc), bits(O.ref("addmul", np.concatenate([np.zeros((cnt, N)), ld, la], axis=1),
                                                 np.float64, N).reshape(cnt, N)))
    assert np.array_equal(O.execute_direct_torus32(N, acc), O.ref("dir_t32", acc, np.int32, N).reshape(cnt, N))
    # beyond the int64 range int32_t(int64_t(x)) is undefined in C; the compiled reference yields 0 there
    big = acc * 2.0 ** 40
    assert np.array_equal(O.execute_direct_torus32(N, big), O.ref("dir_t32", big, np.int32, N).reshape(cnt, N))
    acc64 = O.lagrange_addmul(N, np.zeros((cnt, N)), ld, l64)
    assert np.array_equal(O.execute_direct_torus64(N, acc64), O.ref("dir_t64", acc64, np.int64, N).reshape(cnt, N))


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/ref_driver not built here")
def test_oracle_gate_bootstrap_matches_reference_object_code():
    """tfhe_bootstrap_FFT end to end: the oracle against the composition of the reference's OWN transform /
    multiply-accumulate object code (ref_driver boot32: fftp1024.execute_reverse_int, the AddMul assembly,
    fftp1024.execute_direct_torus32 around the integer glue of lwe_functions.cpp:136-171,337-446), on real
    keys at a short n (the per-step arithmetic does not depend on n), incl. the slice argument the GPU
    full-batch test uses."""
    N, n, l, Bgbit, t, bb = 1024, 6, 2, 10, 8, 2
    seed = 0x5446484500000001
    lk, tk = O.keygen_binary(n, seed, 1), O.keygen_binary(N, seed, 2)
    bk = O.bk_create32(N, lk, tk, l, Bgbit, 2.0 ** -25, seed, 1000)
    ks = O.ks_create32(tk, lk, t, bb, 2.0 ** -15, seed, 100000)
    mu = 1 << 29
    rs = np.random.RandomState(5)
    x = np.concatenate([np.stack([O.lwe_encrypt32(mu if i % 2 else -mu, 2.0 ** -15, lk, O.rng(seed, 70 + i)) for i in range(3)]),
                        rs.randint(-2 ** 31, 2 ** 31, size=(3, n + 1)).astype(np.int32)])
    x[4, 2] = 0  # a rotation by zero: the skipped step (lwe_functions.cpp:348-350)
    want = np.stack([O.bootstrap32(N, bk, ks, mu, x[i], l, Bgbit, t, bb) for i in range(6)])
    blob = (np.array([mu, 0], np.int32).tobytes() + np.ascontiguousarray(bk, np.float64).tobytes()
            + np.ascontiguousarray(ks, np.int32).tobytes() + x.tobytes())
    got = O.ref("boot32", blob, np.int32, n, l, Bgbit, t, bb, 6).reshape(6, n + 1)
    assert np.array_equal(got, want)
    part = O.ref("boot32", blob, np.int32, n, l, Bgbit, t, bb, 2, 3).reshape(2, n + 1)   # rows 3..4
    assert np.array_equal(part, want[3:5])
