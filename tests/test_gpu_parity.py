"""GPU parity (-m gpu): the shipped HIP library, through the C ABI, against (a) the oracle on
the same seeded inputs, (b) the golden vectors of the compiled reference, (c) size-independent
properties at BASELINE.json's full sizes.  Bit-exact everywhere (Torus32/Torus64 integers;
Lagrange doubles on their bit patterns)."""
import importlib
import json
import os

import numpy as np
import pytest

import oracle_py as O
import parity_checks as P

pytestmark = pytest.mark.gpu
T = importlib.import_module("experimental-tfhe_amd")
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "ref_vectors.npz")), json.load(
        open(os.path.join(HERE, "golden", "ref_vectors.json")))


@pytest.mark.parametrize("N", [1024, 2048])
def test_fft_plugin_vs_oracle(gpu_lib, N):
    P.check_fft_plugin(gpu_lib, N, count=37)  # ragged against 4 waves per workgroup


@pytest.mark.parametrize("N", [1024, 2048])
def test_fft_plugin_vs_reference_golden(gpu_lib, golden, N):
    """no oracle in the loop: GPU outputs == outputs of the reference's own object code"""
    G, meta = golden
    e = T.Engine(torus_bits=32, n=1, N=N, l=2, Bgbit=10, ks_t=0, lib_path=gpu_lib)
    try:
        import hashlib
        f, r = e.tables()
        assert hashlib.sha256(f.tobytes()).hexdigest() == meta["table_sha256"][f"fft_trig_{N}"]
        assert hashlib.sha256(r.tobytes()).hexdigest() == meta["table_sha256"][f"ifft_trig_{N}"]
        assert P.same_doubles(e.ifft_int32(G[f"a32_{N}"]), G[f"rev_int_a32_{N}"])
        assert P.same_doubles(e.ifft_int32(G[f"dig_{N}"]), G[f"rev_int_dig_{N}"])
        assert P.same_doubles(e.ifft_torus64(G[f"a64_{N}"]), G[f"rev_t64_{N}"])
        z = np.zeros_like(G[f"addmul32_{N}"])
        assert P.same_doubles(e.lagrange_addmul(z, G[f"rev_int_dig_{N}"], G[f"rev_int_a32_{N}"]), G[f"addmul32_{N}"])
        assert P.same_doubles(e.lagrange_addmul(G[f"addmul32_{N}"], G[f"rev_int_dig_{N}"][::-1].copy(),
                                                G[f"rev_int_a32_{N}"]), G[f"addmul32b_{N}"])
        assert np.array_equal(e.fft_torus32(G[f"addmul32_{N}"]), G[f"dir_t32_{N}"])
        assert np.array_equal(e.fft_torus64(G[f"addmul64_{N}"]), G[f"dir_t64_{N}"])
    finally:
        e.close()


def test_gate_path_small(gpu_lib):
    P.check_gate_path(gpu_lib, N=1024, n=9, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=19)


def test_gate_path_other_gadgets(gpu_lib):
    P.check_gate_path(gpu_lib, N=1024, n=5, l=3, Bgbit=7, ks_t=16, ks_bb=1, B=9, seed=5)
    P.check_gate_path(gpu_lib, N=1024, n=4, l=1, Bgbit=12, ks_t=5, ks_bb=3, B=1, seed=6)
    P.check_gate_path(gpu_lib, N=2048, n=5, l=2, Bgbit=9, ks_t=4, ks_bb=3, B=7, seed=7)


def test_gate_path_runtime_gadget(gpu_lib):
    """the instantiations that read the gadget at run time: l = 2 with another Bgbit, l = 4 (two digit pairs
    per polynomial), and the circuit bootstrap's output gadget (l = 2, Bgbit = 8: compile-time too)"""
    P.check_gate_path(gpu_lib, N=1024, n=6, l=2, Bgbit=9, ks_t=8, ks_bb=2, B=9, check_export=False, seed=12)
    P.check_gate_path(gpu_lib, N=1024, n=5, l=4, Bgbit=6, ks_t=8, ks_bb=2, B=5, check_export=False, seed=13)
    P.check_gate_path(gpu_lib, N=1024, n=7, l=2, Bgbit=8, ks_t=8, ks_bb=2, B=9, check_export=False, seed=14)


def test_rounding_extremes(gpu_lib):
    """worst-case magnitude 2^52 of the external product: exact fallback of the Torus32 rounding"""
    P.check_rounding_extremes(gpu_lib)


@pytest.mark.parametrize("N", [1024, 2048])
def test_rounding_extremes_torus64(gpu_lib, N):
    """Torus64 rounding: short sequence (|x| < 2^83), guard and exact fallback at magnitudes up to 2^85"""
    P.check_rounding_extremes64(gpu_lib, N=N, l=4, Bgbit=9)


def test_gate_path_full_parameters(gpu_lib):
    """BASELINE config 1/2 parameter set: n=630, N=1024, k=1, l=2, Bgbit=10, ks 8x2"""
    P.check_gate_path(gpu_lib, N=1024, n=630, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=6, check_export=True)


def test_keyswitch_second_config(gpu_lib):
    """the reference's other gate key-switch setting: length 16, base 2 (params-gb.html:130-131)"""
    P.check_gate_path(gpu_lib, N=1024, n=12, l=2, Bgbit=10, ks_t=16, ks_bb=1, B=5, seed=8, check_export=False)


@pytest.mark.parametrize("n_out,t,bb,B", [(630, 8, 2, 67), (500, 6, 2, 33), (630, 16, 1, 16), (700, 4, 3, 5),
                                          (500, 10, 3, 9), (300, 5, 3, 261), (33, 31, 1, 3)])
def test_keyswitch_real_shapes(gpu_lib, n_out, t, bb, B):
    """lweKeySwitch at the gate sizes, preKeySwitch at the PoC sizes (both through the matrix-core
    and the gather kernel), base-8 shapes with 1, 2 and 3 K-steps per input coefficient, a batch that spans two
    256-sample tiles, and the longest base-2 decomposition"""
    P.check_keyswitch_shapes(gpu_lib, 1024, n_out, t, bb, B)


@pytest.mark.parametrize("N,n,l,Bgbit,B", [(2048, 6, 4, 9, 7), (1024, 5, 3, 10, 9)])
def test_torus64_path(gpu_lib, N, n, l, Bgbit, B):
    P.check_torus64_path(gpu_lib, N=N, n=n, l=l, Bgbit=Bgbit, B=B)


def test_circuit_bootstrap_blind_rotation_full(gpu_lib):
    """PoC parameter block (poc:70-85): n0=500, N2=2048, l2=4, Bgbit2=9, Torus64"""
    P.check_torus64_path(gpu_lib, N=2048, n=500, l=4, Bgbit=9, B=4, seed=41)


@pytest.mark.parametrize("B,l,Bgbit,n", [(1, 4, 9, 9), (7, 4, 9, 6), (6, 3, 7, 8), (5, 1, 12, 8), (2, 8, 4, 3)])
def test_torus64_n2048_gadget_shapes(gpu_lib, B, l, Bgbit, n):
    """gadget lengths 1..8 on Torus64 / N = 2048: digits in the high word, the low word and across both; skipped steps"""
    P.check_torus64_path(gpu_lib, N=2048, n=n, l=l, Bgbit=Bgbit, B=B, seed=200 + B)


def test_cmux_on_data(gpu_lib):
    """vertical-packing building block on TGSW selectors with the circuit-bootstrap output gadget"""
    P.check_cmux_data(gpu_lib, B=37)
    P.check_cmux_data(gpu_lib, N=2048, l=3, Bgbit=7, B=6, seed=72)


@pytest.mark.parametrize("bits,N,l,Bgbit,bound", [(32, 1024, 2, 10, 4), (32, 1024, 2, 8, 4), (64, 2048, 4, 9, 2 ** 32),
                                                  (64, 1024, 3, 10, 2 ** 32)])
def test_exact_external_product(gpu_lib, bits, N, l, Bgbit, bound):
    """the reference's FFT-free backend on the GPU, bit-exact vs the oracle's exact products; and the
    fp64 external product within the transforms' rounding noise of it"""
    P.check_exact_extprod(gpu_lib, bits, N, l, Bgbit, B=7, fft_bound=bound)


def test_streamed_schedule_hipgraph(gpu_lib):
    """TFHE_AMD_OPT_STREAMED_GRAPH: capture, replay, replay on new data, re-capture"""
    P.check_streamed_graph(gpu_lib, n=40, B=33)


@pytest.mark.gpu
@pytest.mark.parametrize("d,B", [(3, 5), (10, 9), (13, 17)])
def test_lut_eval_gpu(gpu_lib, d, B):
    """LUT evaluation by vertical packing (BASELINE config 3's consumer of the circuit bootstrap)"""
    P.check_lut_eval(gpu_lib, d=d, B=B)
    if d == 13:  # the circuit bootstrap's output gadget (l1=2, Bgbit1=8): bit-compare only
        P.check_lut_eval(gpu_lib, l=2, Bgbit=8, d=d, B=B, decrypt_tol=None)


def test_circuit_bootstrap_pipeline(gpu_lib):
    """BASELINE config 3: tfhe_CircuitBootstrapFFT with the PoC ring sizes and gadgets (N1=1024,
    N2=2048, l2=4, Bgbit2=9, l1=2, Bgbit1=8, preKS 6x2, privKS base 8); n0 and the privKS length are
    shortened so the oracle side finishes in seconds (the full-length blind rotation is covered by
    test_circuit_bootstrap_blind_rotation_full, the full key-switch shapes by the key switch tests)."""
    P.check_circuit_bootstrap(gpu_lib, n0=24, N1=1024, N2=2048, l1=2, bg1=8, l2=4, bg2=9, t10=6, bb10=2, t21=3,
                              bb21=3, B=19)
    P.check_circuit_bootstrap(gpu_lib, n0=5, N1=1024, N2=1024, l1=3, bg1=6, l2=3, bg2=10, t10=4, bb10=3, t21=5,
                              bb21=2, B=3, seed=62)


def test_circuit_bootstrap_full_poc_block(gpu_lib):
    """BASELINE config 3 at the PoC's FULL parameter block (poc:70-85): n0=500, N1=1024, N2=2048, l2=4,
    Bgbit2=9, l1=2, Bgbit1=8, preKS 6x2, privKS 10x3 (2.69 GB table, synthetic).  Every stage and the whole
    tfhe_CircuitBootstrapFFT bit-compared with the oracle on 3 samples."""
    P.check_circuit_bootstrap(gpu_lib, n0=500, N1=1024, N2=2048, l1=2, bg1=8, l2=4, bg2=9, t10=6, bb10=2, t21=10,
                              bb21=3, B=3, seed=63)


def test_privks_two_word_digits(gpu_lib):
    """a private key switch whose digits span both words of the 64-bit inputs (t21 * basebit = 33 > 32): the
    shape the matrix-core kernel does not cover, served by the k_privks fallback (1.5 GB synthetic table)"""
    P.check_circuit_bootstrap(gpu_lib, n0=4, N1=1024, N2=1024, l1=2, bg1=8, l2=3, bg2=10, t10=4, bb10=3, t21=11,
                              bb21=3, B=3, seed=64)


def test_privks_wide_counts_and_pipeline_poc_shape(gpu_lib):
    """the launch shape BASELINE config 3's number comes from: circuitPrivKS at the PoC shape (N2=2048 -> N1=1024, 10 x 3 bits,
    1.34 GB per plane) with 300 and 1031 samples per launch -- several 256-sample tiles, every wave of a tile live, a ragged
    last tile -- and tfhe_CircuitBootstrapFFT on 600 inputs (1200 samples per plane, grouped output layout); against the
    oracle on scattered rows and on the rows around every tile boundary, and against 3-sample launches of the same inputs"""
    P.check_privks_wide(gpu_lib)


def test_privks_wide_counts_other_bases(gpu_lib):
    """the same for the int64 key switch's other digit widths (base 4: two K-steps per coefficient; base 2), 2 tiles each"""
    P.check_privks_wide(gpu_lib, N2=1024, t21=12, bb21=2, counts=(300,), pipeline_B=140, l2=3, bg2=10, seed=66)
    P.check_privks_wide(gpu_lib, N2=1024, t21=20, bb21=1, counts=(259,), pipeline_B=0, l2=3, bg2=10, seed=67)


def test_pool_two_members_on_device_0(gpu_lib):
    """tfhe_amd_pool_* on hardware: two members that share the one GPU (two host threads, two streams, two pinned staging
    buffers, the key uploaded once per member from HOST arrays).  1031 samples = 516 + 515; equal to the single-context
    engine and, on a subset, to the oracle; per-member split reported."""
    N, n, l, Bgbit, t, bb = 1024, 24, 2, 10, 8, 2
    s = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    pool = T.Pool([0, 0], torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=gpu_lib)
    try:
        pool.load_keys(s.bk, s.ks)
        rs = np.random.RandomState(1031)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(1031, n + 1)).astype(np.int32)
        single = s.eng.bootstrap(1 << 29, x)
        for rep in range(3):
            assert np.array_equal(pool.bootstrap(1 << 29, x), single), rep
        counts, seconds = pool.last_split()
        assert counts == [516, 515] and min(seconds) > 0
        for i in (0, 515, 516, 1030):
            assert np.array_equal(single[i], O.bootstrap32(N, s.bk, s.ks, 1 << 29, x[i], l, Bgbit, t, bb)), i
        u = pool.bootstrap_woks(1 << 29, x[:7])
        assert np.array_equal(pool.keyswitch(u), single[:7])
        assert np.array_equal(pool.bootstrap(1 << 29, x[:1]), single[:1])  # one sample: the second member idles
        bk_t = T.keygen_bk_torus(32, s.lwe_key, s.tkey, l, Bgbit, 2.0 ** -25, P.SEED, 1000, lib_path=gpu_lib)
        pool.load_keys_torus(bk_t, None)  # the key again, in coefficient form: converted on the device, replaces the first
        assert np.array_equal(pool.bootstrap(1 << 29, x), single)
        # the pipelined form inside each member: 516 = 128 + 128 + 128 + 128 + 4 on two streams per member, four streams on the GPU
        pool.set_chunk_rows(128)
        for rep in range(3):
            assert np.array_equal(pool.bootstrap(1 << 29, x), single), ("pipelined", rep)
        assert np.array_equal(pool.keyswitch(pool.bootstrap_woks(1 << 29, x)), single)
    finally:
        pool.close()
        s.close()


def test_pool_pipelined_member_full_parameters(gpu_lib):
    """one member, BASELINE config 2's parameters and batch: 4096 host samples through tfhe_amd_pool_bootstrap_host = two chunks
    of 2048 on two streams (the default), and 4500 = 2048 + 2048 + 404; equal to the single-context engine on every row and to
    the oracle on a subset"""
    N, n, l, Bgbit, t, bb = 1024, 630, 2, 10, 8, 2
    s = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    pool = T.Pool([0], torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=gpu_lib)
    try:
        pool.load_keys(s.bk, s.ks)
        rs = np.random.RandomState(4500)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(4500, n + 1)).astype(np.int32)
        single = s.eng.bootstrap(1 << 29, x)
        assert np.array_equal(pool.bootstrap(1 << 29, x[:4096]), single[:4096])
        assert np.array_equal(pool.bootstrap(1 << 29, x), single)
        for i in (0, 2047, 2048, 4095, 4096, 4499):
            assert np.array_equal(single[i], O.bootstrap32(N, s.bk, s.ks, 1 << 29, x[i], l, Bgbit, t, bb)), i
    finally:
        pool.close()
        s.close()


def test_circuit_bootstrap_pool_two_members_on_device_0(gpu_lib):
    n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21 = 6, 1024, 2048, 2, 8, 4, 9, 6, 2, 3, 3
    key0, key2 = O.keygen_binary(n0, P.SEED, 21), O.keygen_binary(N2, P.SEED, 23)
    bk = O.bk_create64(N2, key0, key2, l2, bg2, 2.0 ** -44, P.SEED, 3000)
    preks = O.fill32(101, N1 * t10 * (1 << bb10) * (n0 + 1)).reshape(N1, t10, 1 << bb10, n0 + 1)
    privks = O.fill32(202, 2 * (N2 + 1) * t21 * (1 << bb21) * 2 * N1).reshape(2, N2 + 1, t21, 1 << bb21, 2, N1)
    pool = T.CircuitBootstrapPool([0, 0], n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, lib_path=gpu_lib)
    try:
        pool.load_preks(preks)
        pool.load_bk_fft(bk)
        pool.load_privks(privks)
        x = np.random.RandomState(9).randint(-2 ** 31, 2 ** 31, size=(37, N1 + 1)).astype(np.int32)
        got = pool.circuit_bootstrap(x)
        for b in (0, 18, 19, 36):
            want = O.circuit_bootstrap(x[b], preks, bk, privks, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21)
            assert np.array_equal(got[b], want), b
        pool.set_chunk_rows(4)  # the pipelined member: 19 = 4 + 4 + 4 + 4 + 3 on three streams, in both members at once
        for rep in range(2):
            assert np.array_equal(pool.circuit_bootstrap(x), got), rep
    finally:
        pool.close()


def test_keys_as_device_layout_bytes(gpu_lib):
    """the bytes a key broadcast carries: context A's resident keys exported in their device layout, imported by context B
    (nothing regenerated or re-converted; the matrix-core key-switch layout rebuilt locally), same outputs"""
    N, n, l, Bgbit, t, bb = 1024, 12, 2, 10, 8, 2
    src = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    dst = T.Engine(torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=gpu_lib)
    try:
        bk_bytes = np.empty(src.eng.gsw_packed_bytes(n), np.uint8)
        ks_bytes = np.empty(src.eng.keyswitch_key_bytes(), np.uint8)
        src.eng.gsw_export_packed(src.gsw, T._np_ptr(bk_bytes))
        src.eng.keyswitch_key_export(T._np_ptr(ks_bytes))
        assert np.array_equal(ks_bytes.view(np.int32).reshape(src.ks.shape), src.ks)
        # through device memory, as a collective delivers them
        d_bk, d_ks = dst.to_device(bk_bytes), dst.to_device(ks_bytes)
        dst.set_bootstrap_key(dst.gsw_from_packed(d_bk.ptr, n))
        dst.load_keyswitch_key_d(d_ks.ptr)
        x = np.random.RandomState(3).randint(-2 ** 31, 2 ** 31, size=(70, n + 1)).astype(np.int32)
        got = dst.bootstrap(1 << 29, x)
        assert np.array_equal(got, src.eng.bootstrap(1 << 29, x))
        assert np.array_equal(got[5], O.bootstrap32(N, src.bk, src.ks, 1 << 29, x[5], l, Bgbit, t, bb))
    finally:
        dst.close()
        src.close()


def test_device_identity_and_clock_probe(gpu_lib):
    """PCI bus id of the device behind ordinal 0 (what the bench line's rank table carries), and the shader-clock probe
    run BESIDE a blind rotation that fills the chip: a plausible clock, below the 2.4 GHz the data sheet names"""
    import re
    assert T.device_count(gpu_lib) >= 1
    bus = T.device_pci_bus_id(0, gpu_lib)
    assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F]", bus), bus
    assert bus in T.device_info(0, gpu_lib)
    s = P.GateSetup(gpu_lib, 1024, 64, 2, 10, 8, 2)
    try:
        x = np.random.RandomState(1).randint(-2 ** 31, 2 ** 31, size=(4096, 65)).astype(np.int32)
        d_x, d_o = s.eng.to_device(x), s.eng.alloc(4096 * 1025 * 4)
        idle = s.eng.clock_probe(500)
        s.eng._chk(s.eng.lib.tfhe_amd_bootstrap_woks(s.eng.ctx, d_o.ptr, 1 << 29, d_x.ptr, 4096))  # ~1.7 ms of full-chip work
        busy = s.eng.clock_probe(1000)
        s.eng.sync()
        for med, lo, hi in (idle, busy):
            assert 0.3 < lo <= med <= hi < 2.6, (idle, busy)
    finally:
        s.close()


def test_batch_4096_properties(gpu_lib):
    """BASELINE config 2 at full size: 4096 gate bootstraps.  Checked by (i) decrypt-sign of every
    output, (ii) bit-equality with the oracle on a subset, (iii) persistent schedule == one launch
    per CMux, (iv) position independence: duplicated inputs give identical outputs."""
    N, n, l, Bgbit, t, bb, B = 1024, 630, 2, 10, 8, 2, 4096
    s = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    try:
        mu = 1 << 29
        rs = np.random.RandomState(99)
        base = 64
        msgs = [mu if rs.randint(2) else -mu for _ in range(base)]
        xb = s.encrypt(msgs)
        # 4096 samples: the 64 real encryptions tiled, then re-randomised by adding encryptions of 0
        # is unnecessary for the property -- duplicates are the point of (iv)
        x = np.tile(xb, (B // base, 1))
        out = s.eng.bootstrap(mu, x)
        ph = np.array([O.lwe_phase32(out[i], s.lwe_key) for i in range(B)])
        want_sign = np.tile(np.array(msgs) > 0, B // base)
        assert np.array_equal(ph > 0, want_sign), "decrypt-sign"
        assert np.abs(np.abs(ph.astype(np.int64)) - mu).max() < mu // 2
        assert np.array_equal(out, np.tile(out[:base], (B // base, 1))), "position independence"
        sub = rs.choice(base, 12, replace=False)
        want = np.stack([O.bootstrap32(N, s.bk, s.ks, mu, xb[i], l, Bgbit, t, bb) for i in sub])
        assert np.array_equal(out[sub], want), "oracle subset"
        out2 = s.eng.bootstrap(mu, x[:512], streamed=True)
        assert np.array_equal(out2, out[:512]), "streamed schedule"
        # throughput workload: uniformly random samples, subset vs oracle
        xr = rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32)
        outr = s.eng.bootstrap(mu, xr)
        sub = rs.choice(B, 8, replace=False)
        want = np.stack([O.bootstrap32(N, s.bk, s.ks, mu, xr[i], l, Bgbit, t, bb) for i in sub])
        assert np.array_equal(outr[sub], want), "oracle subset, random samples"
    finally:
        s.close()


def test_config5_single_gpu_leg_one_launch(gpu_lib):
    """BASELINE config 5, the leg one GPU can run: 2^20 gate bootstraps in ONE launch of the blind rotation and ONE
    of the key switch (what every rank of `bench.py --gpus N` does with its slice).  Checked by (i) bit-equality with
    the oracle on 16 scattered samples, (ii) the last samples of the batch being copies of the first ones (64-bit
    offsets at the far end of every buffer), (iii) every 65,536th output equal to a small-batch run of the same
    inputs -- which the library serves with its other blind-rotation kernel, (iv) decrypt-sign of real encryptions
    placed at both ends."""
    N, n, l, Bgbit, t, bb, B = 1024, 630, 2, 10, 8, 2, 1 << 20
    s = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    try:
        mu = 1 << 29
        rs = np.random.RandomState(5)
        x = np.frombuffer(rs.bytes(B * (n + 1) * 4), dtype=np.int32).reshape(B, n + 1).copy()  # uniformly random samples
        msgs = [mu if (i % 3) else -mu for i in range(8)]
        real = s.encrypt(msgs)
        x[:8] = real
        x[B // 2:B // 2 + 8] = real
        ntail = 24
        x[B - ntail:] = x[:ntail]
        out = s.eng.bootstrap(mu, x)
        assert out.shape == (B, n + 1)
        assert np.array_equal(out[B - ntail:], out[:ntail]), "far end of the batch"
        assert np.array_equal(out[B // 2:B // 2 + 8], out[:8]), "middle of the batch"
        for i in range(8):
            assert (O.lwe_phase32(out[i], s.lwe_key) > 0) == (msgs[i] > 0), "decrypt-sign"
        strided = np.arange(0, B, 65536)
        small = s.eng.bootstrap(mu, np.ascontiguousarray(x[strided]))
        assert np.array_equal(out[strided], small), "every 65,536th output vs a 16-sample launch of the same inputs"
        scattered = np.concatenate([[0, 1, 65535, 65536, B // 2 - 1, B // 2 + 8, B - ntail - 1, B - 1],
                                    rs.choice(B, 8, replace=False)])
        want = np.stack([O.bootstrap32(N, s.bk, s.ks, mu, x[i], l, Bgbit, t, bb) for i in scattered])
        assert np.array_equal(out[scattered], want), "oracle, 16 scattered samples"
    finally:
        s.close()


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 6, 7])
def test_gate_path_latency_kernel(gpu_lib, B):
    """BASELINE config 1's shape (the reference bootstraps one sample per call, lwe_functions.cpp:434-446): batches
    1..7 through the latency-shaped blind rotation (one ciphertext per 4-wave workgroup) AND through the
    one-wave-per-ciphertext kernel, every entry point of the gate path bit-compared with the oracle"""
    P.check_gate_path(gpu_lib, N=1024, n=11, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=B, check_export=False, seed=100 + B,
                      br_split=1 << 30)
    P.check_gate_path(gpu_lib, N=1024, n=11, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=B, check_export=False, seed=100 + B,
                      br_split=0)


def test_gate_path_latency_kernel_full_parameters(gpu_lib):
    """n = 630 on the latency-shaped kernel, including a batch above one workgroup per CU (300 > 256)"""
    P.check_gate_path(gpu_lib, N=1024, n=630, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=3, check_export=False, br_split=1 << 30)
    N, n, l, Bgbit, t, bb, B = 1024, 630, 2, 10, 8, 2, 300
    s = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    try:
        rs = np.random.RandomState(300)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32)
        s.eng.set_option(T.OPT_BR_SPLIT, 1 << 30)
        a = s.eng.bootstrap(1 << 29, x)
        s.eng.set_option(T.OPT_BR_SPLIT, 0)
        b = s.eng.bootstrap(1 << 29, x)
        assert np.array_equal(a, b), "latency-shaped kernel vs one wave per ciphertext, 300 samples"
        sub = rs.choice(B, 6, replace=False)
        want = np.stack([O.bootstrap32(N, s.bk, s.ks, 1 << 29, x[i], l, Bgbit, t, bb) for i in sub])
        assert np.array_equal(a[sub], want)
    finally:
        s.close()


def test_gate_path_workgroup_widths(gpu_lib):
    """one wave per ciphertext in its two workgroup widths: 4 waves (batches up to 1024, here 1000) and 8 waves
    (above: 1031 = 128 full workgroups + a ragged one), both against the latency-shaped kernel on the same inputs
    and the oracle on a subset"""
    N, n, l, Bgbit, t, bb = 1024, 40, 2, 10, 8, 2
    s = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    try:
        rs = np.random.RandomState(1031)
        for B in (1000, 1031):
            x = rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32)
            s.eng.set_option(T.OPT_BR_SPLIT, 0)
            one_wave = s.eng.bootstrap(1 << 29, x)
            s.eng.set_option(T.OPT_BR_SPLIT, 1 << 30)
            split = s.eng.bootstrap(1 << 29, x)
            assert np.array_equal(one_wave, split), B
            sub = rs.choice(B, 6, replace=False)
            want = np.stack([O.bootstrap32(N, s.bk, s.ks, 1 << 29, x[i], l, Bgbit, t, bb) for i in sub])
            assert np.array_equal(one_wave[sub], want), B
    finally:
        s.close()


def test_gate_path_8wave_workgroups_other_gadgets(gpu_lib):
    """the 8-wave instantiations <10,8,2,2,8>, <10,8,2,2> and <10,8,2> (two waves per SIMD + the partner balance) are only
    reachable above 1024 samples since the 4-wave form serves the batches below: B = 1031 (128 full workgroups + a ragged
    one) for (l=2, Bgbit=8), (l=2, Bgbit=9) and (l=3, Bgbit=7)"""
    for l, Bgbit, seed in ((2, 8, 31), (2, 9, 32), (3, 7, 33)):
        P.check_gate_wide_batch(gpu_lib, l=l, Bgbit=Bgbit, B=1031, seed=seed)


def test_kernel_choice_boundaries(gpu_lib):
    """the library picks the blind-rotation form by batch size (split kernel up to 512, 4-wave workgroups up to 1024, 8-wave
    above): a sample's output must not depend on the batch it travels in -- batches on both sides of each boundary, the
    library's own choice, against one reference run (oracle-checked on a subset) of the same inputs"""
    N, n, l, Bgbit, t, bb = 1024, 8, 2, 10, 8, 2
    s = P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb)
    try:
        rs = np.random.RandomState(512)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(1026, n + 1)).astype(np.int32)
        ref = s.eng.bootstrap(1 << 29, x)
        for i in (0, 511, 512, 1023, 1024, 1025):
            assert np.array_equal(ref[i], O.bootstrap32(N, s.bk, s.ks, 1 << 29, x[i], l, Bgbit, t, bb)), i
        for B in (1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025):
            assert np.array_equal(s.eng.bootstrap(1 << 29, x[:B]), ref[:B]), B
            assert np.array_equal(s.eng.bootstrap_woks(1 << 29, x[:B]), s.eng.bootstrap_woks(1 << 29, x)[:B]), B
    finally:
        s.close()


def test_two_contexts_on_two_host_threads(gpu_lib):
    """SURVEY 8(b) threading contract: a context is bound to one stream and is not thread-safe, DISTINCT contexts are
    independent -- two host threads, one context each (own stream, own key replicas), bootstrapping different batches at the
    same time, several times over; every result equals the oracle-checked single-threaded one"""
    import threading
    N, n, l, Bgbit, t, bb = 1024, 24, 2, 10, 8, 2
    setups = [P.GateSetup(gpu_lib, N, n, l, Bgbit, t, bb) for _ in range(2)]
    try:
        rs = np.random.RandomState(77)
        xs = [rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32) for B in (700, 37)]
        want = [setups[0].eng.bootstrap(1 << 29, x) for x in xs]
        for i in (0, 1):
            assert np.array_equal(want[i][3], O.bootstrap32(N, setups[0].bk, setups[0].ks, 1 << 29, xs[i][3], l, Bgbit, t, bb))
        bad = []

        def work(k):
            try:
                for rep in range(6):
                    got = setups[k].eng.bootstrap(1 << 29, xs[k])
                    if not np.array_equal(got, want[k]):
                        bad.append((k, rep))
            except Exception as e:  # surfaces in the main thread's assert
                bad.append((k, repr(e)))

        th = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not bad, bad
    finally:
        for s in setups:
            s.close()


def test_streamed_graph_across_kernel_classes(gpu_lib):
    P.check_streamed_graph_batch_classes(gpu_lib)
    P.check_streamed_graph_batch_classes(gpu_lib, br_split=1 << 30, order=(8, 1031))  # the 0-step launches change kernel form


def test_gate_path_latency_kernel_other_bgbit(gpu_lib):
    """k_blind_rotate_split<8> (the circuit bootstrap's output gadget) and <0> (Bgbit read at run time)"""
    P.check_gate_path(gpu_lib, N=1024, n=7, l=2, Bgbit=8, ks_t=8, ks_bb=2, B=5, check_export=False, seed=14, br_split=1 << 30)
    P.check_gate_path(gpu_lib, N=1024, n=6, l=2, Bgbit=9, ks_t=8, ks_bb=2, B=5, check_export=False, seed=12, br_split=1 << 30)


def test_n2048_transform_batch_8192(gpu_lib):
    """BASELINE config 4: N=2048 transforms, batch 8192.  Round-trip and linearity properties on
    the whole batch, oracle equality on a subset."""
    N, B = 2048, 8192
    rs = np.random.RandomState(123)
    e = T.Engine(torus_bits=64, n=1, N=N, l=4, Bgbit=9, ks_t=0, lib_path=gpu_lib)
    try:
        dig = rs.randint(-256, 256, size=(B, N)).astype(np.int32)
        lag = e.ifft_int32(dig)
        sub = rs.choice(B, 16, replace=False)
        assert P.same_doubles(lag[sub], O.execute_reverse_int(N, dig[sub]))
        # small integers survive the round trip up to the reference's truncation toward zero
        # (execute_direct_torus64 truncates: 108.99999... -> 108), and exactly as the oracle's do
        back = e.fft_torus64(lag)
        assert np.abs(back - dig.astype(np.int64)).max() <= 1, "round trip on digits"
        assert np.array_equal(back[sub], O.execute_direct_torus64(N, lag[sub])), "round trip vs oracle"
        a64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(B, N), dtype=np.int64)
        l64 = e.ifft_torus64(a64)
        assert P.same_doubles(l64[sub], O.execute_reverse_torus64(N, a64[sub]))
        t64 = e.fft_torus64(l64)
        assert np.array_equal(t64[sub], O.execute_direct_torus64(N, l64[sub]))
        # Torus64 round trip keeps the top 53-ish bits (execute_reverse_torus64 drops 11)
        err = (t64 - a64).astype(np.int64)
        assert np.abs(err).max() < 2 ** 14
    finally:
        e.close()


@pytest.mark.parametrize("N,B", [(1024, 40000), (2048, 12000)])
def test_transform_batches_beyond_the_cache(gpu_lib, N, B):
    """launches whose working set exceeds the 256 MB Infinity Cache take the nontemporal variants of the transform
    kernels (lane-contiguous loads / stores with the nt policy): same bits.  Round trip on the whole batch, the oracle
    on a subset, both torus widths."""
    rs = np.random.RandomState(N + B)
    e = T.Engine(torus_bits=64, n=1, N=N, l=4, Bgbit=9, ks_t=0, lib_path=gpu_lib)
    try:
        sub = rs.choice(B, 12, replace=False)
        a64 = np.frombuffer(rs.bytes(B * N * 8), dtype=np.int64).reshape(B, N)
        l64 = e.ifft_torus64(a64)                                  # B * N * 16 bytes: 655 MB / 393 MB
        assert P.same_doubles(l64[sub], O.execute_reverse_torus64(N, a64[sub])), "execute_reverse_torus64"
        t64 = e.fft_torus64(l64)
        assert np.array_equal(t64[sub], O.execute_direct_torus64(N, l64[sub])), "execute_direct_torus64"
        assert np.abs((t64 - a64).astype(np.int64)).max() < 2 ** 14
        del a64, t64
        dig = rs.randint(-512, 512, size=(B, N)).astype(np.int32)
        lag = e.ifft_int32(dig)                                    # B * N * 12 bytes: 491 MB / 295 MB
        assert P.same_doubles(lag[sub], O.execute_reverse_int(N, dig[sub])), "execute_reverse_int"
        t32 = e.fft_torus32(lag)
        assert np.array_equal(t32[sub], O.execute_direct_torus32(N, lag[sub])), "execute_direct_torus32"
        assert np.abs(t32.astype(np.int64) - dig).max() <= 1
    finally:
        e.close()


def test_abi_edges(gpu_lib):
    """empty batches, calls in the wrong state, bad arguments (status codes of include/tfhe_amd.h)"""
    P.check_abi_edges(gpu_lib)
