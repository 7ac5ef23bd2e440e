"""Several devices in one process, on the CPU emulator of the kernels (tests/emu, TFHE_EMU_DEVICES=8 from conftest.py).
The emulator tags every allocation, stream, event, captured graph and per-kernel LDS attribute with its device and ABORTS on
a launch / copy / free / event record whose operands belong to another device than the calling thread's current one, so
every test here also proves that the host library selects the right device on every path it takes.

Covers: contexts on devices 0 and 3 on two host threads; the pool (tfhe_amd_pool_*: the caller's keys uploaded once per
member, contiguous slices, one host thread per member) with 2 and 8 members and ragged counts, equal to the single-device
run and to the oracle; the circuit-bootstrap pool; keys handed over as the bytes of their device layout; and that the
emulator's checks do fire (a child process that misuses a buffer across devices must die)."""
import importlib
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import oracle_py as O
import parity_checks as P

T = importlib.import_module("experimental-tfhe_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulator_presents_eight_devices(emu_lib):
    assert T.device_count(emu_lib) == 8
    ids = [T.device_pci_bus_id(d, emu_lib) for d in range(8)]
    assert len(set(ids)) == 8 and ids[3] == "0000:e3:00.0"
    assert T.device_pci_bus_id(8, emu_lib) is None
    assert "ordinal 5 of 8" in T.device_info(5, emu_lib) and "PCI 0000:e5:00.0" in T.device_info(5, emu_lib)
    with pytest.raises(T.TfheAmdError):
        T.Engine(torus_bits=32, n=2, N=1024, l=2, Bgbit=10, ks_t=0, device=8, lib_path=emu_lib)


def test_two_contexts_on_devices_0_and_3(emu_lib):
    """the GPU suite's two-context test (test_two_contexts_on_two_host_threads) with the contexts on DIFFERENT devices:
    every entry point of the gate path, the streamed schedule and its captured graph, on device 3 while device 0 works"""
    N, n, l, Bgbit, t, bb = 1024, 3, 2, 10, 8, 2
    setups = [P.GateSetup(emu_lib, N, n, l, Bgbit, t, bb, device=d) for d in (0, 3)]
    try:
        rs = np.random.RandomState(77)
        xs = [rs.randint(-2 ** 31, 2 ** 31, size=(B, n + 1)).astype(np.int32) for B in (5, 3)]
        want = [np.stack([O.bootstrap32(N, setups[0].bk, setups[0].ks, 1 << 29, x[i], l, Bgbit, t, bb) for i in range(len(x))]) for x in xs]
        bad = []

        def work(k):
            try:
                e = setups[k].eng
                for rep in range(2):
                    if not np.array_equal(e.bootstrap(1 << 29, xs[k]), want[k]):
                        bad.append((k, rep, "bootstrap"))
                    if not np.array_equal(e.keyswitch(e.bootstrap_woks(1 << 29, xs[k])), want[k]):
                        bad.append((k, rep, "woks + keyswitch"))
                e.set_option(T.OPT_STREAMED_GRAPH, 1)
                for rep in range(3):  # plain, capture, replay
                    if not np.array_equal(e.bootstrap(1 << 29, xs[k], streamed=True), want[k]):
                        bad.append((k, rep, "streamed"))
                ev0, ev1 = e.event(), e.event()
                e.record(ev0)
                e.record(ev1)
                assert e.elapsed_ms(ev0, ev1) >= 0
                assert e.clock_probe(200)[0] > 0
            except Exception as ex:  # surfaces in the main thread's assert
                bad.append((k, repr(ex)))

        th = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not bad, bad
    finally:
        for s in setups:
            s.close()


def test_other_paths_on_device_5(emu_lib):
    """the paths the gate test does not reach, on a device that is not 0: transforms, Torus64 blind rotation, LUT
    evaluation (its own device-side constant table), the Real96 transforms (lazily uploaded twiddles), the circuit bootstrap"""
    rs = np.random.RandomState(5)
    e = T.Engine(torus_bits=64, n=2, N=2048, l=4, Bgbit=9, ks_t=0, device=5, lib_path=emu_lib)
    try:
        a64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(3, 2048), dtype=np.int64)
        lag = e.ifft_torus64(a64)
        assert P.same_doubles(lag, O.execute_reverse_torus64(2048, a64))
        assert np.array_equal(e.fft_torus64(lag), O.execute_direct_torus64(2048, lag))
        spec = e.hp_ifft(a64[:1])
        assert e.hp_fft(spec).shape == (1, 2048)
    finally:
        e.close()
    cb = T.CircuitBootstrap(2, 1024, 1024, 2, 8, 3, 10, 2, 2, 2, 3, device=5, lib_path=emu_lib)
    try:
        key0, key2 = O.keygen_binary(2, P.SEED, 21), O.keygen_binary(1024, P.SEED, 23)
        bk = O.bk_create64(1024, key0, key2, 3, 10, 2.0 ** -44, P.SEED, 3000)
        preks = O.fill32(101, 1024 * 2 * 4 * 3).reshape(1024, 2, 4, 3)
        privks = O.fill32(202, 2 * 1025 * 2 * 8 * 2 * 1024).reshape(2, 1025, 2, 8, 2, 1024)
        cb.load_preks(preks)
        cb.load_bk_fft(bk)
        cb.load_privks(privks)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(2, 1025)).astype(np.int32)
        want = np.stack([O.circuit_bootstrap(x[b], preks, bk, privks, 2, 1024, 1024, 2, 8, 3, 10, 2, 2, 2, 3) for b in range(2)])
        assert np.array_equal(cb.circuit_bootstrap(x), want)
        tgsw, lwe = cb.circuit_bootstrap_lut(x, 2, O.fill32(77, 1024))
        assert np.array_equal(tgsw, want)
    finally:
        cb.close()


@pytest.mark.parametrize("devices,count", [([0, 1], 7), ([2, 2], 5), ([0, 1, 2, 3, 4, 5, 6, 7], 19), ([0, 1, 2, 3, 4, 5, 6, 7], 3), ([4], 4)])
def test_pool_equals_single_device_and_oracle(emu_lib, devices, count):
    """tfhe_amd_pool_*: keys from HOST arrays (the caller's key, not a seed), one upload per member; ragged contiguous
    slices (19 over 8 = 3,3,3,2,...; 3 over 8 leaves five members idle; two members on ONE device); every output equal to the
    single-device engine's and the oracle's"""
    N, n, l, Bgbit, t, bb = 1024, 2, 2, 10, 8, 2
    s = P.GateSetup(emu_lib, N, n, l, Bgbit, t, bb)
    pool = T.Pool(devices, torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=emu_lib)
    try:
        pool.load_keys(s.bk, s.ks)
        rs = np.random.RandomState(count)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(count, n + 1)).astype(np.int32)
        single = s.eng.bootstrap(1 << 29, x)
        got = pool.bootstrap(1 << 29, x)
        assert np.array_equal(got, single)
        counts, seconds = pool.last_split()
        base, rem = divmod(count, len(devices))
        assert counts == [base + (1 if r < rem else 0) for r in range(len(devices))]
        assert all((c > 0) == (sec > 0) for c, sec in zip(counts, seconds))
        for i in (0, count - 1):
            assert np.array_equal(got[i], O.bootstrap32(N, s.bk, s.ks, 1 << 29, x[i], l, Bgbit, t, bb))
        u = pool.bootstrap_woks(1 << 29, x)
        assert np.array_equal(u, s.eng.bootstrap_woks(1 << 29, x))
        assert np.array_equal(pool.keyswitch(u), single)
        assert pool.bootstrap(1 << 29, x[:0]).shape == (0, n + 1)  # empty batch: a no-op on every member
        # the same key in coefficient form: converted on every device (tGswToFFTConvert), a second load replaces the first
        bk_t = T.keygen_bk_torus(32, s.lwe_key, s.tkey, l, Bgbit, 2.0 ** -25, P.SEED, 1000, lib_path=emu_lib)
        pool.load_keys_torus(bk_t, None)
        assert np.array_equal(pool.bootstrap(1 << 29, x), single)
    finally:
        pool.close()
        s.close()


@pytest.mark.parametrize("devices,count,chunk", [([3], 7, 2), ([1, 5], 11, 2), ([6, 6], 9, 1), ([2], 5, 0)])
def test_pool_pipelined_chunks(emu_lib, devices, count, chunk):
    """the pipelined form inside a member -- chunks issued alternately on two streams with two staging sets (chunk k + 1 copied
    in while chunk k computes) -- with a ragged last chunk (7 = 2 + 2 + 2 + 1), two members pipelining at once, two members of
    ONE device, and switched off: the same bits as the single-context engine, for all three sharded operations"""
    N, n, l, Bgbit, t, bb = 1024, 2, 2, 10, 8, 2
    s = P.GateSetup(emu_lib, N, n, l, Bgbit, t, bb)
    pool = T.Pool(devices, torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=emu_lib)
    try:
        pool.load_keys(s.bk, s.ks)
        pool.set_chunk_rows(chunk)
        rs = np.random.RandomState(count + chunk)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(count, n + 1)).astype(np.int32)
        single = s.eng.bootstrap(1 << 29, x)
        for rep in range(2):  # the second call reuses streams and staging sets
            assert np.array_equal(pool.bootstrap(1 << 29, x), single), rep
        u = pool.bootstrap_woks(1 << 29, x)
        assert np.array_equal(u, s.eng.bootstrap_woks(1 << 29, x))
        assert np.array_equal(pool.keyswitch(u), single)
        assert np.array_equal(pool.bootstrap(1 << 29, x[:1]), single[:1])  # a short call after a pipelined one: one piece
        with pytest.raises(T.TfheAmdError):
            pool.set_chunk_rows(-1)
    finally:
        pool.close()
        s.close()


def test_pool_rows_callbacks(emu_lib):
    """tfhe_amd_pool_*_rows: the members fetch and deliver rows through the caller's two callbacks (here: rows kept in REVERSE order
    in the caller's arrays), on their own threads, for disjoint ranges; pipelined chunks of 2 over two devices"""
    import ctypes as C
    N, n, l, Bgbit, t, bb = 1024, 2, 2, 10, 8, 2
    s = P.GateSetup(emu_lib, N, n, l, Bgbit, t, bb)
    pool = T.Pool([2, 5], torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=emu_lib)
    try:
        pool.load_keys(s.bk, s.ks)
        pool.set_chunk_rows(2)
        count = 9
        x = np.random.RandomState(9).randint(-2 ** 31, 2 ** 31, size=(count, n + 1)).astype(np.int32)
        xr = np.ascontiguousarray(x[::-1])          # the caller's storage: row r of the call lives at index count - 1 - r
        outr = np.zeros((count, n + 1), np.int32)
        seen = []
        IN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32))
        OUT = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32))

        def get(user, first, rows, dst):
            seen.append((first, rows))
            for r in range(rows):
                C.memmove(C.addressof(dst.contents) + r * (n + 1) * 4, xr[count - 1 - (first + r)].ctypes.data, (n + 1) * 4)

        def put(user, first, rows, src):
            for r in range(rows):
                C.memmove(outr[count - 1 - (first + r)].ctypes.data, C.addressof(src.contents) + r * (n + 1) * 4, (n + 1) * 4)

        lib = pool.lib
        lib.tfhe_amd_pool_bootstrap_rows.argtypes = [C.c_void_p, OUT, IN, C.c_void_p, C.c_int32, C.c_int]
        g, p = IN(get), OUT(put)
        assert lib.tfhe_amd_pool_bootstrap_rows(pool.pool, p, g, None, 1 << 29, count) == T.OK
        assert np.array_equal(outr[::-1], s.eng.bootstrap(1 << 29, x))
        assert sorted(seen) == [(0, 2), (2, 2), (4, 1), (5, 2), (7, 2)]  # member 0: rows 0-4 in chunks of 2, member 1: rows 5-8
        assert lib.tfhe_amd_pool_bootstrap_rows(pool.pool, C.cast(None, OUT), g, None, 1 << 29, count) == T.ERR_PARAM  # a null callback
    finally:
        pool.close()
        s.close()


def test_pool_calls_from_several_host_threads_are_serialised(emu_lib):
    """a pool is not re-entrant: calls from several host threads take its one lock and run one after the other, each with the
    right answer (the shims' array forms rely on it)"""
    N, n, l, Bgbit, t, bb = 1024, 2, 2, 10, 8, 2
    s = P.GateSetup(emu_lib, N, n, l, Bgbit, t, bb)
    pool = T.Pool([0, 7], torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=emu_lib)
    try:
        pool.load_keys(s.bk, s.ks)
        pool.set_chunk_rows(1)
        rs = np.random.RandomState(4)
        xs = [rs.randint(-2 ** 31, 2 ** 31, size=(c, n + 1)).astype(np.int32) for c in (5, 3, 6, 2)]
        want = [s.eng.bootstrap(1 << 29, x) for x in xs]
        bad = []

        def work(k):
            try:
                for rep in range(2):
                    if not np.array_equal(pool.bootstrap(1 << 29, xs[k]), want[k]):
                        bad.append((k, rep))
            except Exception as ex:
                bad.append((k, repr(ex)))

        th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not bad, bad
    finally:
        pool.close()
        s.close()


def test_pool_errors(emu_lib):
    with pytest.raises(T.TfheAmdError):
        T.Pool([0, 9], n=2, lib_path=emu_lib)  # device 9 does not exist: the whole pool fails, nothing half-made is returned
    pool = T.Pool([1, 6], n=2, lib_path=emu_lib)
    try:
        x = np.zeros((3, 3), np.int32)
        with pytest.raises(T.TfheAmdError, match="member 0 .device 1.*key"):
            pool.bootstrap(1 << 29, x)  # no keys yet: ERR_STATE from the member, with its device in the message
    finally:
        pool.close()


def test_pool_uniform_load_failure_is_not_a_mixed_state(emu_lib):
    """a key load refused by EVERY member alike (here: a key-switch key offered to contexts without key-switch parameters) leaves
    all members with what they had -- nothing differs between them, so the operations on the key that did load keep running"""
    N, n, l, Bgbit = 1024, 2, 2, 10
    lk, tk = O.keygen_binary(n, P.SEED, 1), O.keygen_binary(N, P.SEED, 2)
    bk = O.bk_create32(N, lk, tk, l, Bgbit, 2.0 ** -25, P.SEED, 1000)
    pool = T.Pool([2, 4], torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=0, lib_path=emu_lib)
    try:
        with pytest.raises(T.TfheAmdError, match="no key-switch parameters"):
            pool.load_keys(bk, np.zeros(16, np.int32))  # bk loads everywhere, ks is refused everywhere
        x = np.random.RandomState(3).randint(-2 ** 31, 2 ** 31, size=(3, n + 1)).astype(np.int32)
        want = np.stack([O.bootstrap_woks32(N, bk, 1 << 29, x[i], l, Bgbit) for i in range(3)])
        assert np.array_equal(pool.bootstrap_woks(1 << 29, x), want)
    finally:
        pool.close()


def test_circuit_bootstrap_pool(emu_lib):
    n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21 = 2, 1024, 1024, 2, 8, 3, 10, 2, 2, 2, 3
    key0, key2 = O.keygen_binary(n0, P.SEED, 21), O.keygen_binary(N2, P.SEED, 23)
    bk = O.bk_create64(N2, key0, key2, l2, bg2, 2.0 ** -44, P.SEED, 3000)
    preks = O.fill32(101, N1 * t10 * (1 << bb10) * (n0 + 1)).reshape(N1, t10, 1 << bb10, n0 + 1)
    privks = O.fill32(202, 2 * (N2 + 1) * t21 * (1 << bb21) * 2 * N1).reshape(2, N2 + 1, t21, 1 << bb21, 2, N1)
    pool = T.CircuitBootstrapPool([3, 6], n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, lib_path=emu_lib)
    try:
        pool.load_preks(preks)
        pool.load_bk_fft(bk)
        pool.load_privks(privks)
        x = np.random.RandomState(9).randint(-2 ** 31, 2 ** 31, size=(3, N1 + 1)).astype(np.int32)
        want = np.stack([O.circuit_bootstrap(x[b], preks, bk, privks, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21) for b in range(3)])
        assert np.array_equal(pool.circuit_bootstrap(x), want)
        pool.set_chunk_rows(1)  # member 0's two inputs as two pipelined chunks (copy-in / compute / copy-out streams), member 1's one as one piece
        assert np.array_equal(pool.circuit_bootstrap(x), want)
    finally:
        pool.close()


def test_keys_as_device_layout_bytes_between_devices(emu_lib):
    """tfhe_amd_gsw_export_packed / _from_packed / keyswitch_key_export / load_keyswitch_key_d: device 1's resident keys
    become device 6's without regenerating or re-converting anything (what a broadcast delivers)"""
    N, n, l, Bgbit, t, bb = 1024, 2, 2, 10, 8, 2
    src = P.GateSetup(emu_lib, N, n, l, Bgbit, t, bb, device=1)
    dst = T.Engine(torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, device=6, lib_path=emu_lib)
    try:
        assert src.eng.gsw_packed_bytes(n) == n * 2 * l * 2 * N * 8
        bk_bytes = np.empty(src.eng.gsw_packed_bytes(n), np.uint8)
        ks_bytes = np.empty(src.eng.keyswitch_key_bytes(), np.uint8)
        assert ks_bytes.size == N * t * (1 << bb) * (n + 1) * 4
        src.eng.gsw_export_packed(src.gsw, T._np_ptr(bk_bytes))
        src.eng.keyswitch_key_export(T._np_ptr(ks_bytes))
        assert np.array_equal(ks_bytes.view(np.int32).reshape(src.ks.shape), src.ks)
        dst.set_bootstrap_key(dst.gsw_from_packed(T._np_ptr(bk_bytes), n))
        dst.load_keyswitch_key_d(T._np_ptr(ks_bytes))
        x = np.random.RandomState(3).randint(-2 ** 31, 2 ** 31, size=(4, n + 1)).astype(np.int32)
        assert np.array_equal(dst.bootstrap(1 << 29, x), src.eng.bootstrap(1 << 29, x))
    finally:
        dst.close()
        src.close()


MISUSE = r"""
import importlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
T = importlib.import_module("experimental-tfhe_amd")
a = T.Engine(torus_bits=32, n=1, N=1024, l=2, Bgbit=10, ks_t=0, device=0, lib_path=%(lib)r)
b = T.Engine(torus_bits=32, n=1, N=1024, l=2, Bgbit=10, ks_t=0, device=2, lib_path=%(lib)r)
x = np.zeros((1, 1024), np.int32)
mode = sys.argv[1]
if mode == "launch":      # device 2's buffer handed to a launch of device 0's context
    d_in, d_out = b.to_device(x), a.alloc(8192)
    a._chk(a.lib.tfhe_amd_ifft_int32(a.ctx, d_out.ptr, d_in.ptr, 1))
elif mode == "copy":      # device 0's context asked to fill device 2's buffer
    d = b.alloc(4096)
    a._chk(a.lib.tfhe_amd_memcpy_h2d(a.ctx, d.ptr, x.ctypes.data, 4096))
elif mode == "free":
    d = b.alloc(4096)
    a.lib.tfhe_amd_free(a.ctx, d.ptr)
elif mode == "ok":        # the same calls with matching devices
    d_in, d_out = b.to_device(x), b.alloc(8192)
    b._chk(b.lib.tfhe_amd_ifft_int32(b.ctx, d_out.ptr, d_in.ptr, 1))
print("survived")
"""


@pytest.mark.parametrize("mode,dies", [("launch", True), ("copy", True), ("free", True), ("ok", False)])
def test_emulator_aborts_on_cross_device_operands(emu_lib, mode, dies):
    """the checks themselves: a buffer of device 2 used through a context of device 0 kills the process with a message
    naming the operand; the same calls on matching devices run"""
    out = subprocess.run([sys.executable, "-c", MISUSE % {"root": ROOT, "lib": emu_lib}, mode], capture_output=True, text=True,
                         timeout=300, env=dict(os.environ, TFHE_EMU_DEVICES="4"))
    if dies:
        assert out.returncode == -6 and "survived" not in out.stdout, (out.returncode, out.stdout, out.stderr[-500:])
        assert "belongs to device 2, the calling thread's current device is 0" in out.stderr
    else:
        assert out.returncode == 0 and "survived" in out.stdout, out.stderr[-500:]
