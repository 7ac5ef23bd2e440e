"""ctypes access to the CPU ORACLE (oracle/liboracle.so) and to the compiled reference
(oracle/_ref/ref_driver).  Test infrastructure only: nothing under
experimental-tfhe_amd/ imports this module.

The oracle restates /root/reference (see oracle/tfhe_oracle.h for file:line citations);
`ref()` runs the reference's own object code on binary files and exists only where
oracle/_ref/ref_driver was built (in the build container, and on the GPU box because the
binary travels with the snapshot)."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")
REF_DRIVER = os.path.join(ORACLE_DIR, "_ref", "ref_driver")

_lib = None


def build_oracle():
    """Compile oracle/liboracle.so if missing or stale (gcc only, a second or two)."""
    src = os.path.join(ORACLE_DIR, "tfhe_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "tfhe_oracle.h")
    if (not os.path.exists(LIB_PATH)
            or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def have_ref():
    return os.path.exists(REF_DRIVER) and os.access(REF_DRIVER, os.X_OK)


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = C.CDLL(LIB_PATH)
        L = _lib
        L.orc_tables_new.restype = C.c_void_p
        L.orc_tables_new.argtypes = [C.c_int]
        L.orc_tables_free.argtypes = [C.c_void_p]
        L.orc_tables_ifft_trig.restype = C.POINTER(C.c_double)
        L.orc_tables_ifft_trig.argtypes = [C.c_void_p]
        L.orc_tables_fft_trig.restype = C.POINTER(C.c_double)
        L.orc_tables_fft_trig.argtypes = [C.c_void_p]
        L.orc_tables_len.restype = C.c_int
        L.orc_tables_len.argtypes = [C.c_void_p]
        L.orc_modswitch32.restype = C.c_int32
        L.orc_modswitch32.argtypes = [C.c_int32, C.c_int]
        L.orc_lwe_phase32.restype = C.c_int32
        L.orc_lwe_phase64.restype = C.c_int64
        L.orc_rng_next.restype = C.c_uint64
        L.orc_rng_gauss.restype = C.c_double
        L.orc_cb_bootstrap_woks64_poc_quirks.restype = C.c_int
    return _lib


class RNG(C.Structure):
    _fields_ = [("s", C.c_uint64)]


_tables = {}


def tables(N):
    if N not in _tables:
        t = lib().orc_tables_new(N)
        assert t, "bad N"
        _tables[N] = C.c_void_p(t)
    return _tables[N]


def table_arrays(N):
    t = tables(N)
    n = lib().orc_tables_len(t)
    f = np.ctypeslib.as_array(lib().orc_tables_fft_trig(t), shape=(n,)).copy()
    r = np.ctypeslib.as_array(lib().orc_tables_ifft_trig(t), shape=(n,)).copy()
    return f, r


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ----------------------------------------------------------------- FFT level
def ifft(N, data):
    d = f64(data).copy()
    for row in d.reshape(-1, N):
        lib().orc_ifft(tables(N), _p(row, C.c_double))
    return d


def fft(N, data):
    d = f64(data).copy()
    for row in d.reshape(-1, N):
        lib().orc_fft(tables(N), _p(row, C.c_double))
    return d


def _batch(fn, N, a, in_t, out_dtype):
    a = np.ascontiguousarray(a)
    out = np.empty(a.shape, dtype=out_dtype)
    ar, outr = a.reshape(-1, N), out.reshape(-1, N)
    out_ct = {np.float64: C.c_double, np.int32: C.c_int32, np.int64: C.c_int64}[out_dtype]
    for i in range(ar.shape[0]):
        fn(tables(N), _p(outr[i], out_ct), _p(ar[i], in_t))
    return out


def execute_reverse_int(N, a):
    return _batch(lib().orc_execute_reverse_int, N, i32(a), C.c_int32, np.float64)


def execute_reverse_torus64(N, a):
    return _batch(lib().orc_execute_reverse_torus64, N, i64(a), C.c_int64, np.float64)


def execute_direct_torus32(N, a):
    return _batch(lib().orc_execute_direct_torus32, N, f64(a), C.c_double, np.int32)


def execute_direct_torus64(N, a):
    return _batch(lib().orc_execute_direct_torus64, N, f64(a), C.c_double, np.int64)


def lagrange_addmul(N, res, a, b):
    r = f64(res).copy()
    a, b = f64(a), f64(b)
    rr, ar, br = r.reshape(-1, N), a.reshape(-1, N), b.reshape(-1, N)
    for i in range(rr.shape[0]):
        lib().orc_lagrange_addmul(_p(rr[i], C.c_double), _p(ar[i], C.c_double), _p(br[i], C.c_double),
                                  C.c_long(N // 2))
    return r


def negacyclic_mul32(ipoly, tpoly):
    ipoly, tpoly = i32(ipoly), i32(tpoly)
    N = ipoly.shape[-1]
    out = np.empty(N, np.int32)
    lib().orc_negacyclic_mul32(_p(out, C.c_int32), _p(ipoly, C.c_int32), _p(tpoly, C.c_int32), N)
    return out


def negacyclic_mul64(ipoly, tpoly):
    ipoly, tpoly = i32(ipoly), i64(tpoly)
    N = ipoly.shape[-1]
    out = np.empty(N, np.int64)
    lib().orc_negacyclic_mul64(_p(out, C.c_int64), _p(ipoly, C.c_int32), _p(tpoly, C.c_int64), N)
    return out


# ------------------------------------------------------------- ring / TGSW
def decomp32(a, l, Bgbit):
    a = i32(a)
    N = a.shape[-1]
    out = np.empty((l, N), np.int32)
    lib().orc_decomp32(_p(out, C.c_int32), _p(a, C.c_int32), N, l, Bgbit)
    return out


def decomp64(a, l, Bgbit):
    a = i64(a)
    N = a.shape[-1]
    out = np.empty((l, N), np.int32)
    lib().orc_decomp64(_p(out, C.c_int32), _p(a, C.c_int64), N, l, Bgbit)
    return out


def mul_xai_minus_one32(a, poly):
    poly = i32(poly)
    out = np.empty_like(poly)
    lib().orc_mul_xai_minus_one32(_p(out, C.c_int32), int(a), _p(poly, C.c_int32), poly.shape[-1])
    return out


def mul_xai32(a, poly):
    poly = i32(poly)
    out = np.empty_like(poly)
    lib().orc_mul_xai32(_p(out, C.c_int32), int(a), _p(poly, C.c_int32), poly.shape[-1])
    return out


def mul_xai_minus_one64(a, poly):
    poly = i64(poly)
    out = np.empty_like(poly)
    lib().orc_mul_xai_minus_one64(_p(out, C.c_int64), int(a), _p(poly, C.c_int64), poly.shape[-1])
    return out


def mul_xai64(a, poly):
    poly = i64(poly)
    out = np.empty_like(poly)
    lib().orc_mul_xai64(_p(out, C.c_int64), int(a), _p(poly, C.c_int64), poly.shape[-1])
    return out


def extprod32(N, acc, gsw, l, Bgbit):
    acc = i32(acc).copy()
    gsw = f64(gsw)
    lib().orc_extprod32(tables(N), _p(acc, C.c_int32), _p(gsw, C.c_double), l, Bgbit)
    return acc


def hp_twiddles(n):
    """Real96 twiddles of the oracle (libquadmath): (powomega, powombar), each [n][4] uint64 = (re.lo, re.hi, im.lo, im.hi)"""
    a, b = np.empty((n, 4), np.uint64), np.empty((n, 4), np.uint64)
    lib().orc_hp_twiddles(n, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
    return a, b


def hp_ifft(x, powomega):
    x = i64(x)
    N = x.size
    out = np.empty((N // 2, 4), np.uint64)
    pw = np.ascontiguousarray(powomega, np.uint64)
    lib().orc_hp_ifft(out.ctypes.data_as(C.c_void_p), _p(x, C.c_int64), N, pw.ctypes.data_as(C.c_void_p))
    return out


def hp_fft(spec, powombar):
    spec = np.ascontiguousarray(spec, np.uint64).copy()
    N = spec.shape[0] * 2
    out = np.empty(N, np.int64)
    pw = np.ascontiguousarray(powombar, np.uint64)
    lib().orc_hp_fft(_p(out, C.c_int64), spec.ctypes.data_as(C.c_void_p), N, pw.ctypes.data_as(C.c_void_p))
    return out


def extprod_exact32(N, acc, gsw_torus, l, Bgbit):
    acc, g = i32(acc).copy(), i32(gsw_torus)
    lib().orc_extprod_exact32(_p(acc, C.c_int32), _p(g, C.c_int32), N, l, Bgbit)
    return acc


def extprod_exact64(N, acc, gsw_torus, l, Bgbit):
    acc, g = i64(acc).copy(), i64(gsw_torus)
    lib().orc_extprod_exact64(_p(acc, C.c_int64), _p(g, C.c_int64), N, l, Bgbit)
    return acc


def cmux32(N, gsw, d0, d1, l, Bgbit):
    out = np.empty(2 * N, np.int32)
    d0, d1, gsw = i32(d0), i32(d1), f64(gsw)
    lib().orc_cmux32(tables(N), _p(out, C.c_int32), _p(gsw, C.c_double), _p(d0, C.c_int32), _p(d1, C.c_int32), l, Bgbit)
    return out


def lut_eval32(N, bits, d, lut, l, Bgbit):
    """bits [d][2l][2][N] Lagrange TGSW samples of one item, lut [max(1,2^(d-logN))][N] -> LWE [N+1]"""
    out = np.empty(N + 1, np.int32)
    bits, lut = f64(bits), i32(lut)
    lib().orc_lut_eval32(tables(N), _p(out, C.c_int32), _p(bits, C.c_double), d, _p(lut, C.c_int32), l, Bgbit)
    return out


def extprod64(N, acc, gsw, l, Bgbit):
    acc = i64(acc).copy()
    gsw = f64(gsw)
    lib().orc_extprod64(tables(N), _p(acc, C.c_int64), _p(gsw, C.c_double), l, Bgbit)
    return acc


def mux_rotate32(N, acc, bki, barai, l, Bgbit):
    acc = i32(acc)
    out = np.empty_like(acc)
    lib().orc_mux_rotate32(tables(N), _p(out, C.c_int32), _p(acc, C.c_int32), _p(f64(bki), C.c_double),
                           int(barai), l, Bgbit)
    return out


def blind_rotate32(N, acc, bkfft, bara, l, Bgbit):
    acc = i32(acc).copy()
    bara = i32(bara)
    lib().orc_blind_rotate32(tables(N), _p(acc, C.c_int32), _p(f64(bkfft), C.c_double), _p(bara, C.c_int32),
                             len(bara), l, Bgbit)
    return acc


def blind_rotate64(N, acc, bkfft, bara, l, Bgbit):
    acc = i64(acc).copy()
    bara = i32(bara)
    lib().orc_blind_rotate64(tables(N), _p(acc, C.c_int64), _p(f64(bkfft), C.c_double), _p(bara, C.c_int32),
                             len(bara), l, Bgbit)
    return acc


def sample_extract32(acc):
    acc = i32(acc)
    N = acc.size // 2
    out = np.empty(N + 1, np.int32)
    lib().orc_sample_extract32(_p(out, C.c_int32), _p(acc, C.c_int32), N)
    return out


def blind_rotate_extract32(N, v, bkfft, barb, bara, l, Bgbit):
    out = np.empty(N + 1, np.int32)
    bara = i32(bara)
    lib().orc_blind_rotate_extract32(tables(N), _p(out, C.c_int32), _p(i32(v), C.c_int32),
                                     _p(f64(bkfft), C.c_double), int(barb), _p(bara, C.c_int32), len(bara),
                                     l, Bgbit)
    return out


def modswitch32(x, Msize):
    x = i32(x)
    return np.array([lib().orc_modswitch32(int(v), Msize) for v in x.ravel()], np.int32).reshape(x.shape)


def bootstrap_woks32(N, bkfft, mu, x, l, Bgbit):
    x = i32(x)
    n = x.size - 1
    out = np.empty(N + 1, np.int32)
    lib().orc_bootstrap_woks32(tables(N), _p(out, C.c_int32), _p(f64(bkfft), C.c_double), C.c_int32(mu),
                               _p(x, C.c_int32), n, l, Bgbit)
    return out


def keyswitch32(ks, x, n_in, n_out, t, basebit):
    x = i32(x)
    ks = i32(ks)
    out = np.empty(n_out + 1, np.int32)
    lib().orc_keyswitch32(_p(out, C.c_int32), _p(ks, C.c_int32), _p(x, C.c_int32), n_in, n_out, t, basebit)
    return out


def bootstrap32(N, bkfft, ks, mu, x, l, Bgbit, ks_t, ks_basebit):
    x = i32(x)
    n = x.size - 1
    out = np.empty(n + 1, np.int32)
    lib().orc_bootstrap32(tables(N), _p(out, C.c_int32), _p(f64(bkfft), C.c_double), _p(i32(ks), C.c_int32),
                          C.c_int32(mu), _p(x, C.c_int32), n, l, Bgbit, ks_t, ks_basebit)
    return out


# --------------------------------------------------------- circuit bootstrap
def pre_modswitch(x, N2):
    x = i32(x)
    out = np.empty_like(x)
    lib().orc_pre_modswitch(_p(out, C.c_int32), _p(x, C.c_int32), x.size - 1, N2)
    return out


def cb_bootstrap_woks64(N2, mu, abar, bkfft, l, Bgbit):
    abar = i32(abar)
    out = np.empty(N2 + 1, np.int64)
    lib().orc_cb_bootstrap_woks64(tables(N2), _p(out, C.c_int64), C.c_int64(mu), _p(abar, C.c_int32),
                                  _p(f64(bkfft), C.c_double), abar.size - 1, l, Bgbit)
    return out


def cb_bootstrap_woks64_poc_quirks(N2, mu, abar, bkfft0, l, Bgbit):
    abar = i32(abar)
    out = np.empty(N2 + 1, np.int64)
    rc = lib().orc_cb_bootstrap_woks64_poc_quirks(tables(N2), _p(out, C.c_int64), C.c_int64(mu),
                                                  _p(abar, C.c_int32), _p(f64(bkfft0), C.c_double),
                                                  abar.size - 1, l, Bgbit)
    assert rc == 0
    return out


def privks(table_u, x, n2, N1, t, basebit):
    x = i64(x)
    out = np.empty(2 * N1, np.int32)
    lib().orc_privks(_p(out, C.c_int32), _p(i32(table_u), C.c_int32), _p(x, C.c_int64), n2, N1, t, basebit)
    return out


def circuit_bootstrap(x, preks, bkfft, privks_tab, n0, N1, N2, l1, Bgbit1, l2, Bgbit2, t10, bb10, t21, bb21):
    out = np.empty((2, l1, 2, N1), np.int32)
    lib().orc_circuit_bootstrap(tables(N2), _p(out, C.c_int32), _p(i32(x), C.c_int32), _p(i32(preks), C.c_int32),
                                _p(f64(bkfft), C.c_double), _p(i32(privks_tab), C.c_int32), n0, N1, N2, l1,
                                Bgbit1, l2, Bgbit2, t10, bb10, t21, bb21)
    return out


# ------------------------------------------------------------------ harness
def keygen_binary(n, seed, stream):
    key = np.empty(n, np.int32)
    lib().orc_keygen_binary(_p(key, C.c_int32), n, C.c_uint64(seed), C.c_uint64(stream))
    return key


def rng(seed, stream):
    r = RNG()
    lib().orc_rng_init(C.byref(r), C.c_uint64(seed), C.c_uint64(stream))
    return r


def lwe_encrypt32(mess, stdev, key, r):
    key = i32(key)
    ct = np.empty(key.size + 1, np.int32)
    lib().orc_lwe_encrypt32(_p(ct, C.c_int32), C.c_int32(mess), C.c_double(stdev), _p(key, C.c_int32),
                            key.size, C.byref(r))
    return ct


def lwe_phase32(ct, key):
    ct, key = i32(ct), i32(key)
    return int(lib().orc_lwe_phase32(_p(ct, C.c_int32), _p(key, C.c_int32), key.size))


def lwe_phase64(ct, key):
    ct, key = i64(ct), i32(key)
    return int(lib().orc_lwe_phase64(_p(ct, C.c_int64), _p(key, C.c_int32), key.size))


def bk_create32(N, lwe_key, tkey, l, Bgbit, stdev, seed, stream):
    lwe_key, tkey = i32(lwe_key), i32(tkey)
    bk = np.empty((lwe_key.size, 2 * l, 2, N), np.float64)
    lib().orc_bk_create32(tables(N), _p(bk, C.c_double), _p(lwe_key, C.c_int32), lwe_key.size,
                          _p(tkey, C.c_int32), l, Bgbit, C.c_double(stdev), C.c_uint64(seed), C.c_uint64(stream))
    return bk


def bk_create64(N, lwe_key, tkey, l, Bgbit, stdev, seed, stream):
    lwe_key, tkey = i32(lwe_key), i32(tkey)
    bk = np.empty((lwe_key.size, 2 * l, 2, N), np.float64)
    lib().orc_bk_create64(tables(N), _p(bk, C.c_double), _p(lwe_key, C.c_int32), lwe_key.size,
                          _p(tkey, C.c_int32), l, Bgbit, C.c_double(stdev), C.c_uint64(seed), C.c_uint64(stream))
    return bk


def ks_create32(in_key, out_key, t, basebit, stdev, seed, stream):
    in_key, out_key = i32(in_key), i32(out_key)
    ks = np.empty((in_key.size, t, 1 << basebit, out_key.size + 1), np.int32)
    lib().orc_ks_create32(_p(ks, C.c_int32), _p(in_key, C.c_int32), in_key.size, _p(out_key, C.c_int32),
                          out_key.size, t, basebit, C.c_double(stdev), C.c_uint64(seed), C.c_uint64(stream))
    return ks


def tlwe_phase32(ct, tkey):
    ct, tkey = i32(ct), i32(tkey)
    N = tkey.size
    out = np.empty(N, np.int32)
    lib().orc_tlwe_phase32(_p(out, C.c_int32), _p(ct, C.c_int32), _p(tkey, C.c_int32), N)
    return out


def tlwe_phase64(ct, tkey):
    ct, tkey = i64(ct), i32(tkey)
    N = tkey.size
    out = np.empty(N, np.int64)
    lib().orc_tlwe_phase64(_p(out, C.c_int64), _p(ct, C.c_int64), _p(tkey, C.c_int32), N)
    return out


def privks_create(key2, tkey1, t, basebit, stdev, seed, stream):
    key2, tkey1 = i32(key2), i32(tkey1)
    n2, N1 = key2.size, tkey1.size
    tab = np.empty((2, n2 + 1, t, 1 << basebit, 2, N1), np.int32)
    lib().orc_privks_create(_p(tab, C.c_int32), _p(key2, C.c_int32), n2, _p(tkey1, C.c_int32), N1, t, basebit,
                            C.c_double(stdev), C.c_uint64(seed), C.c_uint64(stream))
    return tab


# ------------------------------------------ splitmix64 in numpy (table filling)
_G = np.uint64(0x9E3779B97F4A7C15)


def splitmix64_stream(seed, count, start=0):
    """outputs start .. start+count-1 of splitmix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _G
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def fill32_numpy(seed, count, chunk=1 << 22):
    """count int32 values = high halves of consecutive splitmix64 outputs (ref_driver.cpp SplitMix)."""
    out = np.empty(count, np.int32)
    for s in range(0, count, chunk):
        c = min(chunk, count - s)
        out[s:s + c] = (splitmix64_stream(seed, c, s) >> np.uint64(32)).astype(np.uint32).view(np.int32)
    return out


def fill32(seed, count):
    """same values as fill32_numpy, produced by the oracle's C loop (fast path for big tables)."""
    out = np.empty(count, np.int32)
    lib().orc_fill32(_p(out, C.c_int32), C.c_uint64(seed), C.c_size_t(count))
    return out


# -------------------------------------------------- compiled reference runner
def ref(op, inp, out_dtype, *args):
    """Run oracle/_ref/ref_driver <op> on the raw bytes of `inp`; returns a flat array."""
    assert have_ref(), "oracle/_ref/ref_driver not built (make -C oracle ref)"
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        if isinstance(inp, (bytes, bytearray)):
            with open(fi, "wb") as f:
                f.write(inp)
        else:
            np.ascontiguousarray(inp).tofile(fi)
        subprocess.check_call([REF_DRIVER, op, fi, fo] + [str(a) for a in args])
        return np.fromfile(fo, dtype=out_dtype)
