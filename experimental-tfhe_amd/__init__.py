"""experimental-tfhe_amd: MI355X-native TFHE bootstrapping engine (thin ctypes binding).

The product is the HIP shared library `libtfhe_amd.so` (built in-tree by build.py from
csrc/) behind the C ABI of include/tfhe_amd.h.  This module only loads it and wraps the
calls; it holds no arithmetic and has NO CPU fallback: if the library is missing, or no
GPU is present, construction raises.

The directory name contains a hyphen, so import it with
    importlib.import_module("experimental-tfhe_amd")
(the repo root on sys.path), as tests/conftest.py, bench.py and __graft_entry__.py do.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libtfhe_amd.so")

OK, ERR_PARAM, ERR_DEVICE, ERR_STATE, ERR_ALLOC = range(5)
OPT_KS_GATHER, OPT_STREAMED_GRAPH, OPT_BR_SPLIT = 2, 4, 5
POOL_OPT_CHUNK_ROWS = 1

# every symbol include/tfhe_amd.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "tfhe_amd_ctx_create", "tfhe_amd_ctx_destroy", "tfhe_amd_last_error", "tfhe_amd_version",
    "tfhe_amd_set_stream", "tfhe_amd_sync", "tfhe_amd_set_option", "tfhe_amd_get_tables",
    "tfhe_amd_event_create", "tfhe_amd_event_record", "tfhe_amd_event_elapsed_ms", "tfhe_amd_event_destroy",
    "tfhe_amd_malloc", "tfhe_amd_free", "tfhe_amd_memcpy_h2d", "tfhe_amd_memcpy_d2h", "tfhe_amd_host_alloc", "tfhe_amd_host_free",
    "tfhe_amd_gsw_from_fft", "tfhe_amd_gsw_from_torus", "tfhe_amd_gsw_from_torus_d", "tfhe_amd_gsw_free", "tfhe_amd_gsw_export_fft",
    "tfhe_amd_set_bootstrap_key", "tfhe_amd_load_keyswitch_key",
    "tfhe_amd_ifft_int32", "tfhe_amd_ifft_torus64", "tfhe_amd_fft_torus32", "tfhe_amd_fft_torus64",
    "tfhe_amd_ifft_f64", "tfhe_amd_fft_f64", "tfhe_amd_build_tables",
    "tfhe_amd_lagrange_addmul", "tfhe_amd_extern_mul", "tfhe_amd_device_info", "tfhe_amd_hp_twiddles", "tfhe_amd_hp_ifft", "tfhe_amd_hp_fft",
    "tfhe_amd_mux_rotate", "tfhe_amd_extern_mul_exact", "tfhe_amd_cmux", "tfhe_amd_lut_eval",
    "tfhe_amd_blind_rotate", "tfhe_amd_blind_rotate_extract", "tfhe_amd_bootstrap_woks",
    "tfhe_amd_keyswitch", "tfhe_amd_bootstrap", "tfhe_amd_bootstrap_streamed", "tfhe_amd_bootstrap_host",
    "tfhe_amd_cb_bootstrap_woks", "tfhe_amd_modswitch",
    "tfhe_amd_cb_create", "tfhe_amd_cb_destroy", "tfhe_amd_cb_last_error", "tfhe_amd_cb_set_stream",
    "tfhe_amd_cb_sync", "tfhe_amd_cb_ctx_lvl10", "tfhe_amd_cb_ctx_lvl2", "tfhe_amd_cb_load_preks",
    "tfhe_amd_cb_load_bk_torus", "tfhe_amd_cb_load_bk_fft", "tfhe_amd_cb_load_privks_plane",
    "tfhe_amd_privks", "tfhe_amd_circuit_bootstrap",
    "tfhe_amd_keygen_binary", "tfhe_amd_lwe_encrypt32", "tfhe_amd_lwe_phase32",
    "tfhe_amd_keygen_bk_torus32", "tfhe_amd_keygen_bk_torus64", "tfhe_amd_keygen_ks32",
    "tfhe_amd_device_count", "tfhe_amd_device_pci_bus_id", "tfhe_amd_clock_probe",
    "tfhe_amd_memcpy_h2d_async", "tfhe_amd_memcpy_d2h_async", "tfhe_amd_stream_create", "tfhe_amd_stream_sync", "tfhe_amd_stream_destroy",
    "tfhe_amd_event_sync", "tfhe_amd_stream_wait_event",
    "tfhe_amd_load_keyswitch_key_d", "tfhe_amd_keyswitch_key_bytes", "tfhe_amd_keyswitch_key_export",
    "tfhe_amd_gsw_packed_bytes", "tfhe_amd_gsw_export_packed", "tfhe_amd_gsw_from_packed",
    "tfhe_amd_pool_create", "tfhe_amd_pool_destroy", "tfhe_amd_pool_last_error", "tfhe_amd_pool_size", "tfhe_amd_pool_device",
    "tfhe_amd_pool_ctx", "tfhe_amd_pool_load_keys", "tfhe_amd_pool_load_keys_torus", "tfhe_amd_pool_bootstrap_host",
    "tfhe_amd_pool_bootstrap_woks_host", "tfhe_amd_pool_keyswitch_host", "tfhe_amd_pool_last_split", "tfhe_amd_pool_set_option",
    "tfhe_amd_pool_bootstrap_rows", "tfhe_amd_pool_bootstrap_woks_rows", "tfhe_amd_pool_keyswitch_rows",
    "tfhe_amd_cb_pool_create", "tfhe_amd_cb_pool_destroy", "tfhe_amd_cb_pool_last_error", "tfhe_amd_cb_pool_size",
    "tfhe_amd_cb_pool_member", "tfhe_amd_cb_pool_load_preks", "tfhe_amd_cb_pool_load_bk_fft", "tfhe_amd_cb_pool_load_bk_torus",
    "tfhe_amd_cb_pool_load_privks_plane", "tfhe_amd_cb_pool_circuit_bootstrap_host", "tfhe_amd_cb_pool_circuit_bootstrap_rows",
    "tfhe_amd_cb_pool_set_option",
]


class Params(C.Structure):
    """mirror of tfhe_amd_params (include/tfhe_amd.h)"""
    _fields_ = [("torus_bits", C.c_int32), ("n", C.c_int32), ("N", C.c_int32), ("k", C.c_int32),
                ("l", C.c_int32), ("Bgbit", C.c_int32), ("ks_t", C.c_int32), ("ks_basebit", C.c_int32),
                ("ks_n_out", C.c_int32)]


class CbParams(C.Structure):
    """mirror of tfhe_amd_cb_params (include/tfhe_amd.h)"""
    _fields_ = [(k, C.c_int32) for k in ("n0", "N1", "N2", "l1", "Bgbit1", "l2", "Bgbit2", "t10", "bb10", "t21", "bb21")]


class TfheAmdError(RuntimeError):
    pass


_libs = {}


def load_library(path=None):
    """dlopen the HIP library.  Fails loudly when it has not been built."""
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise TfheAmdError(
            f"{path} not found: build it with `python experimental-tfhe_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32p, i64p, f64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    lib.tfhe_amd_ctx_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(vp)]
    lib.tfhe_amd_ctx_destroy.argtypes = [vp]
    lib.tfhe_amd_ctx_destroy.restype = None
    lib.tfhe_amd_last_error.argtypes = [vp]
    lib.tfhe_amd_last_error.restype = C.c_char_p
    lib.tfhe_amd_version.restype = C.c_char_p
    lib.tfhe_amd_set_stream.argtypes = [vp, vp]
    lib.tfhe_amd_sync.argtypes = [vp]
    lib.tfhe_amd_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.tfhe_amd_get_tables.argtypes = [vp, f64p, f64p]
    lib.tfhe_amd_event_create.argtypes = [vp, C.POINTER(vp)]
    lib.tfhe_amd_event_record.argtypes = [vp, vp]
    lib.tfhe_amd_event_elapsed_ms.argtypes = [vp, vp, vp, C.POINTER(C.c_float)]
    lib.tfhe_amd_event_destroy.argtypes = [vp, vp]
    lib.tfhe_amd_malloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    lib.tfhe_amd_free.argtypes = [vp, vp]
    lib.tfhe_amd_host_alloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    lib.tfhe_amd_host_free.argtypes = [vp, vp]
    lib.tfhe_amd_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.tfhe_amd_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.tfhe_amd_gsw_from_fft.argtypes = [vp, f64p, C.c_int, C.POINTER(vp)]
    lib.tfhe_amd_gsw_from_torus.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    lib.tfhe_amd_gsw_from_torus_d.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    lib.tfhe_amd_gsw_free.argtypes = [vp]
    lib.tfhe_amd_gsw_free.restype = None
    lib.tfhe_amd_gsw_export_fft.argtypes = [vp, vp, C.c_int, f64p]
    lib.tfhe_amd_set_bootstrap_key.argtypes = [vp, vp]
    lib.tfhe_amd_load_keyswitch_key.argtypes = [vp, i32p]
    lib.tfhe_amd_ifft_int32.argtypes = [vp, f64p, i32p, C.c_int]
    lib.tfhe_amd_ifft_torus64.argtypes = [vp, f64p, i64p, C.c_int]
    lib.tfhe_amd_fft_torus32.argtypes = [vp, i32p, f64p, C.c_int]
    lib.tfhe_amd_fft_torus64.argtypes = [vp, i64p, f64p, C.c_int]
    lib.tfhe_amd_ifft_f64.argtypes = [vp, f64p, f64p, C.c_int]
    lib.tfhe_amd_fft_f64.argtypes = [vp, f64p, f64p, C.c_int]
    lib.tfhe_amd_build_tables.argtypes = [C.c_int, f64p, f64p]
    lib.tfhe_amd_lagrange_addmul.argtypes = [vp, f64p, f64p, f64p, C.c_int, C.c_int]
    lib.tfhe_amd_extern_mul.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    lib.tfhe_amd_mux_rotate.argtypes = [vp, vp, vp, C.c_int, i32p, C.c_int]
    lib.tfhe_amd_cmux.argtypes = [vp, vp, vp, i32p, vp, vp, C.c_int]
    lib.tfhe_amd_extern_mul_exact.argtypes = [vp, vp, vp, C.c_int]
    lib.tfhe_amd_hp_twiddles.argtypes = [C.c_int, vp, vp]
    lib.tfhe_amd_hp_ifft.argtypes = [vp, vp, vp, C.c_int]
    lib.tfhe_amd_hp_fft.argtypes = [vp, vp, vp, C.c_int]
    lib.tfhe_amd_lut_eval.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int]
    lib.tfhe_amd_blind_rotate.argtypes = [vp, vp, i32p, C.c_int]
    lib.tfhe_amd_blind_rotate_extract.argtypes = [vp, vp, vp, C.c_int, i32p, C.c_int]
    lib.tfhe_amd_bootstrap_woks.argtypes = [vp, i32p, C.c_int32, i32p, C.c_int]
    lib.tfhe_amd_keyswitch.argtypes = [vp, i32p, i32p, C.c_int]
    lib.tfhe_amd_bootstrap.argtypes = [vp, i32p, C.c_int32, i32p, C.c_int]
    lib.tfhe_amd_bootstrap_streamed.argtypes = [vp, i32p, C.c_int32, i32p, C.c_int]
    lib.tfhe_amd_bootstrap_host.argtypes = [vp, i32p, C.c_int32, i32p, C.c_int]
    lib.tfhe_amd_cb_bootstrap_woks.argtypes = [vp, i64p, C.c_int64, i32p, C.c_int]
    lib.tfhe_amd_modswitch.argtypes = [vp, i32p, i32p, C.c_int]
    lib.tfhe_amd_cb_create.argtypes = [C.POINTER(CbParams), C.c_int, C.POINTER(vp)]
    lib.tfhe_amd_cb_destroy.argtypes = [vp]
    lib.tfhe_amd_cb_destroy.restype = None
    lib.tfhe_amd_cb_last_error.argtypes = [vp]
    lib.tfhe_amd_cb_last_error.restype = C.c_char_p
    lib.tfhe_amd_cb_set_stream.argtypes = [vp, vp]
    lib.tfhe_amd_cb_sync.argtypes = [vp]
    lib.tfhe_amd_cb_ctx_lvl10.argtypes = [vp]
    lib.tfhe_amd_cb_ctx_lvl10.restype = vp
    lib.tfhe_amd_cb_ctx_lvl2.argtypes = [vp]
    lib.tfhe_amd_cb_ctx_lvl2.restype = vp
    lib.tfhe_amd_cb_load_preks.argtypes = [vp, i32p]
    lib.tfhe_amd_cb_load_bk_torus.argtypes = [vp, i64p]
    lib.tfhe_amd_cb_load_bk_fft.argtypes = [vp, f64p]
    lib.tfhe_amd_cb_load_privks_plane.argtypes = [vp, C.c_int, i32p]
    lib.tfhe_amd_privks.argtypes = [vp, i32p, C.c_int, i64p, C.c_int]
    lib.tfhe_amd_circuit_bootstrap.argtypes = [vp, i32p, i32p, C.c_int]
    lib.tfhe_amd_keygen_binary.argtypes = [i32p, C.c_int, C.c_uint64, C.c_uint64]
    lib.tfhe_amd_lwe_encrypt32.argtypes = [i32p, C.c_int32, C.c_double, i32p, C.c_int, C.c_uint64, C.c_uint64]
    lib.tfhe_amd_lwe_phase32.argtypes = [i32p, i32p, C.c_int]
    lib.tfhe_amd_lwe_phase32.restype = C.c_int32
    lib.tfhe_amd_keygen_bk_torus32.argtypes = [i32p, i32p, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_double,
                                               C.c_uint64, C.c_uint64]
    lib.tfhe_amd_keygen_bk_torus64.argtypes = list(lib.tfhe_amd_keygen_bk_torus32.argtypes)
    lib.tfhe_amd_keygen_ks32.argtypes = [i32p, i32p, C.c_int, i32p, C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.c_uint64, C.c_uint64]
    lib.tfhe_amd_memcpy_h2d_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.tfhe_amd_memcpy_d2h_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.tfhe_amd_stream_create.argtypes = [vp, C.POINTER(vp)]
    lib.tfhe_amd_stream_sync.argtypes = [vp, vp]
    lib.tfhe_amd_stream_destroy.argtypes = [vp, vp]
    lib.tfhe_amd_event_sync.argtypes = [vp, vp]
    lib.tfhe_amd_stream_wait_event.argtypes = [vp, vp]
    lib.tfhe_amd_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    lib.tfhe_amd_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.tfhe_amd_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    lib.tfhe_amd_clock_probe.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.tfhe_amd_load_keyswitch_key_d.argtypes = [vp, vp]
    lib.tfhe_amd_keyswitch_key_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.tfhe_amd_keyswitch_key_export.argtypes = [vp, vp]
    lib.tfhe_amd_gsw_packed_bytes.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    lib.tfhe_amd_gsw_export_packed.argtypes = [vp, vp, vp]
    lib.tfhe_amd_gsw_from_packed.argtypes = [vp, vp, C.c_int, C.POINTER(vp)]
    ip = C.POINTER(C.c_int)
    lib.tfhe_amd_pool_create.argtypes = [C.POINTER(Params), ip, C.c_int, C.POINTER(vp)]
    lib.tfhe_amd_pool_destroy.argtypes = [vp]
    lib.tfhe_amd_pool_destroy.restype = None
    lib.tfhe_amd_pool_last_error.argtypes = [vp]
    lib.tfhe_amd_pool_last_error.restype = C.c_char_p
    lib.tfhe_amd_pool_size.argtypes = [vp]
    lib.tfhe_amd_pool_device.argtypes = [vp, C.c_int]
    lib.tfhe_amd_pool_ctx.argtypes = [vp, C.c_int]
    lib.tfhe_amd_pool_ctx.restype = vp
    lib.tfhe_amd_pool_load_keys.argtypes = [vp, f64p, i32p]
    lib.tfhe_amd_pool_load_keys_torus.argtypes = [vp, vp, i32p]
    lib.tfhe_amd_pool_bootstrap_host.argtypes = [vp, i32p, C.c_int32, i32p, C.c_int]
    lib.tfhe_amd_pool_bootstrap_woks_host.argtypes = [vp, i32p, C.c_int32, i32p, C.c_int]
    lib.tfhe_amd_pool_keyswitch_host.argtypes = [vp, i32p, i32p, C.c_int]
    lib.tfhe_amd_pool_last_split.argtypes = [vp, ip, C.POINTER(C.c_double)]
    lib.tfhe_amd_pool_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.tfhe_amd_cb_pool_create.argtypes = [C.POINTER(CbParams), ip, C.c_int, C.POINTER(vp)]
    lib.tfhe_amd_cb_pool_destroy.argtypes = [vp]
    lib.tfhe_amd_cb_pool_destroy.restype = None
    lib.tfhe_amd_cb_pool_last_error.argtypes = [vp]
    lib.tfhe_amd_cb_pool_last_error.restype = C.c_char_p
    lib.tfhe_amd_cb_pool_size.argtypes = [vp]
    lib.tfhe_amd_cb_pool_member.argtypes = [vp, C.c_int]
    lib.tfhe_amd_cb_pool_member.restype = vp
    lib.tfhe_amd_cb_pool_load_preks.argtypes = [vp, i32p]
    lib.tfhe_amd_cb_pool_load_bk_fft.argtypes = [vp, f64p]
    lib.tfhe_amd_cb_pool_load_bk_torus.argtypes = [vp, i64p]
    lib.tfhe_amd_cb_pool_load_privks_plane.argtypes = [vp, C.c_int, i32p]
    lib.tfhe_amd_cb_pool_circuit_bootstrap_host.argtypes = [vp, i32p, i32p, C.c_int]
    lib.tfhe_amd_cb_pool_set_option.argtypes = [vp, C.c_int, C.c_int]
    _libs[path] = lib
    return lib


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """a hipMalloc'd buffer owned through the C ABI (tfhe_amd_malloc / tfhe_amd_free)"""

    def __init__(self, eng, nbytes):
        self.eng, self.nbytes = eng, int(nbytes)
        p = C.c_void_p()
        eng._chk(eng.lib.tfhe_amd_malloc(eng.ctx, C.byref(p), self.nbytes))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.eng._chk(self.eng.lib.tfhe_amd_memcpy_h2d(self.eng.ctx, self.ptr, _np_ptr(arr), arr.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.eng._chk(self.eng.lib.tfhe_amd_memcpy_d2h(self.eng.ctx, _np_ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.eng.lib.tfhe_amd_free(self.eng.ctx, self.ptr)
            self.ptr = None


class Engine:
    """One context = one device + one stream + one parameter set (tfhe_amd_ctx)."""

    def __init__(self, torus_bits=32, n=630, N=1024, l=2, Bgbit=10, ks_t=8, ks_basebit=2, ks_n_out=None,
                 device=0, lib_path=None):
        self.lib = load_library(lib_path)
        self.params = Params(torus_bits, n, N, 1, l, Bgbit, ks_t, ks_basebit,
                             (n if ks_n_out is None else ks_n_out) if ks_t else 0)
        self.ctx = C.c_void_p()
        rc = self.lib.tfhe_amd_ctx_create(C.byref(self.params), device, C.byref(self.ctx))
        if rc != OK:
            raise TfheAmdError(f"tfhe_amd_ctx_create failed with status {rc} "
                               "(2 = no usable HIP device; this library has no CPU path)")
        self.torus = np.int32 if torus_bits == 32 else np.int64
        self._bufs = []
        self._gsw = []

    # -- plumbing
    def _chk(self, rc):
        if rc != OK:
            raise TfheAmdError(f"status {rc}: {self.lib.tfhe_amd_last_error(self.ctx).decode()}")

    def close(self):
        if self.ctx:
            for g in self._gsw:
                self.lib.tfhe_amd_gsw_free(g)
            for b in self._bufs:
                b.free()
            self.lib.tfhe_amd_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc(self, nbytes):
        b = DeviceBuffer(self, nbytes)
        self._bufs.append(b)
        return b

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return self.alloc(max(arr.nbytes, 1)).upload(arr)

    def set_option(self, option, value):
        """TFHE_AMD_OPT_KS_GATHER = 2, TFHE_AMD_OPT_STREAMED_GRAPH = 4 (include/tfhe_amd.h)"""
        self._chk(self.lib.tfhe_amd_set_option(self.ctx, option, value))

    def sync(self):
        self._chk(self.lib.tfhe_amd_sync(self.ctx))

    def event(self):
        e = C.c_void_p()
        self._chk(self.lib.tfhe_amd_event_create(self.ctx, C.byref(e)))
        return e

    def record(self, ev):
        self._chk(self.lib.tfhe_amd_event_record(self.ctx, ev))

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        self._chk(self.lib.tfhe_amd_event_elapsed_ms(self.ctx, start, stop, C.byref(ms)))
        return float(ms.value)

    def set_stream(self, stream_ptr):
        self._chk(self.lib.tfhe_amd_set_stream(self.ctx, stream_ptr))

    def tables(self):
        n = 2 * self.params.N - 8
        f, r = np.empty(n), np.empty(n)
        self._chk(self.lib.tfhe_amd_get_tables(self.ctx, _np_ptr(f), _np_ptr(r)))
        return f, r

    # -- keys
    def gsw_from_fft(self, host_fft):
        a = np.ascontiguousarray(host_fft, dtype=np.float64)
        count = a.size // (2 * self.params.l * 2 * self.params.N)
        g = C.c_void_p()
        self._chk(self.lib.tfhe_amd_gsw_from_fft(self.ctx, _np_ptr(a), count, C.byref(g)))
        self._gsw.append(g)
        return g

    def gsw_from_torus(self, host_torus):
        a = np.ascontiguousarray(host_torus, dtype=self.torus)
        count = a.size // (2 * self.params.l * 2 * self.params.N)
        g = C.c_void_p()
        self._chk(self.lib.tfhe_amd_gsw_from_torus(self.ctx, _np_ptr(a), count, C.byref(g)))
        self._gsw.append(g)
        return g

    def gsw_from_torus_d(self, dev_torus, count):
        """tGswToFFTConvert of `count` TGSW samples already in device memory (DeviceBuffer or pointer)"""
        g = C.c_void_p()
        self._chk(self.lib.tfhe_amd_gsw_from_torus_d(self.ctx, getattr(dev_torus, "ptr", dev_torus), count, C.byref(g)))
        self._gsw.append(g)
        return g

    def gsw_export_fft(self, g, index):
        out = np.empty((2 * self.params.l, 2, self.params.N))
        self._chk(self.lib.tfhe_amd_gsw_export_fft(self.ctx, g, index, _np_ptr(out)))
        return out

    def set_bootstrap_key(self, g):
        self._chk(self.lib.tfhe_amd_set_bootstrap_key(self.ctx, g))

    def load_keyswitch_key(self, ks):
        ks = np.ascontiguousarray(ks, dtype=np.int32)
        self._chk(self.lib.tfhe_amd_load_keyswitch_key(self.ctx, _np_ptr(ks)))

    # -- keys as the bytes of their device layout (replication: one copy per device, or one broadcast)
    def gsw_packed_bytes(self, count):
        n = C.c_size_t()
        self._chk(self.lib.tfhe_amd_gsw_packed_bytes(self.ctx, count, C.byref(n)))
        return int(n.value)

    def gsw_export_packed(self, g, dst_ptr):
        """writes the TGSW samples' kernel layout to `dst_ptr` (device memory of this context's device, or host)"""
        self._chk(self.lib.tfhe_amd_gsw_export_packed(self.ctx, g, dst_ptr))

    def gsw_from_packed(self, src_ptr, count):
        g = C.c_void_p()
        self._chk(self.lib.tfhe_amd_gsw_from_packed(self.ctx, src_ptr, count, C.byref(g)))
        self._gsw.append(g)
        return g

    def keyswitch_key_bytes(self):
        n = C.c_size_t()
        self._chk(self.lib.tfhe_amd_keyswitch_key_bytes(self.ctx, C.byref(n)))
        return int(n.value)

    def keyswitch_key_export(self, dst_ptr):
        self._chk(self.lib.tfhe_amd_keyswitch_key_export(self.ctx, dst_ptr))

    def load_keyswitch_key_d(self, src_ptr):
        self._chk(self.lib.tfhe_amd_load_keyswitch_key_d(self.ctx, src_ptr))

    def clock_probe(self, duration_us):
        """(median, min, max) shader clock in GHz over `duration_us`, measured BESIDE the work already queued on this
        context's stream (tfhe_amd_clock_probe)"""
        med, lo, hi = C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.tfhe_amd_clock_probe(self.ctx, int(duration_us), C.byref(med), C.byref(lo), C.byref(hi)))
        return float(med.value), float(lo.value), float(hi.value)

    # -- host-array convenience wrappers around the device-pointer ABI (tests, smoke)
    def _roundtrip(self, fn, inp, out_dtype, out_shape, *extra):
        d_in = self.to_device(inp)
        out_nbytes = int(np.prod(out_shape)) * np.dtype(out_dtype).itemsize
        d_out = self.alloc(max(out_nbytes, 1))
        self._chk(fn(self.ctx, d_out.ptr, d_in.ptr, *extra))
        res = d_out.download(out_dtype, out_shape)
        d_in.free()
        d_out.free()
        return res

    def ifft_int32(self, a):
        a = np.ascontiguousarray(a, np.int32).reshape(-1, self.params.N)
        return self._roundtrip(self.lib.tfhe_amd_ifft_int32, a, np.float64, a.shape, a.shape[0])

    def ifft_torus64(self, a):
        a = np.ascontiguousarray(a, np.int64).reshape(-1, self.params.N)
        return self._roundtrip(self.lib.tfhe_amd_ifft_torus64, a, np.float64, a.shape, a.shape[0])

    def fft_torus32(self, a):
        a = np.ascontiguousarray(a, np.float64).reshape(-1, self.params.N)
        return self._roundtrip(self.lib.tfhe_amd_fft_torus32, a, np.int32, a.shape, a.shape[0])

    def fft_torus64(self, a):
        a = np.ascontiguousarray(a, np.float64).reshape(-1, self.params.N)
        return self._roundtrip(self.lib.tfhe_amd_fft_torus64, a, np.int64, a.shape, a.shape[0])

    def ifft_f64(self, a):
        """the reference's bare `ifft` (spqlios-fft.h:53) on [batch][N] doubles"""
        a = np.ascontiguousarray(a, np.float64).reshape(-1, self.params.N)
        return self._roundtrip(self.lib.tfhe_amd_ifft_f64, a, np.float64, a.shape, a.shape[0])

    def fft_f64(self, a):
        """the reference's bare `fft` (spqlios-fft.h:52): no 2/N scale, no rounding"""
        a = np.ascontiguousarray(a, np.float64).reshape(-1, self.params.N)
        return self._roundtrip(self.lib.tfhe_amd_fft_f64, a, np.float64, a.shape, a.shape[0])

    def lagrange_addmul(self, res, a, b, b_shared=False):
        res = np.ascontiguousarray(res, np.float64).reshape(-1, self.params.N)
        d_r, d_a, d_b = self.to_device(res), self.to_device(np.asarray(a, np.float64)), self.to_device(
            np.asarray(b, np.float64))
        self._chk(self.lib.tfhe_amd_lagrange_addmul(self.ctx, d_r.ptr, d_a.ptr, d_b.ptr, res.shape[0], int(b_shared)))
        out = d_r.download(np.float64, res.shape)
        for d in (d_r, d_a, d_b):
            d.free()
        return out

    def extern_mul(self, acc, g, index=0):
        acc = np.ascontiguousarray(acc, self.torus).reshape(-1, 2, self.params.N)
        d = self.to_device(acc)
        self._chk(self.lib.tfhe_amd_extern_mul(self.ctx, d.ptr, g, index, acc.shape[0]))
        out = d.download(self.torus, acc.shape)
        d.free()
        return out

    def hp_ifft(self, x):
        """Real96 iFFT: [B][N] Torus64 -> [B][N/2][4] uint64 (re.lo, re.hi, im.lo, im.hi)"""
        N = self.params.N
        x = np.ascontiguousarray(x, np.int64).reshape(-1, N)
        return self._roundtrip(self.lib.tfhe_amd_hp_ifft, x, np.uint64, (x.shape[0], N // 2, 4), x.shape[0])

    def hp_fft(self, spec):
        """Real96 FFT: [B][N/2][4] uint64 -> [B][N] Torus64 (divided by N/2)"""
        N = self.params.N
        spec = np.ascontiguousarray(spec, np.uint64).reshape(-1, N // 2, 4)
        return self._roundtrip(self.lib.tfhe_amd_hp_fft, spec, np.int64, (spec.shape[0], N), spec.shape[0])

    def extern_mul_exact(self, acc, gsw_torus):
        """FFT-free external product: gsw_torus = one TGSW sample in coefficient form [2l][2][N]"""
        acc = np.ascontiguousarray(acc, self.torus).reshape(-1, 2, self.params.N)
        d, g = self.to_device(acc), self.to_device(np.ascontiguousarray(gsw_torus, self.torus))
        self._chk(self.lib.tfhe_amd_extern_mul_exact(self.ctx, d.ptr, g.ptr, acc.shape[0]))
        out = d.download(self.torus, acc.shape)
        d.free()
        g.free()
        return out

    def mux_rotate(self, acc, g, index, barai):
        acc = np.ascontiguousarray(acc, self.torus).reshape(-1, 2, self.params.N)
        d, r = self.to_device(acc), self.to_device(np.ascontiguousarray(barai, np.int32))
        self._chk(self.lib.tfhe_amd_mux_rotate(self.ctx, d.ptr, g, index, r.ptr, acc.shape[0]))
        out = d.download(self.torus, acc.shape)
        d.free()
        r.free()
        return out

    def cmux(self, g, sel, d0, d1):
        d0 = np.ascontiguousarray(d0, self.torus).reshape(-1, 2, self.params.N)
        d1 = np.ascontiguousarray(d1, self.torus).reshape(d0.shape)
        a0, a1 = self.to_device(d0), self.to_device(d1)
        out = self.alloc(d0.nbytes)
        s = self.to_device(np.ascontiguousarray(sel, np.int32)) if sel is not None else None
        self._chk(self.lib.tfhe_amd_cmux(self.ctx, out.ptr, g, s.ptr if s else None, a0.ptr, a1.ptr, d0.shape[0]))
        res = out.download(self.torus, d0.shape)
        for d in (a0, a1, out) + ((s,) if s else ()):
            d.free()
        return res

    def lut_eval(self, bits, d, lut, batch):
        """LUT evaluation by vertical packing: `bits` = TGSW handle of batch*d samples (item b's bit i
        at b*d + i), lut [max(1, 2^(d-log2 N))][N] plaintext torus polynomials -> [batch][N+1] LWE"""
        N = self.params.N
        lut = np.ascontiguousarray(lut, self.torus).reshape(-1, N)
        t, out = self.to_device(lut), self.alloc(batch * (N + 1) * np.dtype(self.torus).itemsize)
        self._chk(self.lib.tfhe_amd_lut_eval(self.ctx, out.ptr, bits, d, t.ptr, batch))
        res = out.download(self.torus, (batch, N + 1))
        t.free()
        out.free()
        return res

    def blind_rotate(self, acc, bara):
        acc = np.ascontiguousarray(acc, self.torus).reshape(-1, 2, self.params.N)
        bara = np.ascontiguousarray(bara, np.int32).reshape(acc.shape[0], self.params.n)
        d, r = self.to_device(acc), self.to_device(bara)
        self._chk(self.lib.tfhe_amd_blind_rotate(self.ctx, d.ptr, r.ptr, acc.shape[0]))
        out = d.download(self.torus, acc.shape)
        d.free()
        r.free()
        return out

    def blind_rotate_extract(self, v, rot):
        rot = np.ascontiguousarray(rot, np.int32).reshape(-1, self.params.n + 1)
        v = np.ascontiguousarray(v, self.torus)
        per_sample = int(v.ndim == 2)
        B, N = rot.shape[0], self.params.N
        d_v, d_r = self.to_device(v), self.to_device(rot)
        d_o = self.alloc(B * (N + 1) * v.itemsize)
        self._chk(self.lib.tfhe_amd_blind_rotate_extract(self.ctx, d_o.ptr, d_v.ptr, per_sample, d_r.ptr, B))
        out = d_o.download(self.torus, (B, N + 1))
        for d in (d_v, d_r, d_o):
            d.free()
        return out

    def bootstrap_woks(self, mu, x):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, self.params.n + 1)
        return self._roundtrip(lambda c, o, i, b: self.lib.tfhe_amd_bootstrap_woks(c, o, int(mu), i, b), x, np.int32,
                               (x.shape[0], self.params.N + 1), x.shape[0])

    def keyswitch(self, x):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, self.params.N + 1)
        return self._roundtrip(self.lib.tfhe_amd_keyswitch, x, np.int32, (x.shape[0], self.params.ks_n_out + 1),
                               x.shape[0])

    def bootstrap(self, mu, x, streamed=False):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, self.params.n + 1)
        fn = self.lib.tfhe_amd_bootstrap_streamed if streamed else self.lib.tfhe_amd_bootstrap
        return self._roundtrip(lambda c, o, i, b: fn(c, o, int(mu), i, b), x, np.int32, x.shape, x.shape[0])

    def bootstrap_host(self, mu, x):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, self.params.n + 1)
        out = np.empty_like(x)
        self._chk(self.lib.tfhe_amd_bootstrap_host(self.ctx, _np_ptr(out), int(mu), _np_ptr(x), x.shape[0]))
        return out

    def cb_bootstrap_woks(self, mu, abar):
        abar = np.ascontiguousarray(abar, np.int32).reshape(-1, self.params.n + 1)
        return self._roundtrip(lambda c, o, i, b: self.lib.tfhe_amd_cb_bootstrap_woks(c, o, int(mu), i, b), abar,
                               np.int64, (abar.shape[0], self.params.N + 1), abar.shape[0])

    def modswitch(self, x):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, self.params.n + 1)
        return self._roundtrip(self.lib.tfhe_amd_modswitch, x, np.int32, x.shape, x.shape[0])


class CircuitBootstrap:
    """tfhe_amd_cb: the three-level circuit bootstrap of the PoC (tfhe_CircuitBootstrapFFT)."""

    def __init__(self, n0=500, N1=1024, N2=2048, l1=2, Bgbit1=8, l2=4, Bgbit2=9, t10=6, bb10=2, t21=10, bb21=3,
                 device=0, lib_path=None):
        self.lib = load_library(lib_path)
        self.p = CbParams(n0, N1, N2, l1, Bgbit1, l2, Bgbit2, t10, bb10, t21, bb21)
        self.cb = C.c_void_p()
        rc = self.lib.tfhe_amd_cb_create(C.byref(self.p), device, C.byref(self.cb))
        if rc != OK:
            raise TfheAmdError(f"tfhe_amd_cb_create failed with status {rc}")
        self.ctx = C.c_void_p(self.lib.tfhe_amd_cb_ctx_lvl2(self.cb))  # memory helpers go through lvl2
        self.ctx10 = C.c_void_p(self.lib.tfhe_amd_cb_ctx_lvl10(self.cb))  # Torus32, N1, gadget (l1, Bgbit1)

    def _chk(self, rc):
        if rc != OK:
            raise TfheAmdError(f"status {rc}: {self.lib.tfhe_amd_cb_last_error(self.cb).decode()}")

    def close(self):
        if self.cb:
            self.lib.tfhe_amd_cb_destroy(self.cb)
            self.cb = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_preks(self, preks):
        a = np.ascontiguousarray(preks, np.int32)
        self._chk(self.lib.tfhe_amd_cb_load_preks(self.cb, _np_ptr(a)))

    def load_bk_fft(self, bk):
        a = np.ascontiguousarray(bk, np.float64)
        self._chk(self.lib.tfhe_amd_cb_load_bk_fft(self.cb, _np_ptr(a)))

    def load_bk_torus(self, bk):
        a = np.ascontiguousarray(bk, np.int64)
        self._chk(self.lib.tfhe_amd_cb_load_bk_torus(self.cb, _np_ptr(a)))

    def load_privks(self, privks):
        a = np.ascontiguousarray(privks, np.int32)
        a = a.reshape(2, -1)
        for u in range(2):
            self._chk(self.lib.tfhe_amd_cb_load_privks_plane(self.cb, u, _np_ptr(a[u])))

    def _dev(self, arr):
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._chk(self.lib.tfhe_amd_malloc(self.ctx, C.byref(p), max(arr.nbytes, 1)))
        self._chk(self.lib.tfhe_amd_memcpy_h2d(self.ctx, p, _np_ptr(arr), arr.nbytes))
        return p

    def _out(self, p, dtype, shape):
        out = np.empty(shape, dtype)
        self._chk(self.lib.tfhe_amd_memcpy_d2h(self.ctx, _np_ptr(out), p, out.nbytes))
        self.lib.tfhe_amd_free(self.ctx, p)
        return out

    def privks(self, u, x):
        x = np.ascontiguousarray(x, np.int64).reshape(-1, self.p.N2 + 1)
        B = x.shape[0]
        d_x = self._dev(x)
        d_o = self._dev(np.zeros((B, 2, self.p.N1), np.int32))
        self._chk(self.lib.tfhe_amd_privks(self.cb, d_o, u, d_x, B))
        self.lib.tfhe_amd_free(self.ctx, d_x)
        return self._out(d_o, np.int32, (B, 2, self.p.N1))

    def circuit_bootstrap(self, x):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, self.p.N1 + 1)
        B = x.shape[0]
        d_x = self._dev(x)
        d_o = self._dev(np.zeros((B, 2, self.p.l1, 2, self.p.N1), np.int32))
        self._chk(self.lib.tfhe_amd_circuit_bootstrap(self.cb, d_o, d_x, B))
        self.lib.tfhe_amd_free(self.ctx, d_x)
        return self._out(d_o, np.int32, (B, 2, self.p.l1, 2, self.p.N1))


    def circuit_bootstrap_lut(self, x, d, lut):
        """BASELINE config 3's chain, entirely on the device: circuit-bootstrap B*d LWE-encrypted bits
        (row b*d + i = bit i of item b), convert the TGSW32 outputs to Lagrange form, evaluate the
        2^d-entry table by vertical packing.  Returns ([B*d][2][l1][2][N1] TGSW32, [B][N1+1] LWE)."""
        p = self.p
        x = np.ascontiguousarray(x, np.int32).reshape(-1, p.N1 + 1)
        B = x.shape[0] // d
        assert B * d == x.shape[0]
        lut = np.ascontiguousarray(lut, np.int32).reshape(-1, p.N1)
        d_x, d_lut = self._dev(x), self._dev(lut)
        d_g = self._dev(np.zeros((B * d, 2, p.l1, 2, p.N1), np.int32))
        d_o = self._dev(np.zeros((B, p.N1 + 1), np.int32))
        self._chk(self.lib.tfhe_amd_circuit_bootstrap(self.cb, d_g, d_x, B * d))
        bits = C.c_void_p()
        rc = self.lib.tfhe_amd_gsw_from_torus_d(self.ctx10, d_g, B * d, C.byref(bits))
        if rc == OK:
            rc = self.lib.tfhe_amd_lut_eval(self.ctx10, d_o, bits, d, d_lut, B)
        if rc != OK:
            raise TfheAmdError(f"status {rc}: {self.lib.tfhe_amd_last_error(self.ctx10).decode()}")
        out = self._out(d_o, np.int32, (B, p.N1 + 1))
        tgsw = self._out(d_g, np.int32, (B * d, 2, p.l1, 2, p.N1))
        self.lib.tfhe_amd_gsw_free(bits)
        self.lib.tfhe_amd_free(self.ctx, d_x)
        self.lib.tfhe_amd_free(self.ctx, d_lut)
        return tgsw, out


class Pool:
    """tfhe_amd_pool: one context, one host thread and one pinned staging buffer per member device; the caller's keys
    uploaded once to every member; host batches cut into contiguous slices (include/tfhe_amd.h)."""

    def __init__(self, devices, torus_bits=32, n=630, N=1024, l=2, Bgbit=10, ks_t=8, ks_basebit=2, ks_n_out=None, lib_path=None):
        self.lib = load_library(lib_path)
        self.params = Params(torus_bits, n, N, 1, l, Bgbit, ks_t, ks_basebit, (n if ks_n_out is None else ks_n_out) if ks_t else 0)
        devs = (C.c_int * len(devices))(*devices)
        self.pool = C.c_void_p()
        rc = self.lib.tfhe_amd_pool_create(C.byref(self.params), devs, len(devices), C.byref(self.pool))
        if rc != OK:
            raise TfheAmdError(f"tfhe_amd_pool_create({list(devices)}) failed with status {rc} (2 = a device is missing; there is no CPU path)")
        self.devices = list(devices)

    def _chk(self, rc):
        if rc != OK:
            raise TfheAmdError(f"status {rc}: {self.lib.tfhe_amd_pool_last_error(self.pool).decode()}")

    def close(self):
        if self.pool:
            self.lib.tfhe_amd_pool_destroy(self.pool)
            self.pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_keys(self, bkfft=None, ks=None):
        a = None if bkfft is None else np.ascontiguousarray(bkfft, np.float64)
        k = None if ks is None else np.ascontiguousarray(ks, np.int32)
        self._chk(self.lib.tfhe_amd_pool_load_keys(self.pool, None if a is None else _np_ptr(a), None if k is None else _np_ptr(k)))

    def load_keys_torus(self, bk_torus=None, ks=None):
        a = None if bk_torus is None else np.ascontiguousarray(bk_torus, np.int32 if self.params.torus_bits == 32 else np.int64)
        k = None if ks is None else np.ascontiguousarray(ks, np.int32)
        self._chk(self.lib.tfhe_amd_pool_load_keys_torus(self.pool, None if a is None else _np_ptr(a), None if k is None else _np_ptr(k)))

    def _rows(self, fn, x, in_cols, out_cols, *lead):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, in_cols)
        out = np.empty((x.shape[0], out_cols), np.int32)
        self._chk(fn(self.pool, _np_ptr(out), *lead, _np_ptr(x), x.shape[0]))
        return out

    def bootstrap(self, mu, x):
        n = self.params.n
        return self._rows(self.lib.tfhe_amd_pool_bootstrap_host, x, n + 1, n + 1, int(mu))

    def bootstrap_woks(self, mu, x):
        return self._rows(self.lib.tfhe_amd_pool_bootstrap_woks_host, x, self.params.n + 1, self.params.N + 1, int(mu))

    def keyswitch(self, x):
        return self._rows(self.lib.tfhe_amd_pool_keyswitch_host, x, self.params.N + 1, self.params.ks_n_out + 1)

    def set_chunk_rows(self, rows):
        """rows per chunk of the pipelined form (slices of >= 2 x rows are cut into chunks on two streams); 0 = never"""
        self._chk(self.lib.tfhe_amd_pool_set_option(self.pool, POOL_OPT_CHUNK_ROWS, int(rows)))

    def last_split(self):
        m = len(self.devices)
        cnt, sec = (C.c_int * m)(), (C.c_double * m)()
        self._chk(self.lib.tfhe_amd_pool_last_split(self.pool, cnt, sec))
        return list(cnt), list(sec)


class CircuitBootstrapPool:
    """tfhe_amd_cb_pool: the circuit bootstrap sharded over several devices the same way"""

    def __init__(self, devices, n0=500, N1=1024, N2=2048, l1=2, Bgbit1=8, l2=4, Bgbit2=9, t10=6, bb10=2, t21=10, bb21=3, lib_path=None):
        self.lib = load_library(lib_path)
        self.p = CbParams(n0, N1, N2, l1, Bgbit1, l2, Bgbit2, t10, bb10, t21, bb21)
        devs = (C.c_int * len(devices))(*devices)
        self.pool = C.c_void_p()
        rc = self.lib.tfhe_amd_cb_pool_create(C.byref(self.p), devs, len(devices), C.byref(self.pool))
        if rc != OK:
            raise TfheAmdError(f"tfhe_amd_cb_pool_create({list(devices)}) failed with status {rc}")

    def _chk(self, rc):
        if rc != OK:
            raise TfheAmdError(f"status {rc}: {self.lib.tfhe_amd_cb_pool_last_error(self.pool).decode()}")

    def close(self):
        if self.pool:
            self.lib.tfhe_amd_cb_pool_destroy(self.pool)
            self.pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_preks(self, preks):
        self._chk(self.lib.tfhe_amd_cb_pool_load_preks(self.pool, _np_ptr(np.ascontiguousarray(preks, np.int32))))

    def load_bk_fft(self, bk):
        self._chk(self.lib.tfhe_amd_cb_pool_load_bk_fft(self.pool, _np_ptr(np.ascontiguousarray(bk, np.float64))))

    def load_privks(self, privks):
        a = np.ascontiguousarray(privks, np.int32).reshape(2, -1)
        for u in range(2):
            self._chk(self.lib.tfhe_amd_cb_pool_load_privks_plane(self.pool, u, _np_ptr(a[u])))

    def set_chunk_rows(self, rows):
        self._chk(self.lib.tfhe_amd_cb_pool_set_option(self.pool, POOL_OPT_CHUNK_ROWS, int(rows)))

    def circuit_bootstrap(self, x):
        x = np.ascontiguousarray(x, np.int32).reshape(-1, self.p.N1 + 1)
        out = np.empty((x.shape[0], 2, self.p.l1, 2, self.p.N1), np.int32)
        self._chk(self.lib.tfhe_amd_cb_pool_circuit_bootstrap_host(self.pool, _np_ptr(out), _np_ptr(x), x.shape[0]))
        return out


def device_count(lib_path=None):
    """HIP devices the engine library sees in this process (tfhe_amd_device_count)"""
    lib = load_library(lib_path)
    n = C.c_int()
    if lib.tfhe_amd_device_count(C.byref(n)) != OK:
        return 0
    return int(n.value)


def device_pci_bus_id(device=0, lib_path=None):
    """'domain:bus:device.function' of the GPU behind an ordinal, or None"""
    lib = load_library(lib_path)
    buf = C.create_string_buffer(32)
    return buf.value.decode() if lib.tfhe_amd_device_pci_bus_id(device, buf, 32) == OK else None


# ---- harness wrappers (host side of the ABI; usable without a GPU) --------------------
def build_tables(N, lib_path=None):
    """the reference-layout twiddle tables (fft, ifft) from the host-only builder: no context, no device"""
    lib = load_library(lib_path)
    f, r = np.empty(max(2 * N - 8, 1)), np.empty(max(2 * N - 8, 1))
    rc = lib.tfhe_amd_build_tables(int(N), _np_ptr(f), _np_ptr(r))
    if rc != OK:
        raise TfheAmdError(f"status {rc}: tfhe_amd_build_tables({N}): the ring degree must be a power of two, 16 <= N <= 2^20")
    return f, r


def keygen_binary(n, seed, stream, lib_path=None):
    lib = load_library(lib_path)
    key = np.empty(n, np.int32)
    assert lib.tfhe_amd_keygen_binary(_np_ptr(key), n, seed, stream) == OK
    return key


def lwe_encrypt32(mess, stdev, key, seed, stream, lib_path=None):
    lib = load_library(lib_path)
    key = np.ascontiguousarray(key, np.int32)
    ct = np.empty(key.size + 1, np.int32)
    assert lib.tfhe_amd_lwe_encrypt32(_np_ptr(ct), int(mess), float(stdev), _np_ptr(key), key.size, seed, stream) == OK
    return ct


def lwe_phase32(ct, key, lib_path=None):
    lib = load_library(lib_path)
    ct, key = np.ascontiguousarray(ct, np.int32), np.ascontiguousarray(key, np.int32)
    return int(lib.tfhe_amd_lwe_phase32(_np_ptr(ct), _np_ptr(key), key.size))


def keygen_bk_torus(torus_bits, lwe_key, tlwe_key, l, Bgbit, stdev, seed, stream, lib_path=None):
    lib = load_library(lib_path)
    lwe_key, tlwe_key = np.ascontiguousarray(lwe_key, np.int32), np.ascontiguousarray(tlwe_key, np.int32)
    n, N = lwe_key.size, tlwe_key.size
    dt = np.int32 if torus_bits == 32 else np.int64
    fn = lib.tfhe_amd_keygen_bk_torus32 if torus_bits == 32 else lib.tfhe_amd_keygen_bk_torus64
    bk = np.empty((n, 2 * l, 2, N), dt)
    assert fn(_np_ptr(bk), _np_ptr(lwe_key), n, _np_ptr(tlwe_key), N, l, Bgbit, float(stdev), seed, stream) == OK
    return bk


def keygen_ks32(in_key, out_key, t, basebit, stdev, seed, stream, lib_path=None):
    lib = load_library(lib_path)
    in_key, out_key = np.ascontiguousarray(in_key, np.int32), np.ascontiguousarray(out_key, np.int32)
    ks = np.empty((in_key.size, t, 1 << basebit, out_key.size + 1), np.int32)
    assert lib.tfhe_amd_keygen_ks32(_np_ptr(ks), _np_ptr(in_key), in_key.size, _np_ptr(out_key), out_key.size, t,
                                    basebit, float(stdev), seed, stream) == OK
    return ks


def hp_twiddles(n, lib_path=None):
    """Real96 twiddle tables of tfhe_amd_hp_twiddles: (powomega, powombar), each [n][4] uint64"""
    lib = load_library(lib_path)
    a, b = np.empty((n, 4), np.uint64), np.empty((n, 4), np.uint64)
    assert lib.tfhe_amd_hp_twiddles(n, _np_ptr(a), _np_ptr(b)) == OK
    return a, b


def device_info(device=0, lib_path=None):
    """one-line description of the GPU (tfhe_amd_device_info), or None when there is none"""
    lib = load_library(lib_path)
    buf = C.create_string_buffer(512)
    return buf.value.decode() if lib.tfhe_amd_device_info(device, buf, 512) == OK else None
