"""The C-ABI library: builds for gfx950, loads without a GPU, exports every symbol
include/tfhe_amd.h declares, and refuses to create a context when no device exists
(no CPU fallback).  No compute calls here."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = importlib.import_module("experimental-tfhe_amd")


@pytest.fixture(scope="module")
def hip_lib():
    build = importlib.import_module("experimental-tfhe_amd.build")
    path = build.build()  # hipcc --offload-arch=gfx950 (cross-compiles here)
    assert os.path.exists(path)
    return T.load_library(path)


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tfhe_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfhe_amd_[a-z0-9_]+)\s*\(", text)))


def test_headers_compile_under_the_reference_flags(tmp_path):
    """the reference builds with `-Wall -Werror` (circuit-bootstrapping/src/Makefile): every public header must survive that
    (and -Wextra) in a translation unit of its own; tfhe_amd.h, the C ABI, also as C99"""
    import subprocess
    inc = os.path.join(ROOT, "include")
    for h in sorted(f for f in os.listdir(inc) if f.endswith((".h", ".hpp"))):
        tu = tmp_path / (h + ".cpp")
        tu.write_text('#include "%s"\n' % h)
        subprocess.check_call(["g++", "-std=gnu++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + inc, str(tu)])
    tu = tmp_path / "abi.c"
    tu.write_text('#include "tfhe_amd.h"\nint main(void) { return tfhe_amd_version() == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + inc, str(tu)])


def test_exports_every_declared_symbol(hip_lib):
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(hip_lib, s), f"{s} declared in include/tfhe_amd.h but not exported"
    assert sorted(T.ABI_SYMBOLS) == syms, "binding's symbol list out of date with the header"


def test_dropin_library_exports_the_reference_names(hip_lib):
    """libtfhe_amd_dropin.so: the reference's library-form entry points as global extern "C" symbols
    (include/tfhe_amd_dropin.h; CB/lwe_functions.cpp:163,337,366,399,434, CB/tgsw_functions.cpp:424)"""
    build = importlib.import_module("experimental-tfhe_amd.build")
    lib = C.CDLL(build.build_dropin())
    text = open(os.path.join(ROOT, "include", "tfhe_amd_dropin.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    block = text[text.index('extern "C" {'):text.index("#else")]
    names = sorted(set(re.findall(r"\bvoid\s+([A-Za-z_0-9]+)\s*\(", block)))
    assert names == sorted(["tfhe_blindRotate_FFT", "tfhe_blindRotateAndExtract_FFT", "tfhe_bootstrap_woKS_FFT", "tfhe_bootstrap_FFT",
                            "tGswFFTExternMulToTLwe", "tfhe_MuxRotate_FFT", "lweKeySwitch", "tfhe_amd_dropin_release", "tfhe_amd_dropin_set_device", "tfhe_amd_dropin_set_devices",
                            "tfhe_bootstrap_woKS_FFT_array", "tfhe_bootstrap_FFT_array", "lweKeySwitch_array"])
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tfhe_amd_dropin.h but not exported"


def test_version_and_params_struct(hip_lib):
    assert b"gfx950" in hip_lib.tfhe_amd_version()
    assert C.sizeof(T.Params) == 9 * 4


def test_bad_parameters_rejected(hip_lib):
    ctx = C.c_void_p()
    for bad in (dict(k=2), dict(N=8), dict(N=1000), dict(N=1 << 21), dict(torus_bits=16), dict(l=0), dict(l=4, Bgbit=10), dict(n=0)):
        kw = dict(torus_bits=32, n=630, N=1024, k=1, l=2, Bgbit=10, ks_t=8, ks_basebit=2, ks_n_out=630)
        kw.update(bad)
        p = T.Params(*[kw[f[0]] for f in T.Params._fields_])
        assert hip_lib.tfhe_amd_ctx_create(C.byref(p), 0, C.byref(ctx)) == T.ERR_PARAM, bad
    assert hip_lib.tfhe_amd_ctx_create(None, 0, C.byref(ctx)) == T.ERR_PARAM


def test_no_gpu_means_no_context(hip_lib):
    """On a machine without a GPU the library must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(T.TfheAmdError):
        T.Engine()


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(T.TfheAmdError):
        T.load_library(str(tmp_path / "nope.so"))


def test_host_harness_matches_oracle(hip_lib):
    """keygen/encrypt/phase are host code inside the HIP library: usable (and pinned) without a GPU"""
    import oracle_py as O
    seed = 0x5446484500000001
    k = T.keygen_binary(630, seed, 1)
    assert np.array_equal(k, O.keygen_binary(630, seed, 1))
    tk = T.keygen_binary(1024, seed, 2)
    ct = T.lwe_encrypt32(1 << 29, 2.0 ** -15, k, seed, 9)
    assert np.array_equal(ct, O.lwe_encrypt32(1 << 29, 2.0 ** -15, k, O.rng(seed, 9)))
    assert T.lwe_phase32(ct, k) == O.lwe_phase32(ct, k)
    bk = T.keygen_bk_torus(32, k[:3], tk, 2, 10, 2.0 ** -25, seed, 1000)
    want = O.bk_create32(1024, k[:3], tk, 2, 10, 2.0 ** -25, seed, 1000)
    assert np.array_equal(O.execute_reverse_int(1024, bk.reshape(-1, 1024)).reshape(want.shape).view(np.uint64),
                          want.view(np.uint64))
    ks = T.keygen_ks32(tk[:16], k[:20], 8, 2, 2.0 ** -15, seed, 100000)
    assert np.array_equal(ks, O.ks_create32(tk[:16], k[:20], 8, 2, 2.0 ** -15, seed, 100000))
