"""The LITERAL drop-in: drivers that declare the reference's entry points at global scope
with the reference's full signatures -- no namespace, `const Globals* env` kept -- link against
libtfhe_amd_dropin.so (library form, extern "C") / csrc/dropin_poc.cpp compiled next to poc_types.h
(PoC form, C++ linkage) and get the oracle's results.  CPU: against the tests/emu build, with the stand-in
header AND (where /root/reference exists) the reference's own poc_types.h.  GPU: the same drivers against
the HIP library; they run before this process touches the GPU (conftest.pytest_collection_finish)."""
import importlib
import os
import subprocess

import pytest

import test_compat as TC

ROOT = TC.ROOT
BUILD = os.path.join(ROOT, "tests", "emu", "_build")
INC = os.path.join(ROOT, "include")
REF_SRC = "/root/reference/circuit-bootstrapping/src"
POC = dict(n0=8, N1=1024, N2=2048, l1=2, bg1=8, l2=4, bg2=9, t10=6, bb10=2, t21=2, bb21=3, count=2)      # GPU
POC_EMU = dict(n0=3, N1=1024, N2=1024, l1=2, bg1=8, l2=3, bg2=10, t10=3, bb10=2, t21=2, bb21=2, count=1)  # emulator: small
LIB = dict(n=16, count=3)
LIB_EMU = dict(n=5, count=2)


def _newer(out, deps):
    return os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps)


def build_lib_driver(engine_lib, tag):
    """compat_driver.cpp -DDROPIN against libtfhe_amd_dropin (built for `engine_lib`)"""
    b = importlib.import_module("experimental-tfhe_amd.build")
    os.makedirs(BUILD, exist_ok=True)
    dropin = b.build_dropin(engine_lib, os.path.join(BUILD, f"libtfhe_amd_dropin_{tag}.so")) if tag != "hip" else b.build_dropin()
    out = os.path.join(BUILD, f"dropin_driver_lib_{tag}")
    src = os.path.join(ROOT, "tests", "compat", "compat_driver.cpp")
    if not b.linked_against(out, engine_lib, [src, dropin, os.path.join(INC, "tfhe_amd_dropin.h")]):
        d = os.path.dirname(dropin)
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-fopenmp", "-DDROPIN", "-I" + INC, src, "-o", out, "-L" + d,
                               "-l:" + os.path.basename(dropin), "-Wl,-rpath," + d, "-lpthread"])
        b.record_engine(out, engine_lib)
    return out


def build_poc_driver(engine_lib, tag, real_header=False, POC=POC):
    """dropin_driver_poc.cpp + csrc/dropin_poc.cpp, both compiled next to the same poc_types.h"""
    os.makedirs(BUILD, exist_ok=True)
    sig = "_".join(str(POC[key]) for key in ("n0", "N1", "N2", "l1", "bg1", "l2", "bg2", "t10", "bb10", "t21", "bb21"))
    out = os.path.join(BUILD, f"dropin_driver_poc_{tag}_{sig}" + ("_ref" if real_header else ""))
    srcs = [os.path.join(ROOT, "tests", "compat", "dropin_driver_poc.cpp"),
            os.path.join(ROOT, "experimental-tfhe_amd", "csrc", "dropin_poc.cpp")]
    hdr = REF_SRC if real_header else os.path.join(ROOT, "tests", "compat", "poc_stub")
    b = importlib.import_module("experimental-tfhe_amd.build")
    if not b.linked_against(out, engine_lib, srcs + [engine_lib, os.path.join(INC, "tfhe_amd_compat.hpp"), os.path.join(INC, "tfhe_amd_dropin.h")]):
        defs = [f"-DP_N0={POC['n0']}", f"-DP_N1={POC['N1']}", f"-DP_N2={POC['N2']}", f"-DP_L1={POC['l1']}", f"-DP_BG1={POC['bg1']}",
                f"-DP_L2={POC['l2']}", f"-DP_BG2={POC['bg2']}", f"-DP_T10={POC['t10']}", f"-DP_BB10={POC['bb10']}",
                f"-DP_T21={POC['t21']}", f"-DP_BB21={POC['bb21']}"]
        if real_header:
            defs += ["-DUSE_FFT", "-DDROPIN_REAL_HEADER"]  # as the PoC's Makefile builds poc_types.h
        d = os.path.dirname(engine_lib)
        subprocess.check_call(["g++", "-std=gnu++11", "-O1", "-I" + hdr, "-I" + INC] + defs + srcs + ["-o", out, "-L" + d,
                               "-l:" + os.path.basename(engine_lib), "-Wl,-rpath," + d, "-lpthread"])
        b.record_engine(out, engine_lib)
    return out


def run_poc(driver, d, phase, POC=POC):
    """same files as test_compat.run_poc_form; the drop-in driver takes (in, out) only"""
    if phase in ("run", "both"):
        TC.run_poc_form(None, d, phase="write", **POC)
        subprocess.check_call([driver, os.path.join(str(d), "in.bin"), os.path.join(str(d), "out.bin")])
    if phase in ("check", "both"):
        TC.run_poc_form(None, d, phase="check", **POC)


def test_library_form_literal_dropin_emu(emu_lib, tmp_path):
    TC.run_lib_form(build_lib_driver(emu_lib, "emu"), tmp_path, plugin=False, **LIB_EMU)


def test_array_forms_literal_dropin_emu(emu_lib, tmp_path):
    """extern "C" tfhe_bootstrap_FFT_array / tfhe_bootstrap_woKS_FFT_array / lweKeySwitch_array of libtfhe_amd_dropin.so"""
    import numpy as np
    import oracle_py as O
    stats, got, (bk, ks, x) = TC.run_array_form(build_lib_driver(emu_lib, "emu"), tmp_path, n=4, count=5)
    want = np.stack([O.bootstrap32(1024, bk, ks, 1 << 29, x[c], 2, 10, 8, 2) for c in range(5)])
    assert np.array_equal(got, want)


def test_poc_form_literal_dropin_emu(emu_lib, tmp_path):
    """driver + forwarding source compiled against the reference's OWN poc_types.h where /root/reference
    exists (this container), against the stand-in header elsewhere"""
    real = os.path.isdir(REF_SRC)
    run_poc(build_poc_driver(emu_lib, "emu", real_header=real, POC=POC_EMU), tmp_path, "both", POC=POC_EMU)


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources not present (GPU box)")
def test_stand_in_header_builds_too():
    """the stand-in header used on the GPU box compiles and links the same two sources (no run here)"""
    T = importlib.import_module("experimental-tfhe_amd")
    if os.path.exists(T.DEFAULT_LIB):
        build_poc_driver(T.DEFAULT_LIB, "hip")


GPU_RUN_DIR = os.path.join(BUILD, "dropin_gpu_run")


def prerun_gpu_drivers():
    """conftest.pytest_collection_finish, on a GPU box, before this process initialises the GPU"""
    T = importlib.import_module("experimental-tfhe_amd")
    for form in ("lib", "poc"):
        d = os.path.join(GPU_RUN_DIR, form)
        os.makedirs(d, exist_ok=True)
        for f in ("in.bin", "out.bin"):
            if os.path.exists(os.path.join(d, f)):
                os.remove(os.path.join(d, f))
    TC.run_lib_form(build_lib_driver(T.DEFAULT_LIB, "hip"), os.path.join(GPU_RUN_DIR, "lib"), phase="run", plugin=False, **LIB)
    run_poc(build_poc_driver(T.DEFAULT_LIB, "hip"), os.path.join(GPU_RUN_DIR, "poc"), "run")


def _need(form):
    d = os.path.join(GPU_RUN_DIR, form)
    assert os.path.exists(os.path.join(d, "out.bin")), "the drop-in driver did not run before the session's GPU tests"
    return d


@pytest.mark.gpu
def test_library_form_literal_dropin_gpu():
    TC.run_lib_form(None, _need("lib"), phase="check", plugin=False, **LIB)


@pytest.mark.gpu
def test_poc_form_literal_dropin_gpu():
    run_poc(None, _need("poc"), "check")
