"""include/tfhe_amd_compat.hpp: the reference-named C++ entry points (tfhe_bootstrap_FFT,
tfhe_blindRotateAndExtract_FFT, tGswFFTExternMulToTLwe, lweKeySwitch, the FFT processor look-alike,
and the PoC's tfhe_CircuitBootstrapFFT / circuitBootstrapWoKS / circuitPrivKS / preKeySwitch /
preModSwitch) give the oracle's results.  The C++ driver (tests/compat/compat_driver.cpp) is
linked against the tests/emu build here; on a GPU box the same driver links against the HIP
library (test_compat_gpu)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5446484500000001


def build_driver(lib_path, tag, defines=()):
    out = os.path.join(ROOT, "tests", "emu", "_build", f"compat_driver_{tag}")
    src = os.path.join(ROOT, "tests", "compat", "compat_driver.cpp")
    hdr = os.path.join(ROOT, "include", "tfhe_amd_compat.hpp")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    b = importlib.import_module("experimental-tfhe_amd.build")
    # (rebuilt also when the driver of this name was last linked against ANOTHER engine build -- the sanitizer run links the
    # same sources against libtfhe_amd_emu_san.so)
    if not b.linked_against(out, lib_path, [src, hdr, lib_path]):
        libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-g", "-fopenmp"] + ["-D" + d for d in defines] + ["-I" + os.path.join(ROOT, "include"), src, "-o", out,
                               "-L" + libdir, "-l:" + libname, "-Wl,-rpath," + libdir, "-lpthread"])
        b.record_engine(out, lib_path)
    return out


def run_lib_form(driver, tmp_path, N=1024, n=5, l=2, Bgbit=10, t=8, bb=2, count=2, phase="both", plugin=True):
    """phase: "run" = write the inputs and execute the driver; "check" = compare the outputs the
    driver left in tmp_path with the oracle (no child process); "both" = the two in sequence"""
    rs = np.random.RandomState(7)
    lk, tk = O.keygen_binary(n, SEED, 1), O.keygen_binary(N, SEED, 2)
    bk = O.bk_create32(N, lk, tk, l, Bgbit, 2.0 ** -25, SEED, 1000)
    ks = O.ks_create32(tk, lk, t, bb, 2.0 ** -15, SEED, 100000)
    mu = 1 << 29
    x = np.stack([O.lwe_encrypt32(mu if i % 2 else -mu, 2.0 ** -15, lk, O.rng(SEED, 40 + i)) for i in range(count)])
    rot = rs.randint(0, 2 * N, size=(count, n + 1)).astype(np.int32)
    acc = rs.randint(-2 ** 31, 2 ** 31, size=(count, 2, N)).astype(np.int32)
    v = rs.randint(-2 ** 31, 2 ** 31, size=N).astype(np.int32)
    hdr = np.array([n, N, l, Bgbit, t, bb, count, mu], np.int32)
    fi, fo = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    if phase in ("run", "both"):
        with open(fi, "wb") as f:
            for a in (hdr, bk, ks, x, rot, acc, v):
                f.write(np.ascontiguousarray(a).tobytes())
        subprocess.check_call([driver, "lib", fi, fo])
        if phase == "run":
            return
    raw = open(fo, "rb").read()
    pos = 0

    def take(dtype, cnt):
        nonlocal pos
        a = np.frombuffer(raw, dtype=dtype, count=cnt, offset=pos)
        pos += a.nbytes
        return a

    for c in range(count):
        woks = O.bootstrap_woks32(N, bk, mu, x[c], l, Bgbit)
        assert np.array_equal(take(np.int32, N + 1), woks), "tfhe_bootstrap_woKS_FFT"
        full = O.bootstrap32(N, bk, ks, mu, x[c], l, Bgbit, t, bb)
        assert np.array_equal(take(np.int32, n + 1), full), "tfhe_bootstrap_FFT"
        assert np.array_equal(take(np.int32, n + 1), O.keyswitch32(ks, woks, N, n, t, bb)), "lweKeySwitch"
        assert np.array_equal(take(np.int32, N + 1),
                              O.blind_rotate_extract32(N, v, bk, rot[c, n], rot[c, :n], l, Bgbit)), "blindRotateAndExtract"
        assert np.array_equal(take(np.int32, 2 * N), O.blind_rotate32(N, acc[c], bk, rot[c, :n], l, Bgbit).ravel()), "blindRotate"
        assert np.array_equal(take(np.int32, 2 * N), O.extprod32(N, acc[c], bk[n - 1], l, Bgbit).ravel()), "tGswFFTExternMulToTLwe"
        assert np.array_equal(take(np.int32, 2 * N), O.mux_rotate32(N, acc[c], bk[0], rot[c, 0], l, Bgbit).ravel()), "tfhe_MuxRotate_FFT"
        a_last = 0 if c == 0 else rot[c, n - 1]
        assert np.array_equal(take(np.int32, 2 * N), O.mux_rotate32(N, acc[c], bk[n - 1], a_last, l, Bgbit).ravel()), "tfhe_MuxRotate_FFT in place"
    # after release(bk): the key rebuilt at the same addresses with its TGSW samples in reverse order
    bkr = np.ascontiguousarray(bk[::-1])
    assert np.array_equal(take(np.int32, N + 1), O.bootstrap_woks32(N, bkr, mu, x[0], l, Bgbit)), "release(bk) + rebuilt key, woKS"
    assert np.array_equal(take(np.int32, n + 1), O.bootstrap32(N, bkr, ks, mu, x[0], l, Bgbit, t, bb)), "release(bk) + rebuilt key"
    # the original key put back in place WITHOUT a release: noticed through the content sample
    assert np.array_equal(take(np.int32, n + 1), O.bootstrap32(N, bk, ks, mu, x[0], l, Bgbit, t, bb)), "key rebuilt in place, no release"
    if n >= 5:  # one interior TGSW sample replaced in place (1 := 2), no release; then the original put back the same way
        bk1 = bk.copy()
        bk1[1] = bk[2]
        assert np.array_equal(take(np.int32, n + 1), O.bootstrap32(N, bk1, ks, mu, x[0], l, Bgbit, t, bb)), "one TGSW sample replaced in place"
        assert np.array_equal(take(np.int32, n + 1), O.bootstrap32(N, bk, ks, mu, x[0], l, Bgbit, t, bb)), "... and put back"
    if not plugin:  # the literal drop-in driver (tests/test_dropin.py) has no FFT-plugin section
        assert pos == len(raw)
        return
    lag = O.execute_reverse_int(N, acc[0, 0])
    assert np.array_equal(take(np.float64, N).view(np.uint64), lag.view(np.uint64)), "execute_reverse_int"
    assert np.array_equal(take(np.int32, N), O.execute_direct_torus32(N, lag)), "execute_direct_torus32"
    want = O.lagrange_addmul(N, O.execute_reverse_int(N, acc[0, 1]), lag, bk[0, 0, 0])
    assert np.array_equal(take(np.float64, N).view(np.uint64), want.view(np.uint64)), "AddMul"
    a64 = (acc[0, 0].astype(np.int64) << 32) ^ acc[0, 1].astype(np.int64)
    l64 = O.execute_reverse_torus64(N, a64)
    assert np.array_equal(take(np.float64, N).view(np.uint64), l64.view(np.uint64)), "execute_reverse_torus64"
    assert np.array_equal(take(np.int64, N), O.execute_direct_torus64(N, l64)), "execute_direct_torus64"
    assert pos == len(raw)


def _env(devices):
    """TFHE_COMPAT_DEVICES for the driver: the array forms then ALSO run over a pool of these devices (exit 3 if they differ)"""
    return dict(os.environ, TFHE_COMPAT_DEVICES=devices) if devices else None


def run_poc_form(driver, tmp_path, n0=3, N1=1024, N2=1024, l1=2, bg1=8, l2=3, bg2=10, t10=3, bb10=2, t21=2, bb21=2,
                 count=2, phase="both", devices=None):
    rs = np.random.RandomState(8)
    key0, key2 = O.keygen_binary(n0, SEED, 21), O.keygen_binary(N2, SEED, 23)
    bk = O.bk_create64(N2, key0, key2, l2, bg2, 2.0 ** -44, SEED, 3000)
    preks = O.fill32(303, N1 * t10 * (1 << bb10) * (n0 + 1)).reshape(N1, t10, 1 << bb10, n0 + 1)
    privks = O.fill32(404, 2 * (N2 + 1) * t21 * (1 << bb21) * 2 * N1).reshape(2, N2 + 1, t21, 1 << bb21, 2, N1)
    x = rs.randint(-2 ** 31, 2 ** 31, size=(count, N1 + 1)).astype(np.int32)
    x64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(count, N2 + 1), dtype=np.int64)
    hdr = np.array([n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, count], np.int32)
    fi, fo = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    if phase in ("run", "both", "write"):
        with open(fi, "wb") as f:
            for a in (hdr, preks, bk, privks, x, x64):
                f.write(np.ascontiguousarray(a).tobytes())
        if phase == "write":  # inputs only (tests/test_dropin.py runs its own driver on them)
            return
        subprocess.check_call([driver, "poc", fi, fo], env=_env(devices), stdout=subprocess.DEVNULL)
        if phase == "run":
            return
    raw = open(fo, "rb").read()
    pos = 0

    def take(dtype, cnt):
        nonlocal pos
        a = np.frombuffer(raw, dtype=dtype, count=cnt, offset=pos)
        pos += a.nbytes
        return a

    for c in range(count):
        pre = O.keyswitch32(preks, x[c], N1, n0, t10, bb10)
        assert np.array_equal(take(np.int32, n0 + 1), pre), "preKeySwitch"
        abar = O.pre_modswitch(pre, N2)
        assert np.array_equal(take(np.int32, n0 + 1), abar), "preModSwitch"
        assert np.array_equal(take(np.int64, N2 + 1), O.cb_bootstrap_woks64(N2, 1 << 56, abar, bk, l2, bg2)), "circuitBootstrapWoKS"
        assert np.array_equal(take(np.int32, 2 * N1), O.privks(privks[1], x64[c], N2, N1, t21, bb21)), "circuitPrivKS"
        want = O.circuit_bootstrap(x[c], preks, bk, privks, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21)
        assert np.array_equal(take(np.int32, 2 * l1 * 2 * N1), want.ravel()), "tfhe_CircuitBootstrapFFT"
        rows = np.asarray(want, np.int32).reshape(2 * l1, 2 * N1)
        sel = O.execute_reverse_int(N1, rows.reshape(-1, N1))
        assert np.array_equal(take(np.int32, 2 * N1), O.cmux32(N1, sel, rows[0], rows[-1], l1, bg1)), "CMux"
    assert pos == len(raw)


def run_array_form(driver, tmp_path, N=1024, n=5, l=2, Bgbit=10, t=8, bb=2, count=6, phase="both", inputs=None, devices=None):
    """`compat_driver arr`: the array forms behind the reference names (tfhe_bootstrap_FFT_array, tfhe_bootstrap_woKS_FFT_array
    + lweKeySwitch_array) against the driver's own one-by-one loop (exit code 3 if they differ) and against the oracle.
    inputs: (bk, ks, x) to use instead of the generated ones (the GPU leg shares test_ref_batch's full-size set)."""
    import json
    mu = 1 << 29
    if inputs is None:
        lk, tk = O.keygen_binary(n, SEED, 1), O.keygen_binary(N, SEED, 2)
        bk = O.bk_create32(N, lk, tk, l, Bgbit, 2.0 ** -25, SEED, 1000)
        ks = O.ks_create32(tk, lk, t, bb, 2.0 ** -15, SEED, 100000)
        x = np.stack([O.lwe_encrypt32(mu if i % 2 else -mu, 2.0 ** -15, lk, O.rng(SEED, 40 + i)) for i in range(count)])
    else:
        bk, ks, x = inputs
        count = x.shape[0]
    hdr = np.array([n, N, l, Bgbit, t, bb, count, mu], np.int32)
    fi, fo, fs = (os.path.join(str(tmp_path), f) for f in ("arr_in.bin", "arr_out.bin", "arr_stats.json"))
    if phase in ("run", "both"):
        with open(fi, "wb") as f:
            for a in (hdr, bk, ks, x):
                f.write(np.ascontiguousarray(a).tobytes())
        res = subprocess.run([driver, "arr", fi, fo], stdout=subprocess.PIPE, text=True, env=_env(devices))
        os.remove(fi)
        with open(fs, "w") as f:
            f.write(res.stdout)
        if res.returncode:
            raise RuntimeError(f"compat_driver arr failed ({res.returncode}): {res.stdout}")
        if phase == "run":
            return None
    stats = json.loads(open(fs).read())
    assert stats["count"] == count and stats["array_identical_to_loop"] is True
    assert stats["openmp"] is True and stats["parallel_for_bootstraps_per_s"] > 0  # the reference's parallel construct, coalesced
    got = np.fromfile(fo, np.int32).reshape(count, n + 1)
    return stats, got, (bk, ks, x)


def test_array_forms_emu(emu_lib, tmp_path):
    stats, got, (bk, ks, x) = run_array_form(build_driver(emu_lib, "emu"), tmp_path)
    want = np.stack([O.bootstrap32(1024, bk, ks, 1 << 29, x[c], 2, 10, 8, 2) for c in range(x.shape[0])])
    assert np.array_equal(got, want), "tfhe_bootstrap_FFT_array"


def test_long_array_calls_take_the_pool_route_on_one_device_emu(emu_lib, tmp_path):
    """on ONE device an array call of >= 4096 samples goes through a one-member pool (pipelined gather / copy / compute / copy /
    scatter straight from and to the caller's LweSample objects: tfhe_amd_pool_*_rows); the driver built with the threshold
    lowered to 5 takes that route with 6 samples -- identical to the one-by-one loop and the oracle"""
    drv = build_driver(emu_lib, "emu_poolmin5", defines=("TFHE_AMD_COMPAT_POOL_MIN=5",))
    stats, got, (bk, ks, x) = run_array_form(drv, tmp_path)
    want = np.stack([O.bootstrap32(1024, bk, ks, 1 << 29, x[c], 2, 10, 8, 2) for c in range(x.shape[0])])
    assert np.array_equal(got, want)


def test_array_forms_over_a_pool_emu(emu_lib, tmp_path):
    """set_devices({1, 4, 6}): the three array forms cut the driver's loop of 7 samples into 3 + 2 + 2 over a pool of three
    emulated devices (one upload of the driver's key per device); identical to the one-by-one loop on one device.  The
    emulator aborts on any operand used from another device than its owner's."""
    stats, got, (bk, ks, x) = run_array_form(build_driver(emu_lib, "emu"), tmp_path, count=7, devices="1,4,6")
    assert stats["pool_devices"] == 3 and stats["pool_identical_to_loop"] is True
    want = np.stack([O.bootstrap32(1024, bk, ks, 1 << 29, x[c], 2, 10, 8, 2) for c in range(x.shape[0])])
    assert np.array_equal(got, want)


def test_poc_array_form_over_a_pool_emu(emu_lib, tmp_path):
    """PocEngine::tfhe_CircuitBootstrapFFT_array on one device and over a tfhe_amd_cb_pool of devices {2, 5}: both identical
    to the one-sample calls (the driver exits 3 otherwise), which run_poc_form compares with the oracle"""
    run_poc_form(build_driver(emu_lib, "emu"), tmp_path, count=2, devices="2,5")


def test_library_form_shims_emu(emu_lib, tmp_path):
    run_lib_form(build_driver(emu_lib, "emu"), tmp_path)


@pytest.mark.parametrize("N", [512, 64])
def test_library_form_shims_other_ring_degrees_emu(emu_lib, tmp_path, N):
    """the reference-named entry points (tfhe_bootstrap_FFT, tfhe_blindRotateAndExtract_FFT, tfhe_MuxRotate_FFT, tGswFFTExternMulToTLwe,
    lweKeySwitch, the FFT plugin object) at ring degrees the reference's code accepts but its PoC never instantiates: the shims
    take N from the caller's parameter structs, the engine serves every power of two >= 16"""
    run_lib_form(build_driver(emu_lib, "emu"), tmp_path, N=N, n=5, l=2, Bgbit=8, t=4, bb=2)


def test_poc_form_shims_emu(emu_lib, tmp_path):
    run_poc_form(build_driver(emu_lib, "emu"), tmp_path)


GPU_LIB_ARGS = dict(n=16, count=3)
GPU_POC_ARGS = dict(n0=8, N2=2048, l2=4, bg2=9, t21=2, bb21=3, count=2)
GPU_RUN_DIR = os.path.join(ROOT, "tests", "emu", "_build", "compat_gpu_run")


def prerun_gpu_drivers():
    """called by conftest.pytest_collection_finish on a GPU box, BEFORE this process initialises
    the GPU: build the driver against the HIP library and run it (the driver is its own GPU
    process); the two tests below only compare its output files with the oracle."""
    import importlib
    T = importlib.import_module("experimental-tfhe_amd")
    drv = build_driver(T.DEFAULT_LIB, "hip")
    for form, fn, args in (("lib", run_lib_form, GPU_LIB_ARGS), ("poc", run_poc_form, GPU_POC_ARGS)):
        d = os.path.join(GPU_RUN_DIR, form)
        os.makedirs(d, exist_ok=True)
        for f in ("in.bin", "out.bin"):
            if os.path.exists(os.path.join(d, f)):
                os.remove(os.path.join(d, f))
        fn(drv, d, phase="run", **args, **({"devices": "0,0"} if form == "poc" else {}))  # poc: + the array form over a 2-member pool on device 0
    # array forms at BASELINE config 2's size: 4096 gate bootstraps (n = 630) through tfhe_bootstrap_FFT_array, the same
    # 4096 one by one through tfhe_bootstrap_FFT; inputs = test_ref_batch's (whose reference outputs the test compares with)
    import test_ref_batch as RB
    d = os.path.join(GPU_RUN_DIR, "arr")
    os.makedirs(d, exist_ok=True)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    lk, bk, ks, x = RB._inputs()
    run_array_form(drv, d, n=RB.n, inputs=(bk, ks, x), phase="run", devices="0,0")  # + over a pool of two members, both on device 0
    # the coalesced OpenMP loop with 256 threads (the default run above uses 32): stats only, kept beside the others
    d2 = os.path.join(GPU_RUN_DIR, "arr256")
    os.makedirs(d2, exist_ok=True)
    for f in os.listdir(d2):
        os.remove(os.path.join(d2, f))
    os.environ["TFHE_COMPAT_THREADS"] = "256"
    try:
        run_array_form(drv, d2, n=RB.n, inputs=(bk, ks, x), phase="run")
    finally:
        del os.environ["TFHE_COMPAT_THREADS"]


def _need(form):
    d = os.path.join(GPU_RUN_DIR, form)
    assert os.path.exists(os.path.join(d, "out.bin")), (
        "the shim driver did not run before the session's GPU tests (conftest.pytest_collection_finish runs it "
        "on a GPU box; see stderr for its failure)")
    return d


@pytest.mark.gpu
def test_library_form_shims_gpu():
    run_lib_form(None, _need("lib"), phase="check", **GPU_LIB_ARGS)


@pytest.mark.gpu
def test_poc_form_shims_gpu():
    run_poc_form(None, _need("poc"), phase="check", **GPU_POC_ARGS)


@pytest.mark.gpu
def test_array_forms_gpu():
    """4096 gate bootstraps behind the reference's struct types in ONE launch: identical to the one-by-one loop (checked by the
    driver), to the compiled reference (all 4096, where test_ref_batch's reference run is present) or the oracle (a sample of
    them), and >= 100 k bootstraps/s END TO END -- gather, PCIe both ways, launch, scatter -- where the loop gives ~400/s."""
    import test_ref_batch as RB
    d = _need_arr()
    lk, bk, ks, x = RB._inputs()
    stats, got, _ = run_array_form(None, d, n=RB.n, inputs=(bk, ks, x), phase="check")
    assert stats["count"] == RB.BATCH
    if os.path.exists(os.path.join(RB.RUN_DIR, "done")):
        outs = sorted(f for f in os.listdir(RB.RUN_DIR) if f.startswith("out_"))
        want = np.concatenate([np.fromfile(os.path.join(RB.RUN_DIR, f), np.int32) for f in outs]).reshape(RB.BATCH, RB.n + 1)
        assert np.array_equal(got, want), "tfhe_bootstrap_FFT_array vs the compiled reference, 4096 samples"
    else:
        for c in (0, 1, 63, 64, 2047, 4095):
            assert np.array_equal(got[c], O.bootstrap32(RB.N, bk, ks, RB.MU, x[c], RB.l, RB.Bgbit, RB.t, RB.bb)), c
    assert stats["array_bootstraps_per_s"] >= 100e3, stats
    assert stats["array_bootstraps_per_s"] > 50 * stats["loop_bootstraps_per_s"], stats
    # the same loop over a pool of two members sharing the one GPU (two host threads, two streams, two key copies): identical
    assert stats["pool_devices"] == 2 and stats["pool_identical_to_loop"] is True, stats
    # the reference's `#pragma omp parallel for` over ONE-SAMPLE calls (32 threads), coalesced by the shim into array launches:
    # a multiple of the serial loop's rate from an unmodified loop body
    assert stats["parallel_for_bootstraps_per_s"] > 5 * stats["loop_bootstraps_per_s"], stats
    import json
    s256 = json.loads(open(os.path.join(GPU_RUN_DIR, "arr256", "arr_stats.json")).read())
    assert s256["parallel_for_threads"] == 256 and s256["array_identical_to_loop"] is True
    assert s256["parallel_for_bootstraps_per_s"] > stats["parallel_for_bootstraps_per_s"], (s256, stats)  # more callers, larger launches


def _need_arr():
    d = os.path.join(GPU_RUN_DIR, "arr")
    assert os.path.exists(os.path.join(d, "arr_stats.json")), "the array-form driver did not run before the session's GPU tests"
    return d


REF_SRC = "/root/reference/circuit-bootstrapping/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference sources not present (GPU box)")
def test_shims_compile_against_reference_types():
    """include/tfhe_amd_compat.hpp instantiated on the reference's own poc_types.h (as its Makefile builds it: -DUSE_FFT)"""
    subprocess.check_call(["g++", "-std=gnu++11", "-DUSE_FFT", "-fsyntax-only", "-I" + REF_SRC, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "compat", "poc_types_check.cpp")])
