"""BASELINE configs 2 and 4 at full size against the compiled reference.  Config 2: every output of a 4096-sample batch of gate
bootstraps (n=630, N=1024, l=2, key switch 8x2) from the HIP engine, compared bit for bit with
`oracle/_ref/ref_driver boot32` -- tfhe_bootstrap_FFT composed from the reference's own FFT / AddMul object code
(CB/spqlios/*.s, fft_processor_spqlios.cpp) -- run by one CPU process per host core.

Config 4: the four `execute_*` conversions at N=2048 on a batch of 8192 polynomials against `ref_driver rev_int /
rev_t64 / dir_t32 / dir_t64` (FFT_Processor_Spqlios on the reference's assembly transforms), every polynomial.

The reference processes are children: they run in conftest.pytest_collection_finish (prerun_gpu_drivers), before
this process touches the GPU; the test itself only loads their outputs.  oracle/_ref/ travels to the GPU box
prebuilt (it cannot be rebuilt there: /root/reference is absent)."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RUN_DIR = os.path.join(ROOT, "build", "ref_batch_run")
N, n, l, Bgbit, t, bb = 1024, 630, 2, 10, 8, 2
SEED = 0x5446484500000001
MU = 1 << 29
BATCH = 4096


def _inputs():
    lk, tk = O.keygen_binary(n, SEED, 1), O.keygen_binary(N, SEED, 2)
    bk = O.bk_create32(N, lk, tk, l, Bgbit, 2.0 ** -25, SEED, 1000)
    ks = O.ks_create32(tk, lk, t, bb, 2.0 ** -15, SEED, 100000)
    # 64 real encryptions of +-mu, the rest uniformly random samples ('synthetic random ciphertexts')
    real = np.stack([O.lwe_encrypt32(MU if i % 2 else -MU, 2.0 ** -15, lk, O.rng(SEED, 70 + i)) for i in range(64)])
    rs = np.random.RandomState(4096)
    x = np.concatenate([real, rs.randint(-2 ** 31, 2 ** 31, size=(BATCH - 64, n + 1), dtype=np.int64).astype(np.int32)])
    return lk, bk, ks, x


FFT_N, FFT_B = 2048, 8192


def _fft_inputs():
    rs = np.random.RandomState(2048)
    dig = rs.randint(-256, 256, size=(FFT_B, FFT_N)).astype(np.int32)              # gadget digits (Bgbit2 = 9)
    a64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(FFT_B, FFT_N), dtype=np.int64)   # Torus64 polynomials
    return dig, a64


def _ref_file(op, src, dst, *args):
    rc = subprocess.call([O.REF_DRIVER, op, os.path.join(RUN_DIR, src), os.path.join(RUN_DIR, dst)] + [str(a) for a in args])
    if rc:
        raise RuntimeError("ref_driver %s failed (%d)" % (op, rc))


def _prerun_fft():
    dig, a64 = _fft_inputs()
    dig.tofile(os.path.join(RUN_DIR, "fft_dig.bin"))
    a64.tofile(os.path.join(RUN_DIR, "fft_a64.bin"))
    _ref_file("rev_int", "fft_dig.bin", "fft_lag_dig.bin", FFT_N)       # execute_reverse_int
    _ref_file("rev_t64", "fft_a64.bin", "fft_lag_a64.bin", FFT_N)       # execute_reverse_torus64
    _ref_file("dir_t64", "fft_lag_a64.bin", "fft_t64.bin", FFT_N)       # execute_direct_torus64
    _ref_file("dir_t32", "fft_lag_dig.bin", "fft_t32.bin", FFT_N)       # execute_direct_torus32
    for f in ("fft_dig.bin", "fft_a64.bin"):
        os.remove(os.path.join(RUN_DIR, f))
    with open(os.path.join(RUN_DIR, "fft_done"), "w") as f:
        f.write("ok\n")


def prerun_gpu_drivers():
    """conftest.pytest_collection_finish, on a GPU box, before this process initialises the GPU"""
    if not O.have_ref():
        return
    os.makedirs(RUN_DIR, exist_ok=True)
    for f in os.listdir(RUN_DIR):
        os.remove(os.path.join(RUN_DIR, f))
    _prerun_fft()
    lk, bk, ks, x = _inputs()
    fin = os.path.join(RUN_DIR, "in.bin")
    with open(fin, "wb") as f:
        f.write(np.array([MU, 0], np.int32).tobytes())
        f.write(np.ascontiguousarray(bk, np.float64).tobytes())
        f.write(np.ascontiguousarray(ks, np.int32).tobytes())
        f.write(x.tobytes())
    # one single-threaded process per core, ~260 MB each while it loads the keys; never more than a quarter of
    # the free memory (the rule bench.py's CPU baseline follows)
    procs = min(os.cpu_count() or 1, 64)
    try:
        avail_kb = next(int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable"))
        procs = max(1, min(procs, int(0.25 * avail_kb * 1024 // (300 << 20))))
    except (OSError, StopIteration, ValueError):
        procs = min(procs, 8)
    running = []
    for r in range(procs):
        lo, hi = BATCH * r // procs, BATCH * (r + 1) // procs
        if hi > lo:
            running.append(subprocess.Popen([O.REF_DRIVER, "boot32", fin, os.path.join(RUN_DIR, "out_%03d.bin" % r), str(n), str(l),
                                             str(Bgbit), str(t), str(bb), str(hi - lo), str(lo)], stdout=subprocess.DEVNULL))
    rc = [p.wait() for p in running]
    os.remove(fin)
    if any(rc):
        raise RuntimeError("ref_driver boot32 failed: %s" % rc)
    np.save(os.path.join(RUN_DIR, "x.npy"), x)
    with open(os.path.join(RUN_DIR, "done"), "w") as f:
        f.write("%d processes\n" % len(running))


@pytest.mark.gpu
def test_full_batch_bit_identical_to_compiled_reference(gpu_lib):
    if not O.have_ref():
        pytest.skip("oracle/_ref/ref_driver not present")
    assert os.path.exists(os.path.join(RUN_DIR, "done")), "the reference processes did not run before the session's GPU tests"
    outs = sorted(f for f in os.listdir(RUN_DIR) if f.startswith("out_"))
    want = np.concatenate([np.fromfile(os.path.join(RUN_DIR, f), np.int32) for f in outs]).reshape(BATCH, n + 1)
    x = np.load(os.path.join(RUN_DIR, "x.npy"))
    lk, bk, ks, x2 = _inputs()
    assert np.array_equal(x, x2)
    T = importlib.import_module("experimental-tfhe_amd")
    e = T.Engine(torus_bits=32, n=n, N=N, l=l, Bgbit=Bgbit, ks_t=t, ks_basebit=bb, lib_path=gpu_lib)
    try:
        e.set_bootstrap_key(e.gsw_from_fft(bk))
        e.load_keyswitch_key(ks)
        got = e.bootstrap(MU, x)
    finally:
        e.close()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "%d of %d bootstraps differ from the compiled reference (first: %s)" % (bad.size, BATCH, bad[:8])
    # and the 64 real encryptions decrypt to their messages
    assert all((O.lwe_phase32(got[i], lk) > 0) == bool(i % 2) for i in range(64))


@pytest.mark.gpu
def test_n2048_transforms_batch_8192_bit_identical_to_compiled_reference(gpu_lib):
    """BASELINE config 4: all 8192 polynomials of each conversion against FFT_Processor_Spqlios itself"""
    if not O.have_ref():
        pytest.skip("oracle/_ref/ref_driver not present")
    assert os.path.exists(os.path.join(RUN_DIR, "fft_done")), "the reference processes did not run before the session's GPU tests"
    import parity_checks as P
    dig, a64 = _fft_inputs()
    load = lambda name, dt: np.fromfile(os.path.join(RUN_DIR, name), dt).reshape(FFT_B, FFT_N)
    T = importlib.import_module("experimental-tfhe_amd")
    e = T.Engine(torus_bits=64, n=1, N=FFT_N, l=4, Bgbit=9, ks_t=0, lib_path=gpu_lib)
    try:
        lag_dig = e.ifft_int32(dig)
        assert P.same_doubles(lag_dig, load("fft_lag_dig.bin", np.float64)), "execute_reverse_int"
        lag_a64 = e.ifft_torus64(a64)
        assert P.same_doubles(lag_a64, load("fft_lag_a64.bin", np.float64)), "execute_reverse_torus64"
        assert np.array_equal(e.fft_torus64(lag_a64), load("fft_t64.bin", np.int64)), "execute_direct_torus64"
        assert np.array_equal(e.fft_torus32(lag_dig), load("fft_t32.bin", np.int32)), "execute_direct_torus32"
    finally:
        e.close()


if __name__ == "__main__":  # python tests/test_ref_batch.py prerun   (timing of the reference side on this host)
    import time
    t0 = time.time()
    prerun_gpu_drivers()
    print("reference side: %.1f s" % (time.time() - t0))
    sys.exit(0)
