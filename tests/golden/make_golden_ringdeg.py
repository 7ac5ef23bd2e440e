"""Generate tests/golden/ref_ringdeg.npz / .json from the COMPILED REFERENCE (oracle/_ref/ref_driver) for ring degrees
the reference accepts but never instantiates itself: FFT_Processor_Spqlios(N) / new_fft_table(N) take every power of two
>= 16 (CB/spqlios/fft_processor_spqlios.cpp:18-25, spqlios-fft-impl.cpp:157-160,400-403).  Run in the build container:

    python tests/golden/make_golden_ringdeg.py

Kept apart from make_golden.py so that the N = 1024 / 2048 fixture (and the numpy stream it was drawn from) stays as it is.
Every array is an input drawn here from a fixed numpy seed or the output the reference's object code produced for it;
for N = 8192 only SHA-256 digests of the outputs are kept (inputs are redrawn from the recorded seed by the test).
The gate-bootstrap vectors come from `ref_driver boot32` (tfhe_bootstrap_FFT composed from the reference's FFT / AddMul
object code, integer glue per CB/lwe_functions.cpp:136-171,328-446) on keys the ORACLE's seeded generator makes: the test
regenerates the keys, so only inputs and outputs are stored.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_py as O  # noqa: E402

RING_DEGREES = (16, 64, 512, 4096, 8192)
HASH_ONLY = (8192,)
NUMPY_SEED = 20261004
# gate bootstraps: (N, n, l, Bgbit, ks_t, ks_bb, count); keys: oracle generators with these seeds/streams (tests/parity_checks.GateSetup)
GATE_SETS = ((512, 12, 2, 10, 4, 2, 4), (4096, 5, 2, 10, 3, 2, 2), (16, 6, 3, 6, 5, 2, 4))
KEY_SEED = 0x5446484500000001
MU = 1 << 29


def draw(rs, N, cnt):
    a32 = rs.randint(-2 ** 31, 2 ** 31, size=(cnt, N)).astype(np.int32)
    dig = rs.randint(-512, 512, size=(cnt, N)).astype(np.int32)
    a64 = rs.randint(-2 ** 63, 2 ** 63 - 1, size=(cnt, N), dtype=np.int64)
    raw = rs.randn(cnt, N) * 1e6
    return a32, dig, a64, raw


def reference_outputs(N, a32, dig, a64, raw):
    cnt = a32.shape[0]
    o = {}
    o["rev_int_a32"] = O.ref("rev_int", a32, np.float64, N).reshape(cnt, N)
    o["rev_int_dig"] = O.ref("rev_int", dig, np.float64, N).reshape(cnt, N)
    o["rev_t64"] = O.ref("rev_t64", a64, np.float64, N).reshape(cnt, N)
    z = np.zeros((cnt, N))
    o["addmul32"] = O.ref("addmul", np.concatenate([z, o["rev_int_dig"], o["rev_int_a32"]], axis=1), np.float64, N).reshape(cnt, N)
    o["addmul64"] = O.ref("addmul", np.concatenate([z, o["rev_int_dig"], o["rev_t64"]], axis=1), np.float64, N).reshape(cnt, N)
    o["dir_t32"] = O.ref("dir_t32", o["addmul32"], np.int32, N).reshape(cnt, N)
    o["dir_t64"] = O.ref("dir_t64", o["addmul64"], np.int64, N).reshape(cnt, N)
    o["raw_ifft"] = O.ref("ifft", raw, np.float64, N).reshape(cnt, N)
    o["raw_fft"] = O.ref("fft", raw, np.float64, N).reshape(cnt, N)
    return o


def gate_keys(N, n, l, Bgbit, t, bb):
    lk, tk = O.keygen_binary(n, KEY_SEED, 1), O.keygen_binary(N, KEY_SEED, 2)
    bk = O.bk_create32(N, lk, tk, l, Bgbit, 2.0 ** -25, KEY_SEED, 1000)
    ks = O.ks_create32(tk, lk, t, bb, 2.0 ** -15, KEY_SEED, 100000)
    return lk, bk, ks


def main():
    assert O.have_ref(), "build oracle/_ref first: make -C oracle ref"
    out = {}
    meta = {"generator": "tests/golden/make_golden_ringdeg.py", "numpy_seed": NUMPY_SEED, "ring_degrees": list(RING_DEGREES),
            "hash_only": list(HASH_ONLY), "table_sha256": {}, "output_sha256": {}, "gate_sets": [list(g) for g in GATE_SETS],
            "key_seed": KEY_SEED, "mu": MU}
    for N in RING_DEGREES:
        L = 2 * N - 8
        t = O.ref("tables", b"", np.float64, N)
        meta["table_sha256"][f"fft_trig_{N}"] = hashlib.sha256(t[:L].tobytes()).hexdigest()
        meta["table_sha256"][f"ifft_trig_{N}"] = hashlib.sha256(t[L:].tobytes()).hexdigest()
        rs = np.random.RandomState(NUMPY_SEED + N)  # one stream per ring degree: a test redraws them independently
        cnt = 2 if N <= 512 else 1
        a32, dig, a64, raw = draw(rs, N, cnt)
        o = reference_outputs(N, a32, dig, a64, raw)
        if N in HASH_ONLY:
            meta["output_sha256"][str(N)] = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in o.items()}
            continue
        out[f"a32_{N}"], out[f"dig_{N}"], out[f"a64_{N}"], out[f"raw_in_{N}"] = a32, dig, a64, raw
        for k, v in o.items():
            out[f"{k}_{N}"] = v
    for (N, n, l, Bgbit, t, bb, count) in GATE_SETS:
        lk, bk, ks = gate_keys(N, n, l, Bgbit, t, bb)
        rs = np.random.RandomState(NUMPY_SEED + 7 * N)
        real = np.stack([O.lwe_encrypt32(MU if i % 2 else -MU, 2.0 ** -15, lk, O.rng(KEY_SEED, 70 + i)) for i in range(count // 2)])
        x = np.concatenate([real, rs.randint(-2 ** 31, 2 ** 31, size=(count - count // 2, n + 1), dtype=np.int64).astype(np.int32)])
        buf = np.array([MU, 0], np.int32).tobytes() + np.ascontiguousarray(bk, np.float64).tobytes() + \
            np.ascontiguousarray(ks, np.int32).tobytes() + x.tobytes()
        out[f"boot32_x_{N}"] = x
        out[f"boot32_out_{N}"] = O.ref("boot32", buf, np.int32, n, l, Bgbit, t, bb, count, 0, N).reshape(count, n + 1)
    np.savez_compressed(os.path.join(HERE, "ref_ringdeg.npz"), **out)
    with open(os.path.join(HERE, "ref_ringdeg.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", os.path.join(HERE, "ref_ringdeg.npz"), os.path.getsize(os.path.join(HERE, "ref_ringdeg.npz")), "bytes")


if __name__ == "__main__":
    main()
