#!/usr/bin/env python3
"""Host-array rate of tfhe_amd_cb_pool_circuit_bootstrap_host (PCIe included: 4 KB in, 16 KB out per input) at the PoC's
parameters against the chunk size of the pool's pipelined member, beside the device-resident rate (run on the GPU box):

    python tools/cb_pool_rate.py [--samples 4096] [--chunks 0,1024,2048]

Synthetic (uniformly random) keys: throughput does not depend on key contents.  One JSON object per line."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--chunks", default="0,1024,2048")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    from bench_configs import rand_bits
    n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21 = 500, 1024, 2048, 2, 8, 4, 9, 6, 2, 10, 3
    rs = np.random.RandomState(5)
    pool = T.CircuitBootstrapPool([0], n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, lib_path=a.lib)
    lib = pool.lib
    pool.load_preks(rand_bits(rs, (N1, t10, 1 << bb10, n0 + 1), np.int32))
    bk = rand_bits(rs, (n0, 2 * l2, 2, N2), np.int64)
    assert lib.tfhe_amd_cb_pool_load_bk_torus(pool.pool, T._np_ptr(np.ascontiguousarray(bk))) == 0
    for u in range(2):
        plane = rand_bits(rs, ((N2 + 1) * t21 * (1 << bb21) * 2 * N1,), np.int32)
        assert lib.tfhe_amd_cb_pool_load_privks_plane(pool.pool, u, T._np_ptr(plane)) == 0
        del plane
    x = rand_bits(rs, (a.samples, N1 + 1), np.int32)
    # device-resident reference: 1024 inputs per launch on the member's own handle
    cb = C.c_void_p(lib.tfhe_amd_cb_pool_member(pool.pool, 0))
    ctx = C.c_void_p(lib.tfhe_amd_cb_ctx_lvl2(cb))
    d_x, d_o = C.c_void_p(), C.c_void_p()
    assert lib.tfhe_amd_malloc(ctx, C.byref(d_x), 1024 * (N1 + 1) * 4) == 0 and lib.tfhe_amd_malloc(ctx, C.byref(d_o), 1024 * 2 * l1 * 2 * N1 * 4) == 0
    assert lib.tfhe_amd_memcpy_h2d(ctx, d_x, T._np_ptr(np.ascontiguousarray(x[:1024])), 1024 * (N1 + 1) * 4) == 0
    assert lib.tfhe_amd_circuit_bootstrap(cb, d_o, d_x, 1024) == 0 and lib.tfhe_amd_cb_sync(cb) == 0
    t0 = time.perf_counter()
    for _ in range(3):
        assert lib.tfhe_amd_circuit_bootstrap(cb, d_o, d_x, 1024) == 0
    assert lib.tfhe_amd_cb_sync(cb) == 0
    resident = 3 * 1024 / (time.perf_counter() - t0)
    print(json.dumps({"device_resident_circuit_bootstraps_per_s": resident}), flush=True)
    first = None
    for chunk in [int(v) for v in a.chunks.split(",")]:
        pool.set_chunk_rows(chunk)
        got = pool.circuit_bootstrap(x)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            got = pool.circuit_bootstrap(x)
            ts.append(time.perf_counter() - t0)
        if first is None:
            first = got
        print(json.dumps({"samples_per_call": a.samples, "chunk_rows": chunk, "ms_per_call_min": 1e3 * min(ts),
                          "circuit_bootstraps_per_s": a.samples / min(ts), "over_device_resident": a.samples / min(ts) / resident,
                          "identical_to_first_chunking": bool(np.array_equal(got, first))}), flush=True)
    pool.close()


if __name__ == "__main__":
    main()
