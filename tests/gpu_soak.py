#!/usr/bin/env python3
"""Randomised soak of the GPU parity checks (not collected by pytest): the checks of tests/parity_checks.py over
many seeds, batch sizes (ragged workgroups, waves without a partner on their SIMD) and gadget shapes, each compared
bit for bit with the oracle.  Run ON A GPU BOX:

    python tests/gpu_soak.py [--minutes 5] [--seed 1]

Everything it needs (oracle library, engine library) is built before the first engine context exists."""
import argparse
import importlib
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--device", type=int, default=0, help="device ordinal of the pool members of the `pool` checks")
    ap.add_argument("--lib", default=None, help="engine library (default: the shipped HIP build; the CPU emulation build for a dry run)")
    a = ap.parse_args()
    import oracle_py as O
    O.lib()  # compile / load the oracle now: no child process once the GPU is in use
    T = importlib.import_module("experimental-tfhe_amd")
    if not os.path.exists(T.DEFAULT_LIB):
        importlib.import_module("experimental-tfhe_amd.build").build()
    import parity_checks as P
    lib = a.lib or T.DEFAULT_LIB
    rnd = random.Random(a.seed)
    t_end = time.time() + 60.0 * a.minutes
    runs = 0
    while time.time() < t_end:
        seed = rnd.randrange(1, 1 << 30)
        kind = rnd.choice(["gate", "gate", "gate", "t64", "ks", "cmux", "lut", "fft", "exact", "wide", "pool", "privks", "ring", "ring"])
        if kind == "gate":      # N=1024 Torus32: compile-time and run-time gadgets, ragged batches 1..41
            l, bg = rnd.choice([(2, 10), (2, 10), (2, 8), (2, 9), (3, 7), (4, 6), (1, 12)])
            t, bb = rnd.choice([(8, 2), (16, 1), (5, 3), (6, 2)])
            # which blind-rotation kernel: the library's choice, the latency-shaped one (it applies where l == 2), or
            # one wave per ciphertext whatever the batch
            P.check_gate_path(lib, N=1024, n=rnd.randrange(2, 9), l=l, Bgbit=bg, ks_t=t, ks_bb=bb, B=rnd.randrange(1, 42),
                              seed=seed, check_export=False, br_split=rnd.choice([None, 1 << 30, 0]))
        elif kind == "ring":    # ring degrees other than 1024 / 2048 (tfhe_kernels_generic.h): every path that takes an N
            sub = rnd.choice(["gate", "gate", "t64", "fft", "consumers"])
            if sub == "gate":
                N = rnd.choice([16, 32, 64, 128, 256, 512, 512, 4096, 8192])
                l, bg = rnd.choice([(2, 10), (3, 6), (1, 12), (4, 5)]) if N >= 64 else rnd.choice([(2, 8), (3, 6)])
                big = N >= 4096
                P.check_gate_path(lib, N=N, n=rnd.randrange(2, 4 if big else 8), l=l, Bgbit=bg, ks_t=rnd.choice([3, 4, 8]), ks_bb=rnd.choice([1, 2]),
                                  B=rnd.randrange(1, 4 if big else 30), seed=seed, check_export=rnd.random() < 0.3)
            elif sub == "t64":
                N = rnd.choice([64, 256, 512, 4096])
                l, bg = rnd.choice([(4, 9), (3, 10), (2, 16)])
                P.check_torus64_path(lib, N=N, n=rnd.randrange(2, 5), l=l, Bgbit=bg, B=rnd.randrange(1, 4 if N >= 4096 else 10), seed=seed)
            elif sub == "fft":
                N = 1 << rnd.choice([4, 5, 6, 7, 8, 9, 12, 13, 14, 15])
                P.check_fft_plugin(lib, N, count=rnd.randrange(1, 6 if N >= 4096 else 300), seed=seed)
            else:
                N = rnd.choice([256, 512])
                P.check_cmux_data(lib, N=N, B=rnd.randrange(1, 20), seed=seed)
                P.check_lut_eval(lib, N=N, d=rnd.randrange(1, 12), B=rnd.randrange(1, 4), seed=seed, decrypt_tol=None)
        elif kind == "wide":    # batches above 1024: the 8-wave workgroup instantiations of every gadget class
            l, bg = rnd.choice([(2, 10), (2, 8), (2, 9), (3, 7), (4, 6)])
            P.check_gate_wide_batch(lib, l=l, Bgbit=bg, B=rnd.randrange(1025, 1100), n=rnd.randrange(2, 7), seed=seed)
        elif kind == "t64":     # Torus64, both ring sizes (short rounding + guard)
            N = rnd.choice([1024, 2048])
            l, bg = rnd.choice([(4, 9), (3, 10), (2, 16), (4, 8)])
            P.check_torus64_path(lib, N=N, n=rnd.randrange(2, 5), l=l, Bgbit=bg, B=rnd.randrange(1, 10), seed=seed)
        elif kind == "ks":      # matrix-core key switch and its fallbacks, ragged sample tiles
            t, bb = rnd.choice([(8, 2), (6, 2), (16, 1), (10, 3), (5, 3), (15, 2)])
            P.check_keyswitch_shapes(lib, 1024, rnd.choice([500, 630, 37]), t, bb, rnd.randrange(1, 300), seed=seed)
        elif kind == "pool":    # host arrays through a pool of 1..3 members sharing the device, pipelined chunks of every size
            members = rnd.randrange(1, 4)
            P.check_pool(lib, [a.device] * members, count=rnd.randrange(1, 2600), chunk=rnd.choice([0, 1, 7, 64, 256, 1000, 2048]),
                         n=rnd.randrange(2, 6), seed=seed % 100000)
        elif kind == "privks":  # the int64 key switch with many samples per launch (several 256-sample tiles), small tables
            t21, bb21 = rnd.choice([(2, 3), (3, 3), (4, 2), (6, 1)])
            P.check_privks_wide(lib, N2=1024, t21=t21, bb21=bb21, counts=(rnd.randrange(257, 1100),), pipeline_B=rnd.choice([0, 0, 130]),
                                n0=2, l2=3, bg2=10, t10=2, seed=seed % 100000)
        elif kind == "cmux":
            P.check_cmux_data(lib, B=rnd.randrange(1, 30), seed=seed)
        elif kind == "lut":
            P.check_lut_eval(lib, d=rnd.randrange(1, 14), B=rnd.randrange(1, 6), seed=seed, decrypt_tol=None)
        elif kind == "fft":
            P.check_fft_plugin(lib, rnd.choice([1024, 2048]), count=rnd.randrange(1, 40), seed=seed)
        else:
            bits, N, l, bg, bound = rnd.choice([(32, 1024, 2, 10, 4), (64, 2048, 4, 9, 2 ** 32), (64, 1024, 3, 10, 2 ** 32)])
            P.check_exact_extprod(lib, bits, N, l, bg, B=rnd.randrange(1, 4), seed=seed, fft_bound=bound)
        runs += 1
        if runs % 10 == 0:
            print("%4d checks, last: %s seed %d" % (runs, kind, seed), flush=True)
    print("soak finished: %d randomised checks, all bit-identical to the oracle" % runs, flush=True)


if __name__ == "__main__":
    main()
