#!/usr/bin/env python3
"""One purpose: the two N=2048 Torus64 transform kernels (execute_reverse_torus64 / execute_direct_torus64) on one
batch, for profiling (run ON THE GPU BOX, directly after `rocprofv3 ... --`):

    python3 tools/fft_once.py --batch 32768 --reps 2

No child processes, no torch."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--N", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    N, B = a.N, a.batch
    eng = T.Engine(torus_bits=64, n=1, N=N, l=2, Bgbit=8, ks_t=0)
    rs = np.random.RandomState(4)
    i64 = eng.to_device(np.frombuffer(rs.bytes(B * N * 8), dtype=np.int64).reshape(B, N))
    lag, o64 = eng.alloc(B * N * 8), eng.alloc(B * N * 8)
    e0, e1, e2 = eng.event(), eng.event(), eng.event()
    out = []
    for _ in range(a.reps + 1):
        eng.record(e0)
        eng._chk(eng.lib.tfhe_amd_ifft_torus64(eng.ctx, lag.ptr, i64.ptr, B))
        eng.record(e1)
        eng._chk(eng.lib.tfhe_amd_fft_torus64(eng.ctx, o64.ptr, lag.ptr, B))
        eng.record(e2)
        eng.sync()
        out.append((eng.elapsed_ms(e0, e1), eng.elapsed_ms(e1, e2)))
    algo = B * N * 16
    print(json.dumps({"N": N, "batch": B, "algorithmic_bytes_per_launch": algo,
                      "execute_reverse_torus64_ms": [o[0] for o in out[1:]], "execute_direct_torus64_ms": [o[1] for o in out[1:]],
                      "frac_of_8TBps": [algo / (min(o[k] for o in out[1:]) * 1e-3) / 8e12 for k in (0, 1)]}))
    eng.close()


if __name__ == "__main__":
    main()
