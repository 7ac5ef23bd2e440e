#!/usr/bin/env python3
"""One purpose: the circuit bootstrap's Torus64 blind rotation alone (circuitBootstrapWoKS, poc:530-659: n0 = 500,
N2 = 2048, l2 = 4, Bgbit2 = 9) on a synthetic key, for timing and profiling (run ON THE GPU BOX, directly after
`rocprofv3 ... --`):

    python3 tools/cb_br_once.py --batch 1024 --reps 3 [--lib other.so]

No child processes, no torch.  Prints HIP-event times per call and CMux/s."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FP64_INSTR_PER_CMUX = 8064  # wave64 fp64 instructions per Torus64 / N = 2048 / l = 4 CMux (one wave per ciphertext form)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--n0", type=int, default=500)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    rs = np.random.RandomState(5)
    n0, N2, l2, bg2 = a.n0, 2048, 4, 9
    eng = T.Engine(torus_bits=64, n=n0, N=N2, l=l2, Bgbit=bg2, ks_t=0, lib_path=a.lib)
    bk = np.frombuffer(rs.bytes(n0 * 2 * l2 * 2 * N2 * 8), dtype=np.int64).reshape(n0, 2 * l2, 2, N2)
    eng.set_bootstrap_key(eng.gsw_from_torus(bk))  # tGswToFFTConvert on the GPU
    abar = rs.randint(0, 2 * N2, size=(a.batch, n0 + 1)).astype(np.int32)
    d_abar = eng.to_device(abar)
    d_out = eng.alloc(a.batch * (N2 + 1) * 8)
    e0, e1 = eng.event(), eng.event()
    ts = []
    for _ in range(a.reps + 1):
        eng.record(e0)
        eng._chk(eng.lib.tfhe_amd_cb_bootstrap_woks(eng.ctx, d_out.ptr, 1 << 55, d_abar.ptr, a.batch))
        eng.record(e1)
        eng.sync()
        ts.append(eng.elapsed_ms(e0, e1))
    best = min(ts[1:])
    cmux = a.batch * n0 / (best * 1e-3)
    print(json.dumps({"workload": f"circuitBootstrapWoKS n0={n0} N2=2048 l2=4 Bgbit2=9 batch {a.batch}", "ms": ts[1:],
                      "cmux_per_s": cmux,
                      "fp64_issue_frac": cmux * FP64_INSTR_PER_CMUX * 4 / (1024 * 2.4e9)}))
    eng.close()


if __name__ == "__main__":
    main()
