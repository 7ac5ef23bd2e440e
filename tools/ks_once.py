#!/usr/bin/env python3
"""One purpose: the private key switch alone (circuitPrivKS 10 x 3 bits on one 1.35 GB plane, k_ks_mfma<int64,3,3>) for profiling
(run ON THE GPU BOX, directly after `rocprofv3 ... --`):

    python3 tools/ks_once.py --samples 2048 --reps 3 [--lib other.so]

No child processes, no torch.  Prints HIP-event times per launch."""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    BC = importlib.import_module("bench_configs")
    rs = np.random.RandomState(9)
    n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21 = 500, 1024, 2048, 2, 8, 4, 9, 6, 2, 10, 3
    cb = T.CircuitBootstrap(n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, lib_path=a.lib)
    plane = BC.rand_bits(rs, ((N2 + 1) * t21 * (1 << bb21) * 2 * N1,), np.int32)
    cb._chk(cb.lib.tfhe_amd_cb_load_privks_plane(cb.cb, 0, T._np_ptr(plane)))
    x = cb._dev(BC.rand_bits(rs, (a.samples, N2 + 1), np.int64))
    o = cb._dev(np.zeros((a.samples, 2, N1), np.int32))
    ev = BC.Events(cb.lib, cb.ctx)
    e0, e1 = ev.event(), ev.event()
    ts = []
    for _ in range(a.reps + 1):
        ev.record(e0)
        cb._chk(cb.lib.tfhe_amd_privks(cb.cb, o, 0, x, a.samples))
        ev.record(e1)
        ts.append(ev.elapsed_ms(e0, e1))
    macs = a.samples * 6150 * 32 * 2048 * 4  # K-steps (padded) x 32 x output columns x 4 byte limbs
    print(json.dumps({"workload": f"circuitPrivKS 10x3, one plane, {a.samples} samples", "ms": ts[1:],
                      "int8_mac_per_s": macs / (min(ts[1:]) * 1e-3), "frac_of_2.5e15": macs / (min(ts[1:]) * 1e-3) / 2.5e15}))
    cb.close()


if __name__ == "__main__":
    main()
