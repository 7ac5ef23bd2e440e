#!/usr/bin/env python3
"""A/B timing of several builds of the engine library on the circuit bootstrap's Torus64 blind rotation
(circuitBootstrapWoKS, poc:530-659: n0 = 500, N2 = 2048, l2 = 4, Bgbit2 = 9), interleaved ON ONE GPU BOX:

    python tools/ab_cb.py libA.so libB.so [...] [--batch 1024] [--rounds 4] [--n0 500]

Prints per library the median / min HIP-event time of one blind rotation of the batch and whether its outputs equal
the first library's (experiments must not change a bit)."""
import argparse
import importlib
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--n0", type=int, default=500)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    rs = np.random.RandomState(5)
    n0, N2, l2, bg2 = a.n0, 2048, 4, 9
    bk = np.frombuffer(rs.bytes(n0 * 2 * l2 * 2 * N2 * 8), dtype=np.int64).reshape(n0, 2 * l2, 2, N2)
    abar = rs.randint(0, 2 * N2, size=(a.batch, n0 + 1)).astype(np.int32)
    jobs = []
    for lib in a.libs:
        eng = T.Engine(torus_bits=64, n=n0, N=N2, l=l2, Bgbit=bg2, ks_t=0, lib_path=os.path.abspath(lib))
        eng.set_bootstrap_key(eng.gsw_from_torus(bk))
        jobs.append((lib, eng, eng.to_device(abar), eng.alloc(a.batch * (N2 + 1) * 8), [eng.event(), eng.event()], []))
    for r in range(a.rounds + 1):
        for lib, eng, d_abar, d_out, ev, ts in (jobs if r % 2 == 0 else jobs[::-1]):  # alternate the list order
            eng.record(ev[0])
            eng._chk(eng.lib.tfhe_amd_cb_bootstrap_woks(eng.ctx, d_out.ptr, 1 << 55, d_abar.ptr, a.batch))
            eng.record(ev[1])
            eng.sync()
            if r:
                ts.append(eng.elapsed_ms(ev[0], ev[1]))
    ref = None
    for lib, eng, d_abar, d_out, ev, ts in jobs:
        out = d_out.download(np.int64, (a.batch, N2 + 1))
        same = "" if ref is None else "  outputs==first: %s" % bool(np.array_equal(out, ref))
        if ref is None:
            ref = out
        med = statistics.median(ts)
        print("%-28s BR64 median %.3f min %.3f ms  -> %.2f M CMux/s%s" % (os.path.basename(lib), med, min(ts),
                                                                         a.batch * n0 / med / 1e3, same), flush=True)
    for j in jobs:
        j[1].close()


if __name__ == "__main__":
    main()
