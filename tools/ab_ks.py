#!/usr/bin/env python3
"""A/B timing of several builds of the engine library on the key-switch kernels (k_ks_mfma), interleaved ON ONE GPU BOX:

    python tools/ab_ks.py libA.so libB.so [...] [--rounds 5]

  gate      lweKeySwitch 8 x 2 bits, N = 1024 -> n = 630, 4096 samples        (k_ks_mfma<int32, 2, 1>)
  preks     the PoC's preKeySwitch 6 x 2 bits, 1024 -> 500, 1024 samples      (k_ks_mfma<int32, 2, 1>)
  privks    the PoC's circuitPrivKS 10 x 3 bits, LWE64(2049) -> TLWE32(2048), one 1.35 GB plane, 2048 samples
            (k_ks_mfma<int64, 3, 3>; the circuit bootstrap runs two such launches)
Prints per library the median / min HIP-event time per launch and whether outputs equal the first library's."""
import argparse
import ctypes as C
import importlib
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--privks-samples", type=int, default=2048)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    BC = importlib.import_module("bench_configs")
    rs = np.random.RandomState(9)
    ks_gate = rs.randint(-2 ** 31, 2 ** 31, size=(1024, 8, 4, 631)).astype(np.int32)
    ks_pre = rs.randint(-2 ** 31, 2 ** 31, size=(1024, 6, 4, 501)).astype(np.int32)
    x_gate = rs.randint(-2 ** 31, 2 ** 31, size=(4096, 1025)).astype(np.int32)
    n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21 = 500, 1024, 2048, 2, 8, 4, 9, 6, 2, 10, 3
    plane = BC.rand_bits(rs, ((N2 + 1) * t21 * (1 << bb21) * 2 * N1,), np.int32)
    Bp = a.privks_samples
    x_priv = BC.rand_bits(rs, (Bp, N2 + 1), np.int64)
    jobs = []
    for lib in a.libs:
        path = os.path.abspath(lib)
        eg = T.Engine(torus_bits=32, n=630, N=1024, l=2, Bgbit=10, ks_t=8, ks_basebit=2, lib_path=path)
        eg.load_keyswitch_key(ks_gate)
        ep = T.Engine(torus_bits=32, n=500, N=1024, l=2, Bgbit=8, ks_t=6, ks_basebit=2, ks_n_out=500, lib_path=path)
        ep.load_keyswitch_key(ks_pre)
        cb = T.CircuitBootstrap(n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, lib_path=path)
        cb._chk(cb.lib.tfhe_amd_cb_load_privks_plane(cb.cb, 0, T._np_ptr(plane)))
        jobs.append(dict(lib=lib, eg=eg, ep=ep, cb=cb, xg=eg.to_device(x_gate), og=eg.alloc(4096 * 631 * 4),
                         xp=ep.to_device(x_gate[:1024]), op=ep.alloc(1024 * 501 * 4),
                         xv=cb._dev(x_priv), ov=cb._dev(np.zeros((Bp, 2, N1), np.int32)),
                         ev=BC.Events(cb.lib, cb.ctx), t=dict(gate=[], preks=[], privks=[])))
    for r in range(a.rounds + 1):
        for j in (jobs if r % 2 == 0 else jobs[::-1]):  # alternate the list order
            eg, ep, cb, ev = j["eg"], j["ep"], j["cb"], j["ev"]
            e0, e1 = eg.event(), eg.event()
            eg.record(e0); eg._chk(eg.lib.tfhe_amd_keyswitch(eg.ctx, j["og"].ptr, j["xg"].ptr, 4096)); eg.record(e1)
            tg = eg.elapsed_ms(e0, e1)
            e0, e1 = ep.event(), ep.event()
            ep.record(e0); ep._chk(ep.lib.tfhe_amd_keyswitch(ep.ctx, j["op"].ptr, j["xp"].ptr, 1024)); ep.record(e1)
            tp = ep.elapsed_ms(e0, e1)
            f0, f1 = ev.event(), ev.event()
            ev.record(f0); cb._chk(cb.lib.tfhe_amd_privks(cb.cb, j["ov"], 0, j["xv"], Bp)); ev.record(f1)
            tv = ev.elapsed_ms(f0, f1)
            if r:
                j["t"]["gate"].append(tg); j["t"]["preks"].append(tp); j["t"]["privks"].append(tv)
    ref = None
    for j in jobs:
        outs = (j["og"].download(np.int32, (4096, 631)), j["op"].download(np.int32, (1024, 501)),
                j["cb"]._out(j["ov"], np.int32, (Bp, 2, N1)))
        same = "" if ref is None else "  outputs==first: %s" % all(bool(np.array_equal(x, y)) for x, y in zip(outs, ref))
        if ref is None:
            ref = outs
        t = j["t"]
        print("%-22s gate %.4f (min %.4f)  preks %.4f (min %.4f)  privks[%d] %.4f (min %.4f) ms%s" % (
            os.path.basename(j["lib"]), statistics.median(t["gate"]), min(t["gate"]), statistics.median(t["preks"]), min(t["preks"]),
            Bp, statistics.median(t["privks"]), min(t["privks"]), same), flush=True)


if __name__ == "__main__":
    main()
