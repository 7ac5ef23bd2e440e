#!/usr/bin/env python3
"""A/B timing of several builds of the engine library ON ONE GPU BOX, interleaved (boxes differ by
several percent among themselves, so two gpurun calls cannot be compared):

    python tools/ab.py libA.so libB.so [...] [--batch 4096] [--rounds 6]

Prints per library the median / min of the blind-rotation and key-switch kernel times (HIP events)."""
import argparse
import importlib
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig()
    jobs = []
    for lib in a.libs:
        job = shard.GateJob(cfg, 0x5446484500000001, device=0, lib_path=os.path.abspath(lib))
        eng = job.eng
        x_d = eng.to_device(shard.synthetic_samples(cfg, a.batch, seed=7))
        u_d = eng.alloc(a.batch * (cfg.N + 1) * 4)
        o_d = eng.alloc(a.batch * (cfg.n + 1) * 4)
        jobs.append((lib, job, eng, x_d, u_d, o_d, [eng.event() for _ in range(3)], [], []))
    ref = None
    for r in range(a.rounds + 1):
        # list order forwards and backwards in alternate rounds: a build is not always measured behind the same neighbour
        for lib, job, eng, x_d, u_d, o_d, ev, br, ks in (jobs if r % 2 == 0 else jobs[::-1]):
            eng.record(ev[0])
            eng._chk(eng.lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, 1 << 29, x_d.ptr, a.batch))
            eng.record(ev[1])
            eng._chk(eng.lib.tfhe_amd_keyswitch(eng.ctx, o_d.ptr, u_d.ptr, a.batch))
            eng.record(ev[2])
            eng.sync()
            if r:  # round 0 warms up
                br.append(eng.elapsed_ms(ev[0], ev[1]))
                ks.append(eng.elapsed_ms(ev[1], ev[2]))
    import numpy as np
    for lib, job, eng, x_d, u_d, o_d, ev, br, ks in jobs:
        out = o_d.download(np.int32, (a.batch, cfg.n + 1))
        same = "" if ref is None else ("  outputs==first: %s" % bool(np.array_equal(out, ref)))
        if ref is None:
            ref = out
        print("%-40s BR median %.3f min %.3f ms   KS median %.3f ms   -> %.0f bootstraps/s%s" % (
            os.path.basename(lib), statistics.median(br), min(br), statistics.median(ks),
            a.batch / (statistics.median(br) + statistics.median(ks)) * 1e3, same), flush=True)
    for j in jobs:
        j[1].close()


if __name__ == "__main__":
    main()
