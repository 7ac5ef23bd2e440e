#!/usr/bin/env python3
"""A/B of the one-launch-per-CMux schedule (tfhe_amd_bootstrap_streamed, BASELINE config 2 as worded) between builds of the
engine library ON ONE GPU BOX, interleaved:

    python tools/ab_streamed.py libA.so libB.so [...] [--batch 4096] [--rounds 5]

Per library: median / min time of the whole schedule (plain launches and hipGraph replay), microseconds per external-product
launch, fraction of 8 TB/s on SURVEY 8(d)'s bytes (batch x 16,388 + 65,536 per launch), outputs compared with the first library's."""
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
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig()
    B = a.batch
    jobs = []
    for lib in a.libs:
        job = shard.GateJob(cfg, 0x5446484500000001, device=0, lib_path=os.path.abspath(lib))
        eng = job.eng
        jobs.append(dict(lib=lib, job=job, eng=eng, x=eng.to_device(shard.synthetic_samples(cfg, B, seed=7)),
                         u=eng.alloc(B * (cfg.N + 1) * 4), o=eng.alloc(B * (cfg.n + 1) * 4), ev=[eng.event() for _ in range(2)],
                         plain=[], graph=[], ks=[]))
    for mode in ("plain", "graph"):
        for j in jobs:
            j["eng"].set_option(T.OPT_STREAMED_GRAPH, 1 if mode == "graph" else 0)
        for r in range(a.rounds + 2):
            for j in (jobs if r % 2 == 0 else jobs[::-1]):
                eng, ev = j["eng"], j["ev"]
                eng.record(ev[0])
                eng._chk(eng.lib.tfhe_amd_bootstrap_streamed(eng.ctx, j["o"].ptr, 1 << 29, j["x"].ptr, B))
                eng.record(ev[1])
                eng.sync()
                if r >= 2:  # round 0 plain warm-up, round 1 captures in graph mode
                    j[mode].append(eng.elapsed_ms(ev[0], ev[1]))
    for j in jobs:  # key switch alone, to subtract
        eng, ev = j["eng"], j["ev"]
        eng._chk(eng.lib.tfhe_amd_bootstrap_woks(eng.ctx, j["u"].ptr, 1 << 29, j["x"].ptr, B))
        for r in range(4):
            eng.record(ev[0])
            eng._chk(eng.lib.tfhe_amd_keyswitch(eng.ctx, j["o"].ptr, j["u"].ptr, B))
            eng.record(ev[1])
            eng.sync()
            j["ks"].append(eng.elapsed_ms(ev[0], ev[1]))
    ref = None
    for j in jobs:
        eng = j["eng"]
        eng.set_option(T.OPT_STREAMED_GRAPH, 0)
        eng._chk(eng.lib.tfhe_amd_bootstrap_streamed(eng.ctx, j["o"].ptr, 1 << 29, j["x"].ptr, B))
        out = j["o"].download(np.int32, (B, cfg.n + 1))
        same = "" if ref is None else "  outputs==first: %s" % bool(np.array_equal(out, ref))
        if ref is None:
            ref = out
        ks = statistics.median(j["ks"])
        for mode in ("plain", "graph"):
            ms = statistics.median(j[mode])
            per = (ms - ks) * 1e-3 / cfg.n
            print("%-34s %-5s median %.3f min %.3f ms  %.2f us / launch  hbm_frac %.3f  -> %.0f bootstraps/s%s" % (
                os.path.basename(j["lib"]), mode, ms, min(j[mode]), per * 1e6, (B * 16388 + 65536) / per / 8e12, B / ms * 1e3, same),
                flush=True)
    for j in jobs:
        j["job"].close()


if __name__ == "__main__":
    main()
