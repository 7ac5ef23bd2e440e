#!/usr/bin/env python3
"""Host-array rate of tfhe_amd_pool_bootstrap_host (PCIe included) against the chunk size of the pipelined form, beside the
device-resident rate of the same engine (run on the GPU box):

    python tools/pool_rate.py [--samples 16384] [--chunks 0,1024,2048,4096,8192] [--members 1]

One JSON object per line.  No torch, no child processes."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=16384)
    ap.add_argument("--chunks", default="0,1024,2048,4096,8192")
    ap.add_argument("--members", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig()
    job = shard.GateJob(cfg, 0x5446484500000001, device=0, lib_path=a.lib, keep_host_keys=True)
    eng, lib = job.eng, job.eng.lib
    x = shard.synthetic_samples(cfg, a.samples, seed=3)
    mu = 1 << 29
    # device-resident reference: inputs already in HBM, 4096 per launch
    x_d, u_d, o_d = eng.to_device(x[:4096]), eng.alloc(4096 * (cfg.N + 1) * 4), eng.alloc(4096 * (cfg.n + 1) * 4)
    for _ in range(3):
        eng._chk(lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, mu, x_d.ptr, 4096))
        eng._chk(lib.tfhe_amd_keyswitch(eng.ctx, o_d.ptr, u_d.ptr, 4096))
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        eng._chk(lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, mu, x_d.ptr, 4096))
        eng._chk(lib.tfhe_amd_keyswitch(eng.ctx, o_d.ptr, u_d.ptr, 4096))
    eng.sync()
    resident = 8 * 4096 / (time.perf_counter() - t0)
    print(json.dumps({"device_resident_bootstraps_per_s": resident}), flush=True)
    want = eng.bootstrap(mu, x[:64])
    pool = T.Pool([0] * a.members, torus_bits=32, n=cfg.n, N=cfg.N, l=cfg.l, Bgbit=cfg.Bgbit, ks_t=cfg.ks_t, ks_basebit=cfg.ks_basebit, lib_path=a.lib)
    pool.load_keys_torus(job.bk_host, job.ks_host)
    for chunk in [int(v) for v in a.chunks.split(",")]:
        pool.set_chunk_rows(chunk)
        got = pool.bootstrap(mu, x)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            got = pool.bootstrap(mu, x)
            ts.append(time.perf_counter() - t0)
        counts, secs = pool.last_split()
        print(json.dumps({"members": a.members, "samples_per_call": a.samples, "chunk_rows": chunk, "ms_per_call_min": 1e3 * min(ts),
                          "bootstraps_per_s": a.samples / min(ts), "over_device_resident": a.samples / min(ts) / resident,
                          "identical": bool(np.array_equal(got[:64], want)), "member_ms": [round(1e3 * s, 2) for s in secs]}), flush=True)
    pool.close()
    job.close()


if __name__ == "__main__":
    main()
