#!/usr/bin/env python3
"""What the ring-degree-generic kernels (csrc/tfhe_kernels_generic.h) deliver, beside the tuned N = 1024 path on the same box:

    python tools/generic_rate.py [--batch 4096] [--rounds 5] [--n 64] [--degrees 512,1024,4096]

Per ring degree: gate bootstraps (blind rotation of `--n` CMux steps + extraction; key switch timed apart) and the four
execute_* conversions on `--batch` polynomials, HIP-event times, median of `--rounds`.  Rates are per CMux step so that ring
degrees compare: flop per CMux = 4 x 5 (N/2) log2(N/2) + ... (SURVEY 8d formula, l = 2)."""
import argparse
import importlib
import json
import math
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cmux_flop(N, l):
    half = N // 2
    fft = 5 * half * int(math.log2(half))
    return 2 * l * fft + 2 * (fft + N) + 2 * l * 2 * 8 * half  # 2l inverse, 2 direct (+ scale), 2l x 2 complex MACs of 8 flop


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--degrees", default="512,1024,4096")
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    rows = []
    for N in [int(v) for v in a.degrees.split(",")]:
        batch = a.batch if N <= 4096 else max(64, a.batch * 4096 // N)
        cfg = shard.GateConfig(N=N, n=a.n, ks_t=4, ks_basebit=2)
        job = shard.GateJob(cfg, 0x5446484500000001, device=0, lib_path=a.lib)
        eng = job.eng
        x_d = eng.to_device(shard.synthetic_samples(cfg, batch, seed=7))
        u_d, o_d = eng.alloc(batch * (N + 1) * 4), eng.alloc(batch * (cfg.n + 1) * 4)
        ev = [eng.event() for _ in range(3)]
        br, ks = [], []
        for r in range(a.rounds + 1):
            eng.record(ev[0])
            eng._chk(eng.lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, 1 << 29, x_d.ptr, batch))
            eng.record(ev[1])
            eng._chk(eng.lib.tfhe_amd_keyswitch(eng.ctx, o_d.ptr, u_d.ptr, batch))
            eng.record(ev[2])
            eng.sync()
            if r:
                br.append(eng.elapsed_ms(ev[0], ev[1]))
                ks.append(eng.elapsed_ms(ev[1], ev[2]))
        t_br = statistics.median(br)
        # transforms
        rs = np.random.RandomState(N)
        dig_d = eng.to_device(rs.randint(-512, 512, size=(batch, N)).astype(np.int32))
        lag_d, t32_d = eng.alloc(batch * N * 8), eng.alloc(batch * N * 4)
        tr = {"ifft_int32": [], "fft_torus32": []}
        for r in range(a.rounds + 1):
            eng.record(ev[0])
            eng._chk(eng.lib.tfhe_amd_ifft_int32(eng.ctx, lag_d.ptr, dig_d.ptr, batch))
            eng.record(ev[1])
            eng._chk(eng.lib.tfhe_amd_fft_torus32(eng.ctx, t32_d.ptr, lag_d.ptr, batch))
            eng.record(ev[2])
            eng.sync()
            if r:
                tr["ifft_int32"].append(eng.elapsed_ms(ev[0], ev[1]))
                tr["fft_torus32"].append(eng.elapsed_ms(ev[1], ev[2]))
        row = {"N": N, "kernels": "tuned (wave per polynomial)" if N in (1024, 2048) else "generic (team per polynomial)",
               "batch": batch, "cmux_steps": cfg.n, "blind_rotate_ms": round(t_br, 3),
               "us_per_cmux_per_ciphertext": round(t_br * 1e3 / cfg.n / batch, 4),
               "fp64_tflops": round(cmux_flop(N, cfg.l) * cfg.n * batch / (t_br * 1e-3) / 1e12, 2),
               "keyswitch_ms": round(statistics.median(ks), 3),
               "ifft_int32_TBps": round(batch * N * 12 / (statistics.median(tr["ifft_int32"]) * 1e-3) / 1e12, 3),
               "fft_torus32_TBps": round(batch * N * 12 / (statistics.median(tr["fft_torus32"]) * 1e-3) / 1e12, 3)}
        rows.append(row)
        print(json.dumps(row), flush=True)
        job.close() if hasattr(job, "close") else eng.close()


if __name__ == "__main__":
    main()
