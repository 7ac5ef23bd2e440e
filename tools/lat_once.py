#!/usr/bin/env python3
"""One purpose: a few gate bootstraps of a small batch on ONE chosen blind-rotation kernel, for profiling
(run ON THE GPU BOX, directly after `rocprofv3 ... --`):

    python3 tools/lat_once.py --batch 1 --split 1 --reps 3

No child processes, no torch.  Prints the HIP-event time per call."""
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
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--split", type=int, default=1, help="1: k_blind_rotate_split, 0: k_blind_rotate")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig()
    job = shard.GateJob(cfg, 0x5446484500000001, device=0, lib_path=a.lib)
    eng, lib = job.eng, job.eng.lib
    eng.set_option(T.OPT_BR_SPLIT, (1 << 30) if a.split else 0)
    x_d = eng.to_device(shard.synthetic_samples(cfg, a.batch, seed=99))
    u_d = eng.alloc(a.batch * (cfg.N + 1) * 4)
    e0, e1 = eng.event(), eng.event()
    ts = []
    for _ in range(a.reps + 1):
        eng.record(e0)
        eng._chk(lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, 1 << 29, x_d.ptr, a.batch))
        eng.record(e1)
        eng.sync()
        ts.append(eng.elapsed_ms(e0, e1))
    print(json.dumps({"batch": a.batch, "split": a.split, "blind_rotate_ms": ts[1:]}))
    job.close()


if __name__ == "__main__":
    main()
