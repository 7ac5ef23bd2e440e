#!/usr/bin/env python3
"""Where does the blind-rotation kernel's time go?  (diagnostic, run on the GPU box)

    python tools/ablate.py [--batch 4096] [--reps 3]

Builds experimental-tfhe_amd/libtfhe_amd_ablate.so (hipcc -DTFHE_ABLATE: the same kernels with
hooks that switch one cost component off at a time) BEFORE touching the GPU, then times
tfhe_amd_bootstrap_woks for every schedule variant with each component removed:
    bk      no bootstrapping-key loads (constants instead)      -> what L2/TA traffic costs
    xch     no LDS transposes                                   -> what the transposes cost
    tw      no LDS twiddle reads (variant 1 has none anyway)    -> what the twiddle reads cost
    rot     no rotated accumulator reads                        -> what the rotation costs
Results with a component removed are WRONG by design; only the times are meaningful.  The shipped
library contains none of these hooks."""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    lib_path = importlib.import_module("experimental-tfhe_amd.build").build(ablate=True)  # child process first
    T = importlib.import_module("experimental-tfhe_amd")
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig()
    job = shard.GateJob(cfg, 0x5446484500000001, device=0, lib_path=lib_path)
    eng, lib = job.eng, job.eng.lib
    lib.tfhe_amd_debug_set_ablation.argtypes = [C.c_uint]
    B = a.batch
    x_d = eng.to_device(shard.synthetic_samples(cfg, B, seed=7))
    u_d = eng.alloc(B * (cfg.N + 1) * 4)
    e0, e1 = eng.event(), eng.event()
    masks = [("full", 0), ("-bk", 1), ("-xch", 2), ("-tw", 4), ("-rot", 8), ("-bk-xch-tw-rot", 15)]
    print(f"batch {B}, {cfg.describe()}; ms per blind-rotation launch (min of {a.reps})")
    print("variant " + " ".join(f"{n:>16s}" for n, _ in masks))
    for variant in (0, 1, 2):
        eng.set_option(T.OPT_BR_VARIANT, variant)
        row = []
        for _, mask in masks:
            assert lib.tfhe_amd_debug_set_ablation(mask) == 0
            best = 1e30
            for _ in range(a.reps + 1):  # first repetition warms up
                eng.record(e0)
                eng._chk(lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, 1 << 29, x_d.ptr, B))
                eng.record(e1)
                best = min(best, eng.elapsed_ms(e0, e1))
            row.append(best)
        print(f"{variant:7d} " + " ".join(f"{v:16.3f}" for v in row), flush=True)
    lib.tfhe_amd_debug_set_ablation(0)
    job.close()


if __name__ == "__main__":
    main()
