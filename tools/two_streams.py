#!/usr/bin/env python3
"""Does a second stream buy throughput?  BASELINE config 2 (4096 gate bootstraps per step) with the steps issued
alternately on TWO engine contexts of one device (each its own stream, keys and buffers) against all steps on one:
the key switch of step i (matrix cores, 320 workgroups on 512 slots) and the tail of its blind rotation could run
beside the blind rotation of step i + 1.  Run ON A GPU BOX:

    python3 tools/two_streams.py [--batch 4096] [--steps 20] [--rounds 4]

Prints host wall time per step (call .. device idle) for both schedules, rounds interleaved."""
import argparse
import importlib
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig()
    jobs = []
    for k in range(2):
        job = shard.GateJob(cfg, 0x5446484500000001, device=0)
        eng = job.eng
        x_d = eng.to_device(shard.synthetic_samples(cfg, a.batch, seed=7 + k))
        o_d = eng.alloc(a.batch * (cfg.n + 1) * 4)
        jobs.append((job, eng, x_d, o_d))

    def run(two):
        t0 = time.perf_counter()
        for s in range(a.steps):
            job, eng, x_d, o_d = jobs[s % 2 if two else 0]
            eng._chk(eng.lib.tfhe_amd_bootstrap(eng.ctx, o_d.ptr, 1 << 29, x_d.ptr, a.batch))
        for job, eng, x_d, o_d in jobs:
            eng.sync()
        return (time.perf_counter() - t0) / a.steps * 1e3

    run(False), run(True)  # warm-up
    one, two = [], []
    for r in range(a.rounds):
        one.append(run(False))
        two.append(run(True))
    print("one stream : median %.3f ms per step (min %.3f) -> %.0f bootstraps/s" % (statistics.median(one), min(one), a.batch / statistics.median(one) * 1e3))
    print("two streams: median %.3f ms per step (min %.3f) -> %.0f bootstraps/s" % (statistics.median(two), min(two), a.batch / statistics.median(two) * 1e3))
    for j in jobs:
        j[0].close()


if __name__ == "__main__":
    main()
