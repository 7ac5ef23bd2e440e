#!/usr/bin/env python3
"""Shader clock, per-wave lifetime and workgroup schedule of the blind-rotation kernel, from a PROBE build of the
engine (-DTFHE_PROBE: s_memtime / s_memrealtime around the CMux loop of every wave, start / end / HW_ID of every
workgroup; the shipped library has none of this):

    python -c "import importlib; B = importlib.import_module('experimental-tfhe_amd.build'); \
               B.build(out='build/ab/lib_probe.so', defines=['TFHE_PROBE'])"
    python tools/wave_probe.py build/ab/lib_probe.so [--batch 4096]
    python tools/wave_probe.py experimental-tfhe_amd/libtfhe_amd_probe.so --warm-seconds 2 --json-out clock.json --quiet

(build.py also ships that probe build as experimental-tfhe_amd/libtfhe_amd_probe.so.)  The second form is what bench.py runs
in a child process: back-to-back launches for --warm-seconds (the chip settles its clock under the load), then ONE stamped
launch: shader clock = sum of d s_memtime / sum of d s_memrealtime x 100 MHz over all waves' CMux loops
(MI355X_MICROARCH.md, "DVFS give-back" item 6).  profiles/r02_wave_balance.txt is this tool's output before and after WaveLds::balance."""
import argparse
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--warm-seconds", type=float, default=0.0, help="back-to-back launches for this long before the stamped ones")
    ap.add_argument("--json-out", default=None, help="write {'shader_clock_ghz': ..., 'kernel_ms': ..., ...} of the last stamped launch here")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    if a.quiet:
        sys.stdout = open(os.devnull, "w")
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig()
    job = shard.GateJob(cfg, 0x5446484500000001, device=0, lib_path=os.path.abspath(a.lib))
    eng = job.eng
    x_d = eng.to_device(shard.synthetic_samples(cfg, a.batch, seed=7))
    u_d = eng.alloc(a.batch * (cfg.N + 1) * 4)
    ev = [eng.event(), eng.event()]
    out = (ctypes.c_ulonglong * 32)()
    eng.lib.tfhe_amd_dbg_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    import json
    import time
    warm_launches, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < a.warm_seconds:
        for _ in range(8):
            eng._chk(eng.lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, 1 << 29, x_d.ptr, a.batch))
        eng.sync()
        warm_launches += 8
    if warm_launches:
        assert eng.lib.tfhe_amd_dbg_read(out) == 0  # reset the totals the warm launches accumulated
    record = None
    for r in range(3 if not a.json_out else 1):
        eng.record(ev[0])
        eng._chk(eng.lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, 1 << 29, x_d.ptr, a.batch))
        eng.record(ev[1])
        eng.sync()
        assert eng.lib.tfhe_amd_dbg_read(out) == 0
        ms = eng.elapsed_ms(ev[0], ev[1])
        clk, real, waves = out[0], out[1], out[2]
        ghz = clk / real * 0.1  # s_memrealtime ticks at 100 MHz
        per_cmux = clk / waves / cfg.n
        record = {"shader_clock_ghz": ghz, "kernel_ms": ms, "waves": int(waves), "cycles_per_cmux_per_wave": per_cmux, "batch": a.batch,
                  "warm_seconds": a.warm_seconds, "warm_launches": warm_launches,
                  "how": "probe build (-DTFHE_PROBE): s_memtime / s_memrealtime stamped around the CMux loop of every wave of ONE "
                         "k_blind_rotate launch, after the warm launches; clock = sum of cycles / sum of 100 MHz ticks"}
        print("run %d: %.3f ms  waves %d  shader clock %.3f GHz  wave lifetime %.3f ms  cycles per CMux per wave %.0f" % (
            r, ms, waves, ghz, real / waves / 1e5, per_cmux))
        nw = max(1, waves // 8)
        print("    in-loop lifetime by wave index (ms):", " ".join("%.3f" % (out[16 + k] / nw / 1e5) for k in range(8)))
    if a.json_out:
        with open(a.json_out, "w") as f:
            json.dump(record, f)
    # workgroup schedule of the last launch: residency per CU and the gaps
    wg = (ctypes.c_ulonglong * 4096)()
    eng.lib.tfhe_amd_dbg_read_wg.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    assert eng.lib.tfhe_amd_dbg_read_wg(wg) == 0
    nwg = (a.batch + 7) // 8
    rows = [(wg[4 * i], wg[4 * i + 1], wg[4 * i + 2], wg[4 * i + 3]) for i in range(min(nwg, 1024))]
    t0 = min(r[0] for r in rows)
    t1 = max(r[1] for r in rows)
    print("workgroups %d  first start -> last end %.3f ms" % (len(rows), (t1 - t0) / 1e5))
    import collections
    percu = collections.defaultdict(list)
    for i, (s0, e0, hw, xcc) in enumerate(rows):
        cu = (xcc & 0xF, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF)  # xcc, se, sh, cu
        percu[cu].append((s0 - t0, e0 - t0, i))
    print("distinct CUs used: %d; workgroups per CU: %s" % (len(percu), dict(collections.Counter(len(v) for v in percu.values()))))
    durs = sorted((e0 - s0) / 1e5 for s0, e0, _, _ in rows)
    print("workgroup duration ms: min %.3f median %.3f max %.3f" % (durs[0], durs[len(durs) // 2], durs[-1]))
    starts = sorted((s0 - t0) / 1e5 for s0, _, _, _ in rows)
    print("start times ms (deciles): " + " ".join("%.2f" % starts[min(len(starts) - 1, k * len(starts) // 10)] for k in range(11)))
    for cu in sorted(percu)[:6]:
        print("   CU", cu, ["%.2f-%.2f (wg %d)" % (s0 / 1e5, e0 / 1e5, i) for s0, e0, i in sorted(percu[cu])])
    job.close()


if __name__ == "__main__":
    main()
