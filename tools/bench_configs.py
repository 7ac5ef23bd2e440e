#!/usr/bin/env python3
"""Timings for the BASELINE configs that are NOT the bench.py line (run on the GPU box):

    python tools/bench_configs.py fft   [--batch 8192] [--reps 5]     BASELINE config 4
    python tools/bench_configs.py cb    [--cb-batch 768] [--reps 3]   BASELINE config 3
    python tools/bench_configs.py lat   [--lat-batches 1,8,64,...]    BASELINE config 1 (latency) and small batches
    python tools/bench_configs.py ring                                gate bootstraps at N = 512 and 4096 (the generic kernels)
    python tools/bench_configs.py all

fft  batched N=2048 transforms through the FFT-plugin entry points: 8,192
     polynomials (256 MiB working set = Infinity Cache size) and 4x that, each of
     execute_reverse_torus64 / execute_reverse_int / execute_direct_torus64 / execute_direct_torus32(N=1024);
     reported as polynomials/s and as algorithmic GB/s (bytes in + bytes out) against 8 TB/s -- these
     kernels ARE HBM-bound (1.6 flop/B), unlike the blind rotation.
cb   circuit bootstrap at the PoC parameters (poc:70-85: n0=500, N1=1024, N2=2048, l2=4, Bgbit2=9,
     l1=2, Bgbit1=8, preKS 6x2, privKS 10x3 = 2.69 GB) on synthetic keys, stage by stage
     (preKeySwitch, preModSwitch, l1 Torus64 blind rotations, 2*l1 private key switches), then the
     whole tfhe_amd_circuit_bootstrap call, then a 16-bit LUT evaluation over its outputs.

One JSON object per line.  No torch, no child processes; HIP events through the engine's C ABI.
Synthetic keys are uniformly random tables: throughput does not depend on key contents.
`--small` shrinks every size (for the CPU emulator build: --lib tests/emu/_build/libtfhe_amd_emu.so)."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12


class Events:
    """HIP events of a raw context handle (same three methods as Engine)"""

    def __init__(self, lib, ctx):
        self.lib, self.ctx = lib, ctx

    def event(self):
        e = C.c_void_p()
        assert self.lib.tfhe_amd_event_create(self.ctx, C.byref(e)) == 0
        return e

    def record(self, e):
        assert self.lib.tfhe_amd_event_record(self.ctx, e) == 0

    def elapsed_ms(self, e0, e1):
        ms = C.c_float()
        assert self.lib.tfhe_amd_event_elapsed_ms(self.ctx, e0, e1, C.byref(ms)) == 0
        return float(ms.value)


def timed(eng, reps, fn):
    """min and mean HIP-event time (ms) of fn() over `reps` runs after one warm-up"""
    e0, e1 = eng.event(), eng.event()
    fn()
    ts = []
    for _ in range(reps):
        eng.record(e0)
        fn()
        eng.record(e1)
        ts.append(eng.elapsed_ms(e0, e1))
    return min(ts), float(np.mean(ts))


def rand_bits(rs, shape, dtype):
    """uniform values of the full integer range (any bit pattern is a valid torus element)"""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer(rs.bytes(nbytes), dtype=dtype).reshape(shape)


def bench_fft(T, a):
    rs = np.random.RandomState(4)
    lines = []
    for N, batch in ((2048, a.batch), (2048, 4 * a.batch), (1024, 4 * a.batch)):
        if a.small:
            batch = 8
        eng = T.Engine(torus_bits=64 if N == 2048 else 32, n=1, N=N, l=2, Bgbit=8, ks_t=0, lib_path=a.lib)
        lib = eng.lib
        i64 = eng.to_device(rand_bits(rs, (batch, N), np.int64))
        i32 = eng.to_device(rs.randint(-256, 256, size=(batch, N)).astype(np.int32))      # digits in [-256, 256)
        lag = eng.alloc(batch * N * 8)
        o64, o32 = eng.alloc(batch * N * 8), eng.alloc(batch * N * 4)
        eng._chk(lib.tfhe_amd_ifft_int32(eng.ctx, lag.ptr, i32.ptr, batch))                # Lagrange doubles to transform back
        cases = [("execute_reverse_int", 4 + 8, lambda: eng._chk(lib.tfhe_amd_ifft_int32(eng.ctx, lag.ptr, i32.ptr, batch))),
                 ("execute_reverse_torus64", 8 + 8, lambda: eng._chk(lib.tfhe_amd_ifft_torus64(eng.ctx, lag.ptr, i64.ptr, batch))),
                 ("execute_direct_torus64", 8 + 8, lambda: eng._chk(lib.tfhe_amd_fft_torus64(eng.ctx, o64.ptr, lag.ptr, batch))),
                 ("execute_direct_torus32", 8 + 4, lambda: eng._chk(lib.tfhe_amd_fft_torus32(eng.ctx, o32.ptr, lag.ptr, batch)))]
        for name, bytes_per_coef, fn in cases:
            if name == "execute_direct_torus64":
                eng._chk(lib.tfhe_amd_ifft_int32(eng.ctx, lag.ptr, i32.ptr, batch))
            best, mean = timed(eng, a.reps, fn)
            algo = batch * N * bytes_per_coef
            lines.append({"workload": f"{name} N={N} batch={batch}", "ms_min": best, "ms_mean": mean,
                          "polynomials_per_s": batch / (best * 1e-3),
                          "roofline": {"bound": "hbm", "achieved": algo / (best * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                                       "unit": "GB/s", "frac": algo / (best * 1e-3) / HBM_PEAK,
                                       "algorithmic_bytes_per_launch": algo,
                                       "working_set_MiB": algo / 2 ** 20}})
            print(json.dumps(lines[-1]), flush=True)
        if N == 2048:  # the Real96 (128-bit fixed point) transforms of the same size: integer-VALU bound
            spec = eng.alloc(batch * (N // 2) * 32)
            for name, bytes_per_poly, fn in (
                    ("Real96 iFFT", N * 8 + N * 16, lambda: eng._chk(lib.tfhe_amd_hp_ifft(eng.ctx, spec.ptr, i64.ptr, batch))),
                    ("Real96 FFT", N * 16 + N * 8, lambda: eng._chk(lib.tfhe_amd_hp_fft(eng.ctx, o64.ptr, spec.ptr, batch)))):
                best, mean = timed(eng, a.reps, fn)
                algo = batch * bytes_per_poly
                lines.append({"workload": f"{name} N={N} batch={batch}", "ms_min": best, "ms_mean": mean,
                              "polynomials_per_s": batch / (best * 1e-3),
                              "hbm": {"achieved_GBps": algo / (best * 1e-3) / 1e9, "frac": algo / (best * 1e-3) / HBM_PEAK}})
                print(json.dumps(lines[-1]), flush=True)
        eng.close()
    return lines


def bench_cb(T, a):
    rs = np.random.RandomState(5)
    if a.small:
        n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, B, d = 3, 1024, 2048, 2, 8, 4, 9, 2, 2, 2, 3, 4, 4
    else:
        n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, B, d = 500, 1024, 2048, 2, 8, 4, 9, 6, 2, 10, 3, a.cb_batch, 16
    B -= B % d
    preks = rand_bits(rs, (N1, t10, 1 << bb10, n0 + 1), np.int32)
    bk = rand_bits(rs, (n0, 2 * l2, 2, N2), np.int64)
    x = rand_bits(rs, (B, N1 + 1), np.int32)
    planes = [rand_bits(rs, ((N2 + 1) * t21 * (1 << bb21) * 2 * N1,), np.int32) for _ in range(2)]  # 1.35 GB each
    # The checker's answers for a few of the TIMED inputs, computed before the engine exists (no GPU use yet in a process that
    # starts here): tfhe_CircuitBootstrapFFT (poc:823-873) restated by the oracle on the same synthetic keys -- first, last and
    # the rows either side of a 256-sample tile boundary of the private key switch.  Compared bit for bit after the timed runs.
    check_rows, want = [r for r in (0, 255, 256, B - 1) if 0 <= r < B], None
    if not getattr(a, "no_oracle", False):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_py as O
        bkfft = O.execute_reverse_torus64(N2, bk.reshape(-1, N2)).reshape(bk.shape)  # tGswToFFTConvert
        table = np.stack([p.reshape(N2 + 1, t21, 1 << bb21, 2, N1) for p in planes])
        want = np.stack([O.circuit_bootstrap(x[r], preks, bkfft, table, n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21) for r in check_rows])
        del bkfft, table
    cb = T.CircuitBootstrap(n0, N1, N2, l1, bg1, l2, bg2, t10, bb10, t21, bb21, lib_path=a.lib)
    lib = cb.lib
    cb.load_preks(preks)
    cb.load_bk_torus(bk)                                                  # tGswToFFTConvert on the GPU
    for u in range(2):                                                    # one 1.35 GB plane at a time
        cb._chk(lib.tfhe_amd_cb_load_privks_plane(cb.cb, u, T._np_ptr(planes[u])))
        planes[u] = None
    del planes
    d_x = cb._dev(x)
    d_out = cb._dev(np.zeros((B, 2, l1, 2, N1), np.int32))

    ev = Events(lib, cb.ctx)  # all of cb shares one stream; events go through the level-2 context
    whole_min, whole_mean = timed(ev, a.reps, lambda: cb._chk(lib.tfhe_amd_circuit_bootstrap(cb.cb, d_out, d_x, B)))
    line = {"workload": f"tfhe_CircuitBootstrapFFT n0={n0} N1={N1} N2={N2} l2={l2} Bgbit2={bg2} l1={l1} Bgbit1={bg1} "
                        f"preKS {t10}x{bb10} privKS {t21}x{bb21}, batch {B}, synthetic keys",
            "ms_min": whole_min, "ms_mean": whole_mean, "circuit_bootstraps_per_s": B / (whole_min * 1e-3)}
    if want is not None:
        got = np.empty((B, 2, l1, 2, N1), np.int32)
        cb._chk(lib.tfhe_amd_memcpy_d2h(cb.ctx, T._np_ptr(got), d_out, got.nbytes))
        line["oracle_bit_identical"] = bool(np.array_equal(got[check_rows], want))
        line["oracle_rows"] = check_rows
        del got
        if not line["oracle_bit_identical"]:
            raise RuntimeError(f"circuit bootstrap: the timed outputs of rows {check_rows} differ from the oracle")
    # stages, timed separately through the same entry points the pipeline composes
    c10, c2 = cb.ctx10, cb.ctx
    d_pre = cb._dev(np.zeros((B, n0 + 1), np.int32))
    d_abar = cb._dev(np.zeros((B, n0 + 1), np.int32))
    d_boot = cb._dev(np.zeros((B, N2 + 1), np.int64))
    d_row = cb._dev(np.zeros((B, 2, N1), np.int32))
    stages = {
        "preKeySwitch": lambda: cb._chk(lib.tfhe_amd_keyswitch(c10, d_pre, d_x, B)),
        "preModSwitch": lambda: cb._chk(lib.tfhe_amd_modswitch(c2, d_abar, d_pre, B)),
        "circuitBootstrapWoKS (one of l1)": lambda: cb._chk(lib.tfhe_amd_cb_bootstrap_woks(c2, d_boot, 1 << 55, d_abar, B)),
        "circuitPrivKS (one plane, batch samples; the pipeline runs 2 launches of l1*batch)": lambda: cb._chk(lib.tfhe_amd_privks(cb.cb, d_row, 0, d_boot, B)),
    }
    line["stages_ms"] = {k: timed(ev, a.reps, f)[0] for k, f in stages.items()}
    # the dominant kernel (87 % of the call): Torus64 / N2 blind rotation, bound by fp64 ISSUE -- 8,064 wave64 fp64
    # instructions per CMux per sample (8 inverse + 2 forward transforms, 16 half-row MACs, conversions; N2 = 2048, l2 = 4)
    # against 1024 SIMDs x 2.4 GHz / 4 cycles; the flop view beside it (647,168 flop per CMux vs 78.6 TF)
    t_br = line["stages_ms"]["circuitBootstrapWoKS (one of l1)"] * 1e-3
    cmux = B * n0 / t_br
    if N2 == 2048 and l2 == 4:
        line["blind_rotation_roofline"] = {"kernel": "k_blind_rotate<int64,N=2048>", "bound": "fp64_issue", "cmux_per_s": cmux,
                                           "achieved": cmux * 8064 / 1e9, "peak": 1024 * 2.4e9 / 4 / 1e9, "unit": "G fp64 wave-instr/s",
                                           "frac": cmux * 8064 * 4 / (1024 * 2.4e9),
                                           "fp64_valu_frac": cmux * 647168 / 78.6e12,
                                           "algorithmic_bytes_per_cmux": 65544, "hbm_contract_frac": cmux * 65544 / HBM_PEAK}
    plane_bytes = (N2 + 1) * t21 * (1 << bb21) * 2 * N1 * 4
    t_priv = line["stages_ms"]["circuitPrivKS (one plane, batch samples; the pipeline runs 2 launches of l1*batch)"] * 1e-3
    line["privks_hbm"] = {"note": "batch-major: one launch streams one table plane once (SURVEY 8a a19: 2.69 GB in two planes)",
                          "algorithmic_bytes_per_launch": plane_bytes, "achieved_GBps": plane_bytes / t_priv / 1e9,
                          "peak_GBps": HBM_PEAK / 1e9, "frac": plane_bytes / t_priv / HBM_PEAK}
    print(json.dumps(line), flush=True)
    # LUT evaluation over the circuit bootstrap's outputs (d bits per item)
    items = B // d
    table = rand_bits(rs, (max(1, (1 << d) // N1), N1), np.int32)
    d_tab, d_lwe = cb._dev(table), cb._dev(np.zeros((items, N1 + 1), np.int32))
    bits = C.c_void_p()
    cb._chk(lib.tfhe_amd_gsw_from_torus_d(c10, d_out, B, C.byref(bits)))
    t_min, t_mean = timed(ev, a.reps, lambda: cb._chk(lib.tfhe_amd_lut_eval(c10, d_lwe, bits, d, d_tab, items)))
    cmux_count = items * ((max(1, (1 << d) // N1) - 1) + min(d, 10))
    l2line = {"workload": f"LUT evaluation by vertical packing, d={d} bits, {items} items (TGSW32 from the circuit bootstrap)",
              "ms_min": t_min, "ms_mean": t_mean, "lut_evaluations_per_s": items / (t_min * 1e-3),
              "cmux_per_s": cmux_count / (t_min * 1e-3)}
    print(json.dumps(l2line), flush=True)
    lib.tfhe_amd_gsw_free(bits)
    cb.close()
    return [line, l2line]


def bench_ring(T, a):
    """gate bootstraps at ring degrees the reference's plugin accepts but never instantiates (csrc/tfhe_kernels_generic.h): N = 512 and
    4096 with the headline set's other parameters (n = 630, l = 2, Bgbit = 10, key switch 8 x 2), synthetic keys, batch 4096 (1024 at
    4096); a few of the TIMED outputs are bit-compared with the oracle (computed before the engine exists)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    lines = []
    for N, B in ((512, 4096), (4096, 1024)):
        n, l, bg, t, bb = (3, 2, 10, 2, 2) if a.small else (630, 2, 10, 8, 2)
        if a.small:
            B = 5
        rs = np.random.RandomState(N)
        bk = rand_bits(rs, (n, 2 * l, 2, N), np.int32)
        ks = rand_bits(rs, (N, t, 1 << bb, n + 1), np.int32)
        x = rand_bits(rs, (B, n + 1), np.int32)
        rows = [0, B // 2, B - 1]
        bkfft = O.execute_reverse_int(N, bk.reshape(-1, N)).reshape(bk.shape)  # tGswToFFTConvert (execute_reverse_torus32)
        want = np.stack([O.bootstrap32(N, bkfft, ks, 1 << 29, x[r], l, bg, t, bb) for r in rows])
        del bkfft
        eng = T.Engine(torus_bits=32, n=n, N=N, l=l, Bgbit=bg, ks_t=t, ks_basebit=bb, lib_path=a.lib)
        try:
            eng.set_bootstrap_key(eng.gsw_from_torus(bk))
            eng.load_keyswitch_key(ks)
            x_d, u_d, o_d = eng.to_device(x), eng.alloc(B * (N + 1) * 4), eng.alloc(B * (n + 1) * 4)
            br_min, _ = timed(eng, a.reps, lambda: eng._chk(eng.lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, 1 << 29, x_d.ptr, B)))
            ks_min, _ = timed(eng, a.reps, lambda: eng._chk(eng.lib.tfhe_amd_keyswitch(eng.ctx, o_d.ptr, u_d.ptr, B)))
            got = o_d.download(np.int32, (B, n + 1))
            half = N // 2
            fft = 5 * half * int(np.log2(half))
            flop = 2 * l * fft + 2 * (fft + N) + 2 * l * 2 * 8 * half  # SURVEY 8d's formula at this N
            lines.append({"N": N, "batch": B, "workload": f"tfhe_bootstrap_FFT n={n} N={N} l={l} Bgbit={bg} ks {t}x{bb}, synthetic keys",
                          "blind_rotate_ms": br_min, "keyswitch_ms": ks_min, "bootstraps_per_s": B / ((br_min + ks_min) * 1e-3),
                          "fp64_tflops": flop * n * B / (br_min * 1e-3) / 1e12, "oracle_bit_identical": bool(np.array_equal(got[rows], want))})
            print(json.dumps(lines[-1]), flush=True)
        finally:
            eng.close()
    return lines


def bench_latency(T, a):
    """BASELINE config 1: a single gate bootstrap (and small batches) -- the reference's own unit of work is one
    sample per call (lwe_functions.cpp:434-446).  Each batch size is timed on both blind-rotation kernels: the
    latency-shaped one (one ciphertext per 4-wave workgroup, k_blind_rotate_split) and one wave per ciphertext
    (k_blind_rotate); `auto` is what the library picks by itself.  HIP-event time of tfhe_amd_bootstrap (blind
    rotation + extraction + key switch) and host wall time of call + sync."""
    import time
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig() if not a.small else shard.GateConfig(n=6)
    # the reference's own latency beside it: ONE process of oracle/_ref/ref_driver bench32 (tfhe_bootstrap_FFT composed
    # from the reference's FFT / AddMul object code) on one host core -- a child process, so it runs BEFORE the engine
    # touches the GPU (bench_latency is the first GPU user of this program when run as `lat`; `all` runs it first too)
    cpu_ref = None
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not a.small and a.lib is None and os.access(ref, os.X_OK) and not getattr(a, "_gpu_touched", False):
        import subprocess
        out = subprocess.run([ref, "bench32", "/dev/null", "/dev/null", str(cfg.n), str(cfg.l), str(cfg.Bgbit), str(cfg.ks_t),
                              str(cfg.ks_basebit), "4"], capture_output=True, text=True).stdout.split()
        if len(out) >= 2 and float(out[0]) > 0:
            cpu_ref = {"ms_per_bootstrap": 1e3 * float(out[1]) / float(out[0]), "cores": 1, "kind": "reference",
                       "sample": f"{out[0]} bootstraps in {float(out[1]):.1f} s, one process of ref_driver bench32"}
    job = shard.GateJob(cfg, 0x5446484500000001, device=0, lib_path=a.lib)
    eng, lib = job.eng, job.eng.lib
    sizes = [1, 2, 3] if a.small else [int(x) for x in a.lat_batches.split(",")]
    bmax = max(sizes)
    x = shard.synthetic_samples(cfg, bmax, seed=99)
    nreal = min(8, bmax)
    msgs = [(1 << 29) if (i & 1) else -(1 << 29) for i in range(nreal)]
    x[:nreal] = job.encrypt(msgs)
    x_d, out_d = eng.to_device(x), eng.alloc(bmax * (cfg.n + 1) * 4)
    eng._chk(lib.tfhe_amd_bootstrap(eng.ctx, out_d.ptr, 1 << 29, x_d.ptr, bmax))  # workspaces sized once
    ref_out = {}
    for B in sizes:
        line = {"workload": f"tfhe_bootstrap_FFT, batch {B}, {cfg.describe()}", "batch": B}
        for name, opt in (("split", 1 << 30), ("one_wave_per_ciphertext", 0), ("auto", -1)):
            eng.set_option(T.OPT_BR_SPLIT, opt)
            fn = lambda: eng._chk(lib.tfhe_amd_bootstrap(eng.ctx, out_d.ptr, 1 << 29, x_d.ptr, B))
            best, mean = timed(eng, a.reps, fn)
            eng.sync()
            walls = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                fn()
                eng.sync()
                walls.append(time.perf_counter() - t0)
            out = out_d.download(np.int32, (bmax, cfg.n + 1))[:B]
            same = bool(np.array_equal(out, ref_out.setdefault(B, out)))
            ok = all((job.phase(out[i]) > 0) == (msgs[i] > 0) for i in range(min(nreal, B)))
            line[name] = {"ms_min": best, "ms_mean": mean, "host_wall_ms_min": 1e3 * min(walls),
                          "bootstraps_per_s": B / (best * 1e-3), "decrypt_check": bool(ok), "identical_to_split": same}
        line["speedup_split_over_one_wave"] = line["one_wave_per_ciphertext"]["ms_min"] / line["split"]["ms_min"]
        if cpu_ref is not None and B == 1:
            line["cpu_baseline"] = cpu_ref
        print(json.dumps(line), flush=True)
    eng.set_option(T.OPT_BR_SPLIT, -1)
    job.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="+", choices=["fft", "cb", "lat", "ring", "all"])
    ap.add_argument("--json-out", default=None,
                    help="also write {'cb': [...], 'fft': [...]} (the lines of the sections that ran, or {'error': ...} for one that "
                         "failed) to this file: how bench.py collects configs 3 and 4 from a child process")
    ap.add_argument("--no-oracle", action="store_true", help="cb: skip the oracle check of the timed outputs")
    ap.add_argument("--lat-batches", default="1,8,64,256,512,768,1024,2048,4096")
    ap.add_argument("--batch", type=int, default=8192, help="polynomials per launch (fft)")
    ap.add_argument("--cb-batch", type=int, default=1024,
                    help="LWE inputs per circuit-bootstrap launch (default 1024 = one Torus64 N=2048 ciphertext per wave, "
                         "4 waves per CU, 256 CUs: the smallest batch that fills the chip)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    T = importlib.import_module("experimental-tfhe_amd")
    if a.lib is None and not os.path.exists(T.DEFAULT_LIB):
        importlib.import_module("experimental-tfhe_amd.build").build()  # child process before any GPU use
    want = set(a.what)
    if "all" in want:
        want = {"lat", "fft", "cb"}
    out, failed = {}, False
    if "lat" in want:   # first: it times the reference in a child process before the GPU is in use
        bench_latency(T, a)
    # cb before fft: its oracle leg wants a process that has not touched the GPU yet
    # (ring before anything else has used the GPU for the same reason)
    for name, fn in (("ring", bench_ring), ("cb", bench_cb), ("fft", bench_fft)):
        if name not in want:
            continue
        try:
            out[name] = fn(T, a)
        except Exception as e:  # noqa: BLE001 -- one section must not cost the other
            out[name] = {"error": repr(e)}
            failed = True
            sys.stderr.write(f"bench_configs {name}: {e!r}\n")
        if a.json_out:  # after every section: a crash in the next one keeps this one
            with open(a.json_out, "w") as f:
                json.dump(out, f)
    if failed:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
