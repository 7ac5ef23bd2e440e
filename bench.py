#!/usr/bin/env python3
"""bench.py -- gate bootstraps/s on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          (N > 1, no launcher: starts its own N ranks)

With --gpus N > 1 and no WORLD_SIZE in the environment this process is a GPU-FREE PARENT: it takes the CPU
baseline and the oracle's answers itself (before any rank exists: uncontended host cores), starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process, relays rank 0's one JSON
line and exits with the child's code.  A run whose world size differs from --gpus never prints a line: it
exits non-zero (there is no silent fall-through to one GPU).  The line carries `ranks_seen` (a SUM all-reduce
of ones on the device tensor the timing MAX uses) and every rank's (device, batch, seconds).

A "step" = one pass of tfhe_bootstrap_FFT (blind rotation + extraction + key switch) over a
batch of synthetic LWE samples ALREADY RESIDENT IN HBM (n=630, N=1024, k=1, l=2; Bgbit=10, key
switch 8x2 bits).  Workload by flags:
  --gpus 1            BASELINE config 2: 4096 samples per step
  --gpus N (N > 1)    BASELINE config 5: 2^20 samples per step cut into N contiguous slices, one
                      launch per rank and step, replicated keys, no data-path collective ("strong")
  --batch B           B samples per GPU per step ("weak");  --total M: M per step over all GPUs
                      (`--gpus 1 --total 1048576` is the N = 1 point of the strong-scaling series config 5 names: the same
                      2^20 samples per step on one GPU)
Every line carries `per_gpu_bootstraps_per_s` (min / max / mean over the ranks' own samples / own seconds) and, with N > 1,
`efficiency_vs`: the rate rank 0 reaches ALONE on its own slice right after the timed region while the other ranks idle behind
a barrier -- the N = 1 reference at the same per-GPU batch.  (No efficiency is computed here: whoever reads the series divides.)
  --dist              the multi-GPU code path (torch first, process group, barrier, max over ranks)
                      with the world size --gpus names -- also 1
value = all ranks' bootstraps / max-over-ranks time.

Beside the timed region of a one-GPU run (never inside `value`; --headline-only skips them, which is what the
rocprofv3 scripts under tools/ pass so that the default command's kernel table holds the headline kernels only):
  ring_degrees        gate bootstraps at N = 512 and 4096 (the generic kernels of every ring degree other than 1024 / 2048), same child
  config3_circuit_bootstrap, config4_transforms   BASELINE configs 3 and 4 as tools/bench_configs.py measures them
                      (--other-configs), in a CHILD process that runs and exits BEFORE this process touches the GPU: a fault
                      there costs its own section, never the headline.  Config 3's timed outputs are bit-compared with the oracle.
  config1_latency     BASELINE config 1: one gate bootstrap per call (and 8) through tfhe_amd_bootstrap, HIP events
  streamed_schedule   BASELINE config 2's literal schedule: one external-product launch per CMux step
  sustained           >= --sustained-seconds (10) of back-to-back steps after the timed region: bootstraps/s, min / max step
                      time; with it the shader clock the chip holds UNDER the blind-rotation kernel, from the probe build of the
                      same sources (libtfhe_amd_probe.so: s_memtime / s_memrealtime around every wave's CMux loop, one stamped
                      launch after 2 s of back-to-back ones) run in a child process before this one touches the GPU
  pool_check          the same batch through tfhe_amd_pool_bootstrap_host (host arrays in, host arrays out, PCIe included): a
                      pool of one member and of two members sharing this GPU, keys uploaded from HOST arrays
The `ranks` table carries every rank's PCI bus id; `n_devices` = distinct GPUs.  Under RCCL a run whose ranks do not sit
on --gpus distinct GPUs exits non-zero.  --broadcast-keys: only rank 0 builds the keys, the others receive the bytes of the
device layout by torch.distributed broadcast (RCCL) instead of regenerating them from the seed.

The ONE JSON line on stdout is kept compact (~3 KB: a few numbers per section); the full record -- every note, stage and
per-transform line described below -- is written to gpurun_out/bench_detail_<n>gpu.json (--detail PATH), named by the line's
`detail` key.  The record carries
  roofline      dominant kernel (k_blind_rotate).  bound = "fp64_issue": wave64 fp64 instructions
                per second (2,144 per CMux per sample, the floor of the bit-exact DAG) / HIP-event
                kernel time, against 1024 SIMDs x 2.4 GHz / 4 cycles.  `hbm_contract` keeps the
                north-star byte accounting (16,388 B per CMux per sample + 65,536 B key
                row per CMux per launch, against 8 TB/s), `traffic` the PMC-measured HBM bytes
                (the persistent kernel moves ~2 % of the contract bytes), `fp64_valu` the flop view
  cpu_baseline  the same bootstrap composed from the REFERENCE's own FFT/MAC object code
                (oracle/_ref/ref_driver bench32), one process per host core, bounded to ~10 s,
                on rank 0; falls back to the C oracle ("port") where the reference binary is absent.
Only that cpu_baseline leg touches oracle/.

Process hygiene on the GPU pool: every child process (library build, CPU baseline) is spawned
BEFORE this process touches the GPU; the single-GPU path does not import torch at all (device
memory, stream and HIP events come through the engine's own C ABI); with N > 1 torch is imported
first -- before the engine library is loaded -- so that one HIP runtime serves both.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_PER_GPU = 4096
TOTAL_MULTI_GPU = 1 << 20
SEED = 0x5446484500000001
HBM_PEAK = 8.0e12        # B/s, MI355X_MICROARCH.md
FP64_PEAK = 78.6e12      # flop/s vector fp64 (256 CU x 128 flop/clk x 2.4 GHz)
BYTES_PER_CMUX = 16388   # acc read+write 16,384 + rotation 4
BYTES_PER_ROW = 65536    # bootstrapping-key row, once per CMux per launch
FLOP_PER_CMUX = 173056
FP64_INSTR_PER_CMUX = 2144  # v_*_f64 wave-instructions per CMux per sample: floor of the bit-exact radix-2 DAG (docs/experiments.md, measured bound of the gate kernel)
FP64_ISSUE_CYCLES = 4       # cycles per wave64 fp64 instruction on one SIMD
SIMDS, CLOCK_HZ = 256 * 4, 2.4e9


def kernel_sources_sha256():
    """identity of the kernel sources a PMC traffic figure belongs to (tools/make_traffic.py writes the same hash)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("tfhe_kernels.h", "tfhe_kernels_generic.h", "tfhe_amd.hip", "devport.h"):
        with open(os.path.join(ROOT, "experimental-tfhe_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def cpu_baseline(cfg, seconds=10.0):
    """bounded CPU sample on this host: returns the cpu_baseline object"""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    # one single-threaded process per core, ~125 MB each (key tables); capped so a many-core host
    # cannot be pushed into memory pressure by a baseline measurement
    cores = min(os.cpu_count() or 1, 64)
    try:  # never more than a quarter of the free memory, whatever the core count
        avail_kb = next(int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable"))
        cores = max(1, min(cores, int(0.25 * avail_kb * 1024 // (130 << 20))))
    except (OSError, StopIteration, ValueError):
        cores = min(cores, 8)
    if os.path.exists(ref) and os.access(ref, os.X_OK) and cfg.N == 1024:
        args = [ref, "bench32", "/dev/null", "/dev/null", str(cfg.n), str(cfg.l), str(cfg.Bgbit), str(cfg.ks_t),
                str(cfg.ks_basebit), str(int(seconds))]
        procs = [subprocess.Popen(args, stdout=subprocess.PIPE, text=True) for _ in range(cores)]
        total = 0.0
        for p in procs:
            out = p.communicate()[0].split()
            total += float(out[0]) / float(out[1])
        return {"value": total, "unit": "bootstraps/s", "cores": cores, "kind": "reference", "per_core_value": total / cores,
                "sample": f"{cores} processes x {int(seconds)} s of tfhe_bootstrap_FFT composed from the reference's "
                          "spqlios FMA assembly (execute_reverse_int x4, AddMul x8, execute_direct_torus32 x2 per "
                          "CMux) + key switch, same parameters, synthetic keys/samples"}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    lk, tk = O.keygen_binary(cfg.n, SEED, 1), O.keygen_binary(cfg.N, SEED, 2)
    bk = O.bk_create32(cfg.N, lk, tk, cfg.l, cfg.Bgbit, cfg.bk_stdev, SEED, 1000)
    ks = O.ks_create32(tk, lk, cfg.ks_t, cfg.ks_basebit, cfg.ks_stdev, SEED, 100000)
    rs = np.random.RandomState(1)
    t0, cnt = time.time(), 0
    while time.time() - t0 < seconds:
        x = rs.randint(-2 ** 31, 2 ** 31, size=cfg.n + 1).astype(np.int32)
        O.bootstrap32(cfg.N, bk, ks, 1 << 29, x, cfg.l, cfg.Bgbit, cfg.ks_t, cfg.ks_basebit)
        cnt += 1
    return {"value": cnt / (time.time() - t0), "unit": "bootstraps/s", "cores": 1, "kind": "port",
            "sample": f"{cnt} bootstraps, scalar C oracle, 1 thread"}


def oracle_answers(cfg, rows):
    """the checker's outputs for a few of the timed inputs (CPU only: oracle/liboracle.so through tests/oracle_py.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py as O
    lk, tk = O.keygen_binary(cfg.n, SEED, 1), O.keygen_binary(cfg.N, SEED, 2)
    obk = O.bk_create32(cfg.N, lk, tk, cfg.l, cfg.Bgbit, cfg.bk_stdev, SEED, 1000)
    oks = O.ks_create32(tk, lk, cfg.ks_t, cfg.ks_basebit, cfg.ks_stdev, SEED, 100000)
    return np.stack([O.bootstrap32(cfg.N, obk, oks, 1 << 29, x, cfg.l, cfg.Bgbit, cfg.ks_t, cfg.ks_basebit) for x in rows])


def workload(a, shard, rank, world):
    """(B on this rank, scaling label, samples per step over all ranks, which BASELINE config this is)"""
    if a.batch is not None:
        return (a.batch, "weak", a.batch * world,
                "config 2 shape (fixed batch per GPU)" if a.batch == BATCH_PER_GPU else "custom batch per GPU")
    if a.total is not None or world > 1:
        total = a.total if a.total is not None else TOTAL_MULTI_GPU
        lo, hi = shard.shard_range(total, rank, world)
        return (hi - lo, "strong", total,
                "BASELINE config 5 (2^20 gate bootstraps per step sharded over the GPUs)" if total == TOTAL_MULTI_GPU
                else "custom total per step, sharded")
    # one GPU, no flags: the config the metric is quoted on.  With a single GPU "weak" and "strong" coincide;
    # the label says which rule --gpus N > 1 WITHOUT flags does not share with it (that is config 5, strong)
    return BATCH_PER_GPU, "weak", BATCH_PER_GPU, "BASELINE config 2 (batch 4096 gate bootstraps, one GPU)"


NCHK = 16  # real encryptions at the front of every rank's batch (decrypt-checked); the 8 rows behind them go to the oracle


def oracle_rows(B):
    n = min(NCHK, B)
    return list(range(n, min(n + 8, B)))


def other_configs_child(a):
    """config3_circuit_bootstrap / config4_transforms of the bench line: tools/bench_configs.py's own measurements (HIP events
    through the C ABI, synthetic keys; config 3's timed outputs bit-compared with the oracle), run as a CHILD process that exits
    before this process touches the GPU -- a GPU fault, a hang (timeout) or an exception there costs that section only."""
    import tempfile
    want = [w.strip() for w in a.other_configs.split(",") if w.strip() in ("3", "4", "ring")]
    if not want:
        return {}
    fd, path = tempfile.mkstemp(prefix="tfhe_bench_configs_", suffix=".json")
    os.close(fd)
    small = a.lwe_n is not None  # the test hook of the CPU emulator runs: tiny sizes
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_configs.py")] + [{"3": "cb", "4": "fft", "ring": "ring"}[w] for w in want] + [
        "--json-out", path, "--reps", str(min(3, max(1, a.extras_reps))), "--cb-batch", "1024", "--batch", "8192"]
    if small:
        cmd.append("--small")
    if a.lib:
        cmd += ["--lib", a.lib]
    out, note = {}, None
    try:
        res = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=a.other_configs_timeout)
        if res.returncode:
            note = f"child exit {res.returncode}: {res.stderr[-300:]}"
    except subprocess.TimeoutExpired:
        note = f"child killed after {a.other_configs_timeout} s"
    try:
        with open(path) as f:
            got = json.load(f)
    except (OSError, ValueError):
        got = {}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    if "3" in want:
        sec = got.get("cb")
        if isinstance(sec, list) and len(sec) == 2:
            out["config3_circuit_bootstrap"] = dict(sec[0], baseline_config="BASELINE config 3 (circuit bootstrap TLWE -> TRGSW at "
                                                    "the PoC parameters, batch 1024)", lut_evaluation=sec[1])
        else:
            out["config3_circuit_bootstrap"] = {"error": (sec or {}).get("error") if isinstance(sec, dict) else (note or "no result")}
    if "ring" in want:
        sec = got.get("ring")
        out["ring_degrees"] = ({"what": "gate bootstraps at ring degrees other than the reference's 1024 / 2048 (generic kernels), headline "
                                        "gadget and key switch, timed outputs bit-compared with the oracle", "lines": sec}
                               if isinstance(sec, list) else {"error": (sec or {}).get("error") if isinstance(sec, dict) else (note or "no result")})
    if "4" in want:
        sec = got.get("fft")
        if isinstance(sec, list):
            out["config4_transforms"] = {"baseline_config": "BASELINE config 4 (batched anticyclic transforms, N = 2048 batch 8192 and "
                                         "4 x that; N = 1024 beside it)", "lines": sec}
        else:
            out["config4_transforms"] = {"error": (sec or {}).get("error") if isinstance(sec, dict) else (note or "no result")}
    out["other_configs_run"] = "child process before this process touched the GPU" + (f" ({note})" if note else "")
    return out


def shader_clock_child(a):
    """The shader clock the chip holds UNDER the headline kernel: the probe build of the same sources
    (experimental-tfhe_amd/libtfhe_amd_probe.so, -DTFHE_PROBE: s_memtime / s_memrealtime stamped around every wave's CMux loop)
    run by tools/wave_probe.py in a CHILD process before this process touches the GPU -- 2 s of back-to-back launches, then one
    stamped launch (MI355X_MICROARCH.md, DVFS give-back item 6).  The shipped kernel executes no stamp.  (A co-resident probe
    kernel cannot see it: the blind rotation's two waves per SIMD hold all 512 registers, nothing becomes resident beside them.)"""
    import tempfile
    T = importlib.import_module("experimental-tfhe_amd")
    lib = os.path.join(os.path.dirname(T.DEFAULT_LIB), "libtfhe_amd_probe.so")
    if a.lib is not None or not os.path.exists(lib):
        return None
    fd, path = tempfile.mkstemp(prefix="tfhe_bench_clock_", suffix=".json")
    os.close(fd)
    try:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wave_probe.py"), lib, "--batch", str(BATCH_PER_GPU), "--warm-seconds", "2",
                              "--json-out", path, "--quiet"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=180)
        if res.returncode:
            return {"error": f"child exit {res.returncode}: {res.stderr[-200:]}"}
        with open(path) as f:
            return json.load(f)
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": repr(e)[:200]}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


def write_detail(line, a):
    """the FULL record (every note, every stage, every config-4 line) as a file: gpurun_out/bench_detail_<n>gpu.json next to the
    repository root (gpurun merges that directory back; the driver pulls it), or --detail PATH.  Returns the path or None."""
    path = a.detail or os.path.join(ROOT, "gpurun_out", f"bench_detail_{line['n_gpus']}gpu.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(line, f, indent=1)
        return os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError:
        return None


def per_gpu_rates(ranks, steps):
    rates = [r["batch"] * steps / r["seconds"] for r in ranks if r["seconds"] > 0]
    return {"min": min(rates), "max": max(rates), "mean": sum(rates) / len(rates)} if rates else None


def compact_line(full, detail_path):
    """The ONE JSON line on stdout, kept to ~3.5 KB: whoever stores only a tail of a run's stdout must still hold the whole line.
    Contract keys first, with `roofline` and `cpu_baseline` in the contract's shape; the other BASELINE configs and the
    sections measured after the timed region as a few numbers each, at the END of the line; every note, stage and per-transform
    line is in the detail file.  Each trailing section is built on its own: a missing key there yields {"error": ...} for that
    section and never costs the headline."""
    r = full["roofline"]
    c = full["config"]
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    out["config"] = {"workload": c["workload"].split(", persistent")[0] + "; persistent kernel, inputs resident in HBM",
                     "batch_per_gpu": c["batch_per_gpu"], "total_per_step": c["total_per_step"],
                     "parallelism": f"batch-sharded x{full['n_gpus']}, keys replicated ({c['key_replication']}), no data-path collective",
                     "process_group": c["process_group"], "library": c["library"]}
    if "launched_by" in c:
        out["config"]["launched_by"] = "bench.py GPU-free parent -> torch.distributed.run child"
    out["roofline"] = {"bound": r["bound"], "kernel": r["kernel"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                       "frac": r["frac"], "traffic": r["traffic"], "persistent_algorithmic_bytes": r["persistent_algorithmic_bytes"],
                       "kernel_ms": r["kernel_ms"], "fp64_issue_frac": r["fp64_issue"]["frac"],
                       "hbm_contract_frac": r["hbm_contract"]["frac"], "hbm_contract_bytes": r["hbm_contract"]["algorithmic_bytes_per_launch"],
                       "note": "frac = B*n*173,056 flop / kernel time / 78.6 TF (SURVEY 8d, persistent variant: compute-bound)"}
    if "cpu_baseline" in full:
        b = full["cpu_baseline"]
        out["cpu_baseline"] = {"value": b["value"], "unit": b["unit"], "cores": b["cores"], "kind": b["kind"],
                               "sample": b["sample"].split(" composed from")[0] + (" (reference FMA assembly + key switch)" if b["kind"] == "reference" else ""),
                               "per_core_value": b.get("per_core_value"), "measured_by": (b.get("measured_by") or "").split(":")[0].split(" (")[0]}
    out["ranks_seen"] = full["ranks_seen"]
    out["n_devices"] = full["n_devices"]
    pg = full.get("per_gpu_bootstraps_per_s")
    out["per_gpu_bootstraps_per_s"] = None if not pg else {k: round(v, 1) for k, v in pg.items()}
    ev = full.get("efficiency_vs")
    out["efficiency_vs"] = None if not ev else {"per_gpu_bootstraps_per_s": round(ev["per_gpu_bootstraps_per_s"], 1), "batch": ev["batch"],
                                                "what": "rank 0 alone on its own slice after the timed region, other ranks idle"}
    out["pci"] = full["ranks"][0].get("pci")
    if full["n_gpus"] > 1:
        out["ranks"] = [[x["rank"], x["device"], x["batch"], round(x["seconds"], 4), x.get("pci")] for x in full["ranks"]]  # rank, device, batch, s, PCI bus id
    out["kernels_ms"] = {k: round(v, 4) for k, v in full["kernels_ms"].items()}
    out["checks"] = {"decrypt": full["decrypt_check"],
                     "oracle_bit_identical": None if full["oracle_bit_check"] is None else full["oracle_bit_check"]["identical"],
                     "tail_identical_to_front": None if full["tail_check"] is None else full["tail_check"]["identical_to_front"]}

    def section(name, src, build):
        v = full.get(src)
        if not v:
            return
        try:
            out[name] = {"error": str(v["error"])[:200]} if "error" in v else build(v)
        except (KeyError, TypeError, IndexError, ValueError) as e:
            out[name] = {"error": "compact form: " + repr(e)[:160]}

    def b_latency(lat):
        d = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in lat.items() if k.startswith("latency_batch") and not k.endswith("_wall_ms")}
        d["identical_to_headline"] = all(v for k, v in lat.items() if k.endswith("identical_to_headline_outputs"))
        return d

    def b_streamed(st):
        return {"bootstraps_per_s": round(st["value"], 1), "launches": st["launches"], "extprod_launch_us": round(st["extprod_launch_us"], 3),
                "hbm_frac": round(st["roofline"]["frac"], 4), "identical_to_persistent": st["identical_to_persistent"],
                "hipgraph_bootstraps_per_s": (round(st["hipgraph"]["value"], 1) if "value" in st.get("hipgraph", {}) else None)}

    def b_sustained(su):
        clk = su.get("shader_clock") or {}
        return {"seconds": round(su["seconds"], 2), "steps": su["steps"], "bootstraps_per_s": round(su["value"], 1),
                "step_ms_min": round(su["step_ms_min"], 3), "step_ms_max": round(su["step_ms_max"], 3),
                "over_timed_region": round(su["value"] / full["value"] * full["n_gpus"], 4) if full["value"] else None,
                "shader_clock_ghz": (round(clk["shader_clock_ghz"], 3) if "shader_clock_ghz" in clk else None),
                "probe_build_kernel_ms": (round(clk["kernel_ms"], 3) if "kernel_ms" in clk else None)}

    def b_pool(pc):
        return {"[devices, samples per call, bootstraps/s]": [[m["devices"], m.get("samples_per_call"), round(m["bootstraps_per_s"], 1)] for m in pc["pools"]],
                "pcie_included": True,
                "identical_to_headline": all(m["identical_to_headline"] for m in pc["pools"] if m["identical_to_headline"] is not None),
                "errors": [str(m["error"])[:120] for m in pc["pools"] if "error" in m] or None}

    def b_c3(c3):
        br = c3.get("blind_rotation_roofline", {})
        return {"circuit_bootstraps_per_s": round(c3["circuit_bootstraps_per_s"], 1), "ms_per_batch": round(c3["ms_min"], 3),
                "oracle_bit_identical": c3.get("oracle_bit_identical"),
                "blind_rotation_ms": round(c3["stages_ms"]["circuitBootstrapWoKS (one of l1)"], 3),
                "blind_rotation_flops_frac": (round(br["fp64_valu_frac"], 4) if "fp64_valu_frac" in br else None),
                "lut_evaluations_per_s": round(c3["lut_evaluation"]["lut_evaluations_per_s"], 1)}

    def b_c4(c4):  # HBM fraction (algorithmic bytes / time / 8 TB/s) per conversion, in the order rev_int, rev_t64, dir_t64, dir_t32
        grp = {}
        for ln in c4["lines"]:
            if "roofline" in ln:
                w = ln["workload"].split()
                grp.setdefault(w[1] + " " + w[2], []).append(round(ln["roofline"]["frac"], 3))
            elif "hbm" in ln:  # the Real96 transforms
                w = ln["workload"].split()
                grp.setdefault("Real96 " + w[2] + " " + w[3], []).append(round(ln["hbm"]["frac"], 3))
        return {"hbm_frac [reverse_int, reverse_torus64, direct_torus64, direct_torus32 | Real96: iFFT, FFT]": grp}

    section("sustained", "sustained", b_sustained)
    section("config1_latency", "config1_latency", b_latency)
    section("config2_streamed", "streamed_schedule", b_streamed)
    section("config3_circuit_bootstrap", "config3_circuit_bootstrap", b_c3)
    section("config4_transforms", "config4_transforms", b_c4)

    def b_ring(rg):  # per ring degree: [bootstraps/s, fp64 TF/s of the blind rotation, timed outputs == oracle]
        return {"[bootstraps/s, blind-rotation TF/s, oracle_bit_identical] by N": {str(x["N"]): [round(x["bootstraps_per_s"], 1), round(x["fp64_tflops"], 2),
                                                                                           x["oracle_bit_identical"]] for x in rg["lines"]}}

    section("ring_degrees", "ring_degrees", b_ring)
    section("pool_check", "pool_check", b_pool)
    if "key_broadcast" in full:
        out["key_broadcast"] = full["key_broadcast"]
    if "pipelined_two_contexts" in full:
        out["pipelined_two_contexts"] = {"value": round(full["pipelined_two_contexts"]["value"], 1)}
    out["detail"] = detail_path
    return out


def launch_ranks(a, argv, shard, cfg):
    """--gpus N > 1 without a launcher: this process stays off the GPU (no torch, no engine library), measures the
    CPU baseline and the oracle's answers on an idle host, then runs the N ranks as a child torch.distributed.run and
    relays rank 0's line (pattern: the reference's independent-item loop, parallel/src/test_parallel_multiplications.cpp:62,
    one item range per worker)."""
    import socket
    import tempfile
    hand = {"from": "bench.py parent (GPU-free): measured before any rank was started"}
    if not a.no_cpu_baseline:
        hand["cpu_baseline"] = cpu_baseline(cfg, a.cpu_seconds)
        B0 = workload(a, shard, 0, a.gpus)[0]
        idx = oracle_rows(B0)
        if idx and (a.lib is None or a.lwe_n is not None):
            x0 = shard.synthetic_samples(cfg, B0, seed=1234)  # rank 0's inputs
            hand["oracle_idx"] = idx
            hand["oracle_want"] = oracle_answers(cfg, [x0[i] for i in idx]).tolist()
    fd, path = tempfile.mkstemp(prefix="tfhe_bench_handoff_", suffix=".json")
    with os.fdopen(fd, "w") as f:
        json.dump(hand, f)
    with socket.socket() as sk:  # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TFHE_BENCH_HANDOFF=path, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    try:
        res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    finally:
        os.remove(path)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    for ln in res.stdout.splitlines():
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if res.returncode != 0:
        raise SystemExit(res.returncode)
    if len(lines) != 1:
        raise SystemExit(f"bench.py: the {a.gpus} ranks printed {len(lines)} JSON lines, expected exactly one")
    d = json.loads(lines[0])
    if d.get("n_gpus") != a.gpus or d.get("ranks_seen") != a.gpus:
        raise SystemExit(f"bench.py: asked for {a.gpus} ranks, the line reports n_gpus={d.get('n_gpus')} ranks_seen={d.get('ranks_seen')}")
    if a.backend == "nccl" and d.get("n_devices") != a.gpus:
        raise SystemExit(f"bench.py: {a.gpus} ranks on {d.get('n_devices')} distinct GPUs")
    d["config"]["launched_by"] = "bench.py GPU-free parent -> python -m torch.distributed.run --nproc-per-node %d" % a.gpus
    print(json.dumps(d), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip's clock ramps over a process's first launches (1.9 -> 2.15 -> 2.2 GHz over three 16 ms launches,
    # profiles/r04_wave_probe.txt), so a run without flags warms up three times before it times ten steps (0.2 s in all)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="bootstraps per GPU per step (weak scaling); default: 4096 on one GPU (BASELINE config 2)")
    ap.add_argument("--total", type=int, default=None,
                    help="bootstraps per step over ALL GPUs, sharded contiguously (strong scaling); default with "
                         "--gpus > 1: 2^20 (BASELINE config 5).  `--gpus 1 --total 1048576` is the strong-scaling N = 1 point "
                         "(with N > 1 every line also carries `efficiency_vs`: rank 0 alone on its own slice)")
    ap.add_argument("--dist", action="store_true",
                    help="take the multi-GPU code path (torch imported before the engine, torch.distributed process group, "
                         "barrier + max-over-ranks on a device tensor) even with one rank: the path the driver's "
                         "N = 2/4/8 runs take, runnable on a 1-GPU box")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the multi-rank path (nccl == RCCL on ROCm; gloo: CPU tests on the "
                         "emulator build)")
    ap.add_argument("--self-launch", action="store_true",
                    help="take the GPU-free-parent path (CPU baseline in the parent, ranks as a child torch.distributed.run) "
                         "even with --gpus 1: the N > 1 launch path, runnable on a 1-GPU box")
    ap.add_argument("--no-self-launch", action="store_true",
                    help="--gpus N > 1 without WORLD_SIZE in the environment normally starts the N ranks itself; with this "
                         "flag it exits non-zero instead (it never runs one rank and calls it N)")
    ap.add_argument("--lwe-n", type=int, default=None,
                    help="TEST HOOK (CPU emulator runs of the plumbing): LWE dimension n instead of 630; the line's workload "
                         "string then names it and is not a BASELINE config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--ks-gather", action="store_true", help="per-sample gather key switch instead of the matrix-core one")
    ap.add_argument("--lib", default=None,
                    help="alternative build of the engine library to time (A/B experiments, tools/ab.py); "
                         "default: the shipped libtfhe_amd.so")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip what a one-GPU run measures AFTER the timed region (config 1 latency, config 2's streamed "
                         "schedule): a rocprofv3 --stats run of the command then sees the headline kernels only")
    ap.add_argument("--streamed", action="store_true", help="(kept for older scripts: the streamed schedule is now on by default)")
    ap.add_argument("--latency-batches", default="1,8", help="batch sizes of the config 1 latency section")
    ap.add_argument("--other-configs", default="3,4,ring",
                    help="which of BASELINE's other one-GPU configs a one-GPU run also measures after the timed region "
                         "(tools/bench_configs.py: 3 = circuit bootstrap at the PoC parameters, 4 = batched N = 2048 transforms, "
                         "ring = gate bootstraps at N = 512 and 4096 on the generic kernels); '' = none.  Never part of `value`; --headline-only skips them too")
    ap.add_argument("--extras-reps", type=int, default=5, help="repetitions of each measurement after the timed region")
    ap.add_argument("--detail", default=None,
                    help="where the FULL record (all notes, stages, per-transform lines) is written as a file; default: "
                         "gpurun_out/bench_detail_<n>gpu.json under the repository.  stdout carries the compact line only")
    ap.add_argument("--other-configs-timeout", type=int, default=900, help="seconds the child process measuring configs 3 and 4 may take")
    ap.add_argument("--sustained-seconds", type=float, default=10.0,
                    help="after the timed region: at least this many seconds of back-to-back steps (bootstraps/s, min / max step time, "
                         "shader clock beside one of the steps); 0 = skip.  Never part of `value`; --headline-only skips it too")
    ap.add_argument("--pool-check", action="store_true",
                    help="also under --headline-only / several ranks: the batch through tfhe_amd_pool_bootstrap_host (host arrays, PCIe "
                         "included) on a pool of one member and of two members sharing this rank's GPU; a default one-GPU run does it anyway")
    ap.add_argument("--pool-devices", default=None,
                    help="device ordinals of ONE more pool to check, e.g. 0,1,2,3 or 'all' (every device this process sees): one process "
                         "driving several GPUs")
    ap.add_argument("--broadcast-keys", action="store_true",
                    help="multi-rank path: only rank 0 builds the keys; one torch.distributed broadcast per key (RCCL under nccl) hands the "
                         "other ranks the bytes of the device layout (SURVEY 8e) instead of every rank regenerating them from the seed")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time (after the timed region, one GPU) the same K steps issued alternately on two contexts / "
                         "streams: consecutive batches are independent, the next blind rotation fills the CUs the current "
                         "one's tail and key switch leave idle; reported as pipelined_two_contexts, never as `value`")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    T = importlib.import_module("experimental-tfhe_amd")  # pure python so far: nothing dlopen'ed yet
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    cfg = shard.GateConfig() if a.lwe_n is None else shard.GateConfig(n=a.lwe_n)
    # ---- children first (see module docstring)
    if a.lib is None and not os.path.exists(T.DEFAULT_LIB):
        importlib.import_module("experimental-tfhe_amd.build").build()

    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.self_launch):
        if a.no_self_launch:
            raise SystemExit(f"bench.py: --gpus {a.gpus} needs {a.gpus} ranks and none were launched (WORLD_SIZE is unset, "
                             "--no-self-launch given): refusing to run one rank and report it as several")
        return launch_ranks(a, sys.argv[1:], shard, cfg)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         "--nproc-per-node == --gpus (or without a launcher: bench.py then starts its own ranks)")
    handoff = os.environ.get("TFHE_BENCH_HANDOFF")
    use_dist = world > 1 or a.dist or handoff is not None  # ranks started by bench.py's own parent always take the multi-rank path
    cpu_line = None
    # workload: one GPU = BASELINE config 2 (batch 4096); several = config 5 (2^20 samples sharded
    # contiguously over the ranks, strong scaling) unless --batch asks for a fixed per-GPU batch
    B, scaling, total_per_step, baseline_config = workload(a, shard, rank, world)
    if a.lwe_n is not None:
        baseline_config = f"NOT a BASELINE config (test hook --lwe-n {a.lwe_n}); shape of: " + baseline_config
    x_host = shard.synthetic_samples(cfg, B, seed=1234 + rank)
    nchk = min(NCHK, B)
    oracle_idx, oracle_want, baseline_by = oracle_rows(B), None, None
    if rank == 0 and handoff:
        # started by bench.py's own GPU-free parent: it measured the baseline and the oracle's answers before the ranks existed
        with open(handoff) as f:
            hand = json.load(f)
        cpu_line = hand.get("cpu_baseline")
        if "oracle_want" in hand and hand.get("oracle_idx") == oracle_idx:
            oracle_want = np.array(hand["oracle_want"], dtype=np.int32)
        baseline_by = hand.get("from")
    elif rank == 0 and not a.no_cpu_baseline:
        # rank 0 only, and BEFORE torch / the engine are loaded (children are spawned here); under an external
        # launcher the other ranks meanwhile wait in init_process_group (they generate their keys after the barrier)
        cpu_line = cpu_baseline(cfg, a.cpu_seconds)
        baseline_by = "rank 0, before torch and the engine were loaded (the other ranks wait in init_process_group)"
        # the checker's answers for a few of the timed inputs, computed NOW (before the GPU is touched);
        # compared bit for bit with the GPU's outputs after the timed region
        if oracle_idx and (a.lib is None or a.lwe_n is not None):  # (an A/B build at full size: skipped, the A/B tools compare outputs themselves)
            oracle_want = oracle_answers(cfg, [x_host[i] for i in oracle_idx])

    # BASELINE configs 3 and 4 (the other workloads one GPU can run): a child process, now -- before this process touches the GPU
    other, shader_clock = None, None
    if world == 1 and not a.headline_only and a.other_configs.strip():
        other = other_configs_child(a)
    if rank == 0 and not a.headline_only and a.sustained_seconds > 0:  # (several ranks: the others wait in init_process_group meanwhile)
        shader_clock = shader_clock_child(a)

    dist, dev, torch = None, None, None
    device = 0
    if use_dist:
        import torch  # BEFORE the engine library: one HIP runtime in the process
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            # a launcher may pin one visible device per rank (ROCR_/HIP_VISIBLE_DEVICES): LOCAL_RANK then exceeds the count
            visible = torch.cuda.device_count()
            if visible < 1:
                raise SystemExit("bench.py needs a GPU: torch sees no device")
            device = local % visible
            torch.cuda.set_device(device)
            dev = torch.device("cuda", device)
        else:
            # gloo: the collectives run on CPU tensors, the engine on whichever devices its library shows (the CPU emulator
            # of the tests: TFHE_EMU_DEVICES; a GPU box: its GPUs) -- rank r on device r while they last
            dev = torch.device("cpu")
            device = local % max(1, T.device_count(a.lib))
        # RCCL prints a version banner on STDOUT when its communicator is created (at the first collective); stdout
        # of this program is the one JSON line, so file descriptor 1 points at stderr until that has happened
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if a.backend == "nccl":
                dist.init_process_group(a.backend, rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
            else:
                dist.init_process_group(a.backend, rank=rank, world_size=world)
            dist.barrier()
            if a.backend == "nccl":
                torch.cuda.synchronize()
        finally:
            # the banner is a C printf: it sits in libc's stdio buffer (stdout is a pipe: fully buffered) until that buffer is
            # flushed -- at exit, i.e. AFTER the JSON line and to the restored descriptor -- unless it is flushed here
            sys.stdout.flush()
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    if a.broadcast_keys and dist is None:
        raise SystemExit("bench.py: --broadcast-keys needs the multi-rank path (--gpus N > 1, or --dist for a world of one)")
    want_pool = (world == 1 and not a.headline_only) or a.pool_check or a.pool_devices
    t_keys = time.perf_counter()
    try:
        job = shard.GateJob(cfg, SEED, device=device, lib_path=a.lib, keys="broadcast" if a.broadcast_keys else "seed",
                            tensor_device=dev, keep_host_keys=bool(want_pool))
    except T.TfheAmdError as e:
        raise SystemExit(f"bench.py needs a GPU: the engine has no CPU path ({e})")
    t_keys = time.perf_counter() - t_keys
    eng, lib = job.eng, job.eng.lib
    pci = T.device_pci_bus_id(device, a.lib)
    eng.set_option(T.OPT_KS_GATHER, int(a.ks_gather))

    # a few real encryptions at the front: decrypt-checked after the timed region
    msgs = [(1 << 29) if (i & 1) else -(1 << 29) for i in range(nchk)]
    x_host[:nchk] = job.encrypt(msgs)
    # large batches (config 5: 2^20 / ranks per launch): the last samples repeat the first ones, so the far end of
    # every buffer (64-bit offsets in the kernels) is checked against the oracle-checked front, whatever the size
    ntail = nchk + 8 if B >= 4 * (nchk + 8) else 0
    if ntail:
        x_host[B - ntail:] = x_host[:ntail]
    x_d = eng.to_device(x_host)                     # inputs resident in HBM before timing
    u_d = eng.alloc(B * (cfg.N + 1) * 4)
    out_d = eng.alloc(B * (cfg.n + 1) * 4)
    mu = 1 << 29
    ev = [[eng.event() for _ in range(3)] for _ in range(a.steps)]

    def step(k=None):
        if k is not None:
            eng.record(ev[k][0])
        eng._chk(lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, mu, x_d.ptr, B))
        if k is not None:
            eng.record(ev[k][1])
        eng._chk(lib.tfhe_amd_keyswitch(eng.ctx, out_d.ptr, u_d.ptr, B))
        if k is not None:
            eng.record(ev[k][2])

    def fence():
        eng.sync()
        if dist is not None:
            dist.barrier()
            if a.backend == "nccl":
                torch.cuda.synchronize()
            eng.sync()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    fence()
    elapsed_local = elapsed = time.perf_counter() - t0
    ranks_seen, ranks = 1, [{"rank": 0, "device": device, "batch": B, "seconds": elapsed_local, "pci": pci}]
    if dist is not None:
        elapsed = shard.max_over_ranks(elapsed_local, dev)
        ranks_seen, ranks = shard.rank_census(rank, device, B, elapsed_local, dev, pci)
        if ranks_seen != world:
            raise SystemExit(f"bench.py: {ranks_seen} ranks answered the census, world size is {world}")
    # What ONE GPU does on the very slice it was given, measured behind a barrier while the other ranks idle: the N = 1 rate the
    # aggregate is to be compared with (same per-GPU batch, same kernels, same box) -- a scaling series whose N = 1 point is another
    # workload (config 2: 4096 per step) cannot say whether a shortfall is the slice size or the other GPUs.  Never part of `value`.
    solo = None
    if dist is not None and world > 1:
        t_solo = None
        if rank == 0:
            ts = time.perf_counter()
            for _ in range(a.steps):
                step()
            eng.sync()
            t_solo = time.perf_counter() - ts
        fence()  # the other ranks wait here, GPUs idle
        if rank == 0:
            solo = {"per_gpu_bootstraps_per_s": B * a.steps / t_solo, "batch": B, "steps": a.steps, "seconds": t_solo,
                    "what": "rank 0 alone on its own slice, after the timed region, the other ranks idle behind a barrier: the N = 1 "
                            "rate at THIS per-GPU batch (value / n_gpus against it = what the other GPUs running cost)"}
    n_devices = shard.distinct_devices(ranks)
    if dist is not None and a.backend == "nccl" and n_devices != world:
        # N ranks under RCCL are N GPUs or nothing: two ranks doubled up on one chip would report half the rate as "N GPUs"
        raise SystemExit(f"bench.py: {world} ranks but {n_devices} distinct GPUs (PCI bus ids {[r['pci'] for r in ranks]}): refusing to "
                         f"report n_gpus = {world}")

    # outside the timed region: the real encryptions must decrypt to their sign
    out_all = out_d.download(np.int32, (B, cfg.n + 1))
    out = out_all[:nchk]
    ok = all((job.phase(out[i]) > 0) == (msgs[i] > 0) for i in range(nchk))
    oracle_ok = None if oracle_want is None else bool(np.array_equal(out_all[oracle_idx], oracle_want))
    tail_ok = None if not ntail else bool(np.array_equal(out_all[B - ntail:], out_all[:ntail]))
    br_ms = float(np.mean([eng.elapsed_ms(ev[k][0], ev[k][1]) for k in range(a.steps)]))
    ks_ms = float(np.mean([eng.elapsed_ms(ev[k][1], ev[k][2]) for k in range(a.steps)]))
    extras = world == 1 and not a.headline_only
    extras_ok = True
    reps = max(1, a.extras_reps)

    # Also outside the timed region (one-GPU runs): BASELINE config 1 -- what a caller of the reference's one-sample
    # tfhe_bootstrap_FFT (lwe_functions.cpp:434-446) waits for: blind rotation + extraction + key switch of ONE sample
    # (and of 8), HIP events around tfhe_amd_bootstrap, median of `reps` calls; host wall time of call + sync beside it.
    latency = None
    if extras:
        latency = {"entry_point": "tfhe_amd_bootstrap (blind rotation + extraction + key switch in one call)", "reps": reps,
                   "baseline_config": "BASELINE config 1 (single gate bootstrap, latency)"}
        l0, l1 = eng.event(), eng.event()
        for lb in [int(v) for v in a.latency_batches.split(",") if v.strip()]:
            lb = min(lb, B)
            lo_d = eng.alloc(lb * (cfg.n + 1) * 4)
            eng._chk(lib.tfhe_amd_bootstrap(eng.ctx, lo_d.ptr, mu, x_d.ptr, lb))  # warm-up (kernel selection, workspaces)
            eng.sync()
            ev_ms, wall_ms = [], []
            for _ in range(reps):
                tw0 = time.perf_counter()
                eng.record(l0)
                eng._chk(lib.tfhe_amd_bootstrap(eng.ctx, lo_d.ptr, mu, x_d.ptr, lb))
                eng.record(l1)
                eng.sync()
                wall_ms.append(1e3 * (time.perf_counter() - tw0))
                ev_ms.append(eng.elapsed_ms(l0, l1))
            same = bool(np.array_equal(lo_d.download(np.int32, (lb, cfg.n + 1)), out_all[:lb]))
            latency[f"latency_batch{lb}_ms"] = float(np.median(ev_ms))
            latency[f"latency_batch{lb}_wall_ms"] = float(np.median(wall_ms))
            latency[f"batch{lb}_identical_to_headline_outputs"] = same
            if not same:
                extras_ok = False

    # BASELINE config 2's literal schedule, one external-product launch per CMux with the accumulators
    # round-tripping through HBM (n + 2 launches + key switch).  Same results bit for bit; reported next to the
    # persistent kernel.
    streamed = None
    if extras:
        out2_d = eng.alloc(B * (cfg.n + 1) * 4)
        s0, s1 = eng.event(), eng.event()
        eng._chk(lib.tfhe_amd_bootstrap_streamed(eng.ctx, out2_d.ptr, mu, x_d.ptr, B))  # warm-up
        st = []
        for _ in range(min(reps, 3)):
            eng.record(s0)
            eng._chk(lib.tfhe_amd_bootstrap_streamed(eng.ctx, out2_d.ptr, mu, x_d.ptr, B))
            eng.record(s1)
            st.append(eng.elapsed_ms(s0, s1))
        st_ms = float(np.median(st))
        same = bool(np.array_equal(out2_d.download(np.int32, (B, cfg.n + 1)), out_all))
        per_launch_s = max(st_ms - ks_ms, 1e-9) * 1e-3 / cfg.n
        streamed = {"baseline_config": "BASELINE config 2, literal schedule: one external-product kernel launch per CMux step",
                    "ms_per_step": st_ms, "value": B / (st_ms * 1e-3), "unit": "bootstraps/s",
                    "launches": cfg.n + 3, "identical_to_persistent": same,
                    "extprod_launch_us": per_launch_s * 1e6,
                    "roofline": {"bound": "hbm", "achieved": (B * BYTES_PER_CMUX + BYTES_PER_ROW) / per_launch_s / 1e9,
                                 "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                 "frac": (B * BYTES_PER_CMUX + BYTES_PER_ROW) / per_launch_s / HBM_PEAK,
                                 "algorithmic_bytes_per_launch": B * BYTES_PER_CMUX + BYTES_PER_ROW},
                    "hbm_frac": (B * BYTES_PER_CMUX + BYTES_PER_ROW) / per_launch_s / HBM_PEAK}
        if not same:
            extras_ok = False
        # the same n+3 launches replayed from a hipGraph (call 1 plain, call 2 captures, call 3 is timed);
        # a failure here must not cost the main metric line
        try:
            eng.set_option(T.OPT_STREAMED_GRAPH, 1)
            for _ in range(2):
                eng._chk(lib.tfhe_amd_bootstrap_streamed(eng.ctx, out2_d.ptr, mu, x_d.ptr, B))
            eng.record(s0)
            eng._chk(lib.tfhe_amd_bootstrap_streamed(eng.ctx, out2_d.ptr, mu, x_d.ptr, B))
            eng.record(s1)
            g_ms = eng.elapsed_ms(s0, s1)
            g_launch_s = max(g_ms - ks_ms, 1e-9) * 1e-3 / cfg.n
            streamed["hipgraph"] = {"ms_per_step": g_ms, "value": B / (g_ms * 1e-3), "extprod_launch_us": g_launch_s * 1e6,
                                    "hbm_frac": (B * BYTES_PER_CMUX + BYTES_PER_ROW) / g_launch_s / HBM_PEAK,
                                    "identical_to_persistent": bool(np.array_equal(
                                        out2_d.download(np.int32, (B, cfg.n + 1)), out_all))}
        except T.TfheAmdError as e:
            streamed["hipgraph"] = {"error": str(e)}
        eng.set_option(T.OPT_STREAMED_GRAPH, 0)

    # >= --sustained-seconds of back-to-back steps (the timed region is a fraction of a second on a chip that moves its clock
    # under load): rate, min / max step time by HIP events.  Every rank runs it (all GPUs of a node loaded); rank 0 reports.
    # The shader clock under the kernel comes from the probe build's stamped launch (shader_clock_child, above).
    sustained = None
    if a.sustained_seconds > 0 and not a.headline_only:
        try:
            # steps per host sync: about one second's worth, at most 16 (a step is 17 ms at batch 4096, 4 s at 2^20 per GPU)
            RING = max(1, min(16, int(1.0 / max(elapsed_local / max(a.steps, 1), 1e-3)))) if a.lwe_n is None else 2
            sev = [(eng.event(), eng.event()) for _ in range(RING)]
            step_ms, n_steps = [], 0
            eng.sync()
            ts = time.perf_counter()
            while True:
                for k in range(RING):
                    eng.record(sev[k][0])
                    step()
                    eng.record(sev[k][1])
                eng.sync()
                n_steps += RING
                step_ms += [eng.elapsed_ms(e0, e1) for e0, e1 in sev]
                if time.perf_counter() - ts >= a.sustained_seconds:
                    break
            t_sus = time.perf_counter() - ts
            sustained = {"seconds": t_sus, "steps": n_steps, "value": B * n_steps / t_sus, "unit": "bootstraps/s (this rank)",
                         "step_ms_min": float(min(step_ms)), "step_ms_max": float(max(step_ms)), "step_ms_median": float(np.median(step_ms)),
                         "shader_clock": shader_clock,
                         "note": f"back-to-back steps in rings of {RING} (one host sync per ring); step time = HIP events around blind rotation + key switch"}
        except T.TfheAmdError as e:
            sustained = {"error": str(e)}

    # One process, several members: the same batch through tfhe_amd_pool_bootstrap_host -- host arrays in and out (PCIe included),
    # keys uploaded from HOST arrays (a caller's key, not a seed), one host thread + pinned staging per member.
    pool_check = None
    if want_pool and job.bk_host is not None:
        pool_check = {"entry_point": "tfhe_amd_pool_load_keys_torus + tfhe_amd_pool_bootstrap_host", "pools": []}
        Bp = min(B, BATCH_PER_GPU)  # (a config-5 sized batch would make the 4 x call a 10 GB host array)
        lists = [([device], 1), ([device, device], 1), ([device], 4)]  # (members, how many times the batch per call)
        if a.pool_devices:
            lists.append((list(range(T.device_count(a.lib))) if a.pool_devices == "all" else [int(v) for v in a.pool_devices.split(",")], 4))
        for devs, times in lists:
            try:
                pool = T.Pool(devs, torus_bits=32, n=cfg.n, N=cfg.N, l=cfg.l, Bgbit=cfg.Bgbit, ks_t=cfg.ks_t, ks_basebit=cfg.ks_basebit,
                              lib_path=a.lib)
                tk = time.perf_counter()
                pool.load_keys_torus(job.bk_host, job.ks_host)
                tk = time.perf_counter() - tk
                xp = x_host[:Bp] if times == 1 else np.tile(x_host[:Bp], (times, 1))
                got = pool.bootstrap(mu, xp)  # warm-up: staging buffers, streams, kernel selection
                tp, reps_p = time.perf_counter(), max(1, min(reps, 3))
                for _ in range(reps_p):
                    got = pool.bootstrap(mu, xp)
                tp = (time.perf_counter() - tp) / reps_p
                counts, secs = pool.last_split()
                same = bool(np.array_equal(got, out_all[:Bp] if times == 1 else np.tile(out_all[:Bp], (times, 1))))
                pool_check["pools"].append({"devices": devs, "pci": [T.device_pci_bus_id(d, a.lib) for d in devs], "key_upload_s": tk,
                                            "samples_per_call": Bp * times, "bootstraps_per_s": Bp * times / tp, "ms_per_call": 1e3 * tp,
                                            "split": counts, "member_seconds": secs, "identical_to_headline": same,
                                            "pipelined": "slices of >= 4096 samples go as chunks of 2048: kernels back to back on the member's stream, copies in / out on two more"})
                del xp, got
                pool.close()
                if not same:
                    extras_ok = False
            except T.TfheAmdError as e:
                # a pool that could not be BUILT or RUN (allocation failure, a bad --pool-devices list) costs its own entry, never
                # the run's exit status: only outputs that DIFFER from the headline's do (identical_to_headline False above)
                pool_check["pools"].append({"devices": devs, "error": str(e), "bootstraps_per_s": 0.0, "identical_to_headline": None})

    pipelined = None
    if world == 1 and a.pipelined:
        job2 = shard.GateJob(cfg, SEED, device=device, lib_path=a.lib)  # second context: its own stream and key replicas
        e2 = job2.eng
        e2.set_option(T.OPT_KS_GATHER, int(a.ks_gather))
        x2_d, u2_d, o2_d = e2.to_device(x_host), e2.alloc(B * (cfg.N + 1) * 4), e2.alloc(B * (cfg.n + 1) * 4)
        lanes = [(eng, x_d, u_d, out_d), (e2, x2_d, u2_d, o2_d)]

        def pstep(k):
            e, xd, ud, od = lanes[k & 1]
            e._chk(e.lib.tfhe_amd_bootstrap_woks(e.ctx, ud.ptr, mu, xd.ptr, B))
            e._chk(e.lib.tfhe_amd_keyswitch(e.ctx, od.ptr, ud.ptr, B))

        for k in range(2 * max(1, a.warmup)):
            pstep(k)
        eng.sync()
        e2.sync()
        tp = time.perf_counter()
        for k in range(a.steps):
            pstep(k)
        eng.sync()
        e2.sync()
        p_elapsed = time.perf_counter() - tp
        same2 = bool(np.array_equal(o2_d.download(np.int32, (B, cfg.n + 1)), out_all))
        pipelined = {"value": B * a.steps / p_elapsed, "unit": "bootstraps/s", "ms_per_step": 1e3 * p_elapsed / a.steps,
                     "contexts": 2, "identical_to_single_context": same2}
        job2.close()

    if rank == 0:
        # HBM bytes per launch of the dominant kernel as measured by the PMC passes of an earlier profile of
        # this same command (tools/make_traffic.py -> profiles/traffic.json); null until such a profile exists
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and B == BATCH_PER_GPU and not a.lib:
            try:
                tj = json.load(open(tpath))
                # a PMC figure is only valid for the kernel sources it was measured on
                if tj.get("kernel_sources_sha256") == kernel_sources_sha256():
                    traffic, traffic_src = float(tj["bytes_per_launch"]), tj.get("source")
                else:
                    traffic_src = "profiles/traffic.json is stale (kernel sources changed since it was measured)"
            except (ValueError, KeyError):
                pass
        total = total_per_step * a.steps
        algo_bytes = B * cfg.n * BYTES_PER_CMUX + cfg.n * BYTES_PER_ROW
        persistent_bytes = B * ((cfg.n + 1) * 4 + (cfg.N + 1) * 4) + cfg.n * BYTES_PER_ROW  # SURVEY 8(d), persistent variant
        achieved = algo_bytes / (br_ms * 1e-3)
        flops = B * cfg.n * FLOP_PER_CMUX / (br_ms * 1e-3)
        fp64_instr_rate = B * cfg.n * FP64_INSTR_PER_CMUX / (br_ms * 1e-3)   # wave64 fp64 instructions per second, whole chip
        fp64_instr_peak = SIMDS * CLOCK_HZ / FP64_ISSUE_CYCLES
        line = {
            "metric": "gate bootstraps/sec (N=1024, 128-bit params)",
            "value": total / elapsed,
            "unit": "bootstraps/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64 (anticyclic FFT) over int32 torus",
            "data": "synthetic",
            "ranks_seen": ranks_seen,
            "n_devices": n_devices,
            "ranks": ranks,
            # every rank's own rate over the timed region (its samples / its seconds): the spread says whether one GPU holds the rest up
            "per_gpu_bootstraps_per_s": per_gpu_rates(ranks, a.steps),
            "efficiency_vs": solo,
            "config": {"workload": f"{baseline_config}: {total_per_step} gate bootstraps per step ({B} on rank 0), "
                                   f"{cfg.describe()}, persistent blind-rotation kernel + key-switch kernel, inputs resident in HBM",
                       "baseline_config": baseline_config,
                       "batch_per_gpu": B, "total_per_step": total_per_step,
                       "parallelism": f"batch-sharded x{world} (contiguous slices), keys replicated, no data-path collective",
                       "key_replication": ("one broadcast per key of the device-layout bytes from rank 0" if a.broadcast_keys
                                           else "regenerated from the seed on every rank"),
                       "scaling_note": "--gpus 1 without flags is config 2 (4096 per step); --gpus N > 1 without flags is config 5 "
                                       "(2^20 per step cut into N slices, strong); a strict strong-scaling series takes its "
                                       "N = 1 point from `--gpus 1 --total 1048576` (per-GPU throughput at 4096 and at 2^20 per "
                                       "launch agree within 1 %: profiles/r05_bench_selflaunch_world1.json)",
                       "process_group": (f"torch.distributed {a.backend}, world {world}" if dist is not None else "none (single process)"),
                       "ks_kernel": "gather" if a.ks_gather else "matrix-core (k_ks_mfma)",
                       "library": os.path.basename(a.lib) if a.lib else "libtfhe_amd.so"},
            # the persistent blind rotation is compute-bound -- its fraction is flops (173,056 per CMux per sample)
            # / HIP-event kernel time against the fp64 vector peak (78.6 TF; the f64 matrix rate of this chip is the same).
            # `hbm_contract` keeps 8(d)'s byte accounting of the one-launch-per-CMux schedule (bytes the persistent kernel never
            # moves: the accumulators stay in LDS), `traffic` is what the PMC counters saw, `fp64_issue` is the builder's own
            # diagnostic: wave64 fp64 instructions (2,144 per CMux, the floor of the bit-exact radix-2 DAG) against the issue rate.
            "roofline": {"bound": "fp64_valu", "kernel": "k_blind_rotate<int32,N=1024>",
                         "achieved": flops / 1e12, "peak": FP64_PEAK / 1e12, "unit": "TFLOP/s", "frac": flops / FP64_PEAK,
                         "flops_frac": flops / FP64_PEAK, "flop_per_cmux": FLOP_PER_CMUX,
                         "cmux_per_s": B * cfg.n / (br_ms * 1e-3),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "persistent_algorithmic_bytes": persistent_bytes,
                         "traffic_over_persistent_algorithmic_bytes": None if traffic is None else traffic / persistent_bytes,
                         "traffic_over_hbm_contract_bytes": None if traffic is None else traffic / algo_bytes,
                         "kernel_ms": br_ms,
                         "note": "frac = B x n x 173,056 flop / kernel time / 78.6 TFLOP/s (SURVEY 8d, persistent variant: compute-bound; "
                                 "MFMA is not the bound). The bit-exact DAG is adds, multiplies and fmas in the ratio 913:496:638 per "
                                 "CMux, so the same instruction stream at one fp64 instruction per 4 cycles tops out at 0.63 of this peak",
                         "hbm_contract": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                          "frac": achieved / HBM_PEAK, "algorithmic_bytes_per_launch": algo_bytes,
                                          "note": "north-star accounting (SURVEY 8d): 16,388 B per CMux per sample + 65,536 B key "
                                                  "row per CMux per launch, as if every CMux were its own launch; the persistent "
                                                  "kernel does not move these bytes (see traffic)"},
                         "fp64_issue": {"bound": "fp64_issue", "achieved": fp64_instr_rate / 1e9, "peak": fp64_instr_peak / 1e9,
                                        "unit": "G fp64 wave-instr/s", "frac": fp64_instr_rate / fp64_instr_peak,
                                        "fp64_wave_instr_per_cmux": FP64_INSTR_PER_CMUX,
                                        "floor_cmux_per_s": fp64_instr_peak / FP64_INSTR_PER_CMUX,
                                        "note": "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 fp64 instruction; 2,144 such "
                                                "instructions per CMux per sample is the floor of the reference's radix-2 DAG "
                                                "reproduced bit for bit (docs/experiments.md, measured bound of the gate kernel)"}},
            "kernels_ms": {"blind_rotate_extract": br_ms, "keyswitch": ks_ms},
            "decrypt_check": bool(ok),
            "oracle_bit_check": None if oracle_want is None else {"samples": len(oracle_idx), "identical": oracle_ok},
            "tail_check": None if tail_ok is None else {"samples": ntail, "identical_to_front": tail_ok},
            "device": T.device_info(device, a.lib),
        }
        if a.broadcast_keys:
            line["key_broadcast"] = {"bytes": job.key_bytes_received, "backend": a.backend, "world": world, "setup_seconds": round(t_keys, 3),
                                     "what": "bkFFT in the kernel layout + key-switch key, torch.distributed.broadcast from rank 0"}
        if sustained is not None:
            line["sustained"] = sustained
        if pool_check is not None:
            line["pool_check"] = pool_check
        if latency is not None:
            line["config1_latency"] = latency
        if streamed is not None:
            line["streamed_schedule"] = streamed
        if other is not None:
            line.update(other)
        if pipelined is not None:
            line["pipelined_two_contexts"] = pipelined
        if cpu_line is not None:
            cpu_line = dict(cpu_line, measured_by=baseline_by)
            line["cpu_baseline"] = cpu_line
        detail_path = write_detail(line, a)
        try:
            compact = compact_line(line, detail_path)
        except Exception as e:  # noqa: BLE001 -- the contract keys alone then (the full record is in the detail file)
            compact = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                            "scaling", "vs_baseline", "dtype", "data")}
            compact.update(config={"workload": line["config"]["workload"]}, roofline=line["roofline"], cpu_baseline=line.get("cpu_baseline"),
                           compact_line_error=repr(e)[:200], detail=detail_path)
        print(json.dumps(compact), flush=True)
    job.close()
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("decrypt check failed")
    if not extras_ok:
        raise SystemExit("a schedule measured after the timed region (config 1 latency / streamed / pool) differs from the headline outputs")
    if oracle_ok is False:
        raise SystemExit("GPU outputs differ from the oracle")
    if tail_ok is False:
        raise SystemExit("outputs at the end of the batch differ from the same inputs at its front")


if __name__ == "__main__":
    main()
