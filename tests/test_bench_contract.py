"""bench.py end to end on the CPU emulator build of the engine (tests/emu): the JSON line must carry
every field of the driver's contract plus `roofline` and `cpu_baseline`, and the decrypt check
inside bench.py must hold.  Numbers are meaningless here (the emulator is ~10^5 times slower than
the GPU); the GPU run of the same command is the measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def test_bench_line_on_emulator(emu_lib, tmp_path):
    detail = str(tmp_path / "detail.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", emu_lib, "--batch", "2", "--steps", "1",
                          "--warmup", "0", "--cpu-seconds", "1", "--headline-only", "--detail", detail], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    assert len(lines[0]) < 3500, "the stdout line must stay compact: a stored tail of the run's stdout has to hold all of it"
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0 and d["higher_is_better"] is True
    assert d["unit"] == "bootstraps/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # SURVEY 8(d), persistent variant: compute-bound, flops against the fp64 vector peak; the north-star HBM accounting and the
    # issue-slot diagnostic ride along as single fractions (their full objects are in the detail file)
    assert r["bound"] == "fp64_valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["peak"] - 78.6) < 1e-9 and "traffic" in r
    assert abs(r["achieved"] * 1e12 - 2 * 630 * 173056 / (r["kernel_ms"] * 1e-3)) < 1e-6 * r["achieved"] * 1e12
    assert r["persistent_algorithmic_bytes"] == 2 * (2524 + 4100) + 630 * 65536
    # 16,388 B per CMux per sample + 65,536 B key row per CMux per launch, n = 630 CMux
    assert r["hbm_contract_bytes"] == 2 * 630 * 16388 + 630 * 65536
    assert abs(r["hbm_contract_frac"] - r["hbm_contract_bytes"] / (r["kernel_ms"] * 1e-3) / 8.0e12) < 1e-9
    assert abs(r["fp64_issue_frac"] - 2 * 630 * 2144 / (r["kernel_ms"] * 1e-3) / (1024 * 2.4e9 / 4)) < 1e-9
    assert d["ranks_seen"] == 1 and "ranks" not in d
    # one rank: its own rate IS the value; no solo reference
    pg = d["per_gpu_bootstraps_per_s"]
    assert pg["min"] == pg["max"] == pg["mean"] and abs(pg["mean"] - d["value"]) < 0.02 * d["value"] + 0.06 and d["efficiency_vs"] is None  # (the line rounds to 0.1)
    assert "config1_latency" not in d and "config2_streamed" not in d and "sustained" not in d and "pool_check" not in d  # --headline-only
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "bootstraps/s" and c["sample"]
    assert d["checks"]["decrypt"] is True
    assert abs(d["value"] - 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # the full record behind the line
    assert d["detail"] == detail
    f = json.load(open(detail))
    assert f["value"] == d["value"] and f["roofline"]["frac"] == r["frac"] and f["roofline"]["flop_per_cmux"] == 173056
    i, h = f["roofline"]["fp64_issue"], f["roofline"]["hbm_contract"]
    assert i["bound"] == "fp64_issue" and abs(i["frac"] - i["achieved"] / i["peak"]) < 1e-12 and i["fp64_wave_instr_per_cmux"] == 2144
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12
    assert f["ranks"] == [{"rank": 0, "device": 0, "batch": 2, "seconds": f["ranks"][0]["seconds"], "pci": "0000:e0:00.0"}]
    assert d["n_devices"] == 1 and d["pci"] == "0000:e0:00.0"
    assert f["decrypt_check"] is True and "scaling_note" in f["config"]


def test_bench_multi_rank_path_on_emulator(emu_lib):
    """the world > 1 branch of bench.py (torch first, process group, barrier, max over ranks on a tensor, --total
    slicing, cpu_baseline on rank 0) under torch.distributed.run with two ranks: gloo + the emulator build here,
    nccl (= RCCL) + the HIP library on the GPU boxes"""
    port = 29600 + (os.getpid() % 300)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--lib", emu_lib, "--backend", "gloo", "--total", "5", "--steps", "1", "--warmup", "0",
                          "--cpu-seconds", "1", "--lwe-n", "24", "--sustained-seconds", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    assert [ln for ln in out.stdout.splitlines() if ln.strip()] == lines, "stdout must hold the one JSON line and nothing else"
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["ranks_seen"] == 2 and [(r[0], r[2]) for r in d["ranks"]] == [(0, 3), (1, 2)]  # [rank, device, batch, seconds, pci]
    # rank r runs on emulated device r (conftest: TFHE_EMU_DEVICES=8; the emulator aborts on cross-device operands): two devices
    assert [(r[1], r[4]) for r in d["ranks"]] == [(0, "0000:e0:00.0"), (1, "0000:e1:00.0")] and d["n_devices"] == 2
    assert "rank 0, before torch" in d["cpu_baseline"]["measured_by"]
    assert d["config"]["total_per_step"] == 5 and d["config"]["batch_per_gpu"] == 3  # ragged: 3 + 2
    assert "gloo, world 2" in d["config"]["process_group"]
    assert d["checks"]["decrypt"] is True
    assert abs(d["value"] - 5 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # the series describes itself: every rank's own rate (samples / its seconds over the timed region) and what rank 0 does
    # ALONE on its slice afterwards (the N = 1 point at the same per-GPU batch)
    rates = [r[2] * d["steps"] / r[3] for r in d["ranks"]]
    pg = d["per_gpu_bootstraps_per_s"]
    assert abs(pg["min"] - min(rates)) < 0.01 * pg["min"] + 0.06 and abs(pg["max"] - max(rates)) < 0.01 * pg["max"] + 0.06  # (rounded on the line)
    ev = d["efficiency_vs"]
    assert ev["batch"] == 3 and ev["per_gpu_bootstraps_per_s"] > 0 and "alone" in ev["what"]


def test_bench_dist_flag_single_rank_on_emulator(emu_lib):
    """--dist: the same branch with world size 1 (what a 1-GPU box can run of the driver's N > 1 launch)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + (os.getpid() % 90)))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", emu_lib, "--dist", "--backend", "gloo", "--total", "2",
                          "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--lwe-n", "24", "--headline-only"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and "gloo, world 1" in d["config"]["process_group"]
    assert "cpu_baseline" not in d


def test_bench_starts_its_own_ranks_on_emulator(emu_lib):
    """`bench.py --gpus 2` with NO launcher: the GPU-free parent takes the CPU baseline, starts the two ranks as a child
    torch.distributed.run and relays rank 0's line -- n_gpus 2, ranks_seen 2 (the way the driver launches --gpus 1)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TFHE_BENCH_HANDOFF")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lib", emu_lib, "--backend", "gloo",
                          "--total", "5", "--steps", "1", "--warmup", "0", "--cpu-seconds", "1", "--lwe-n", "24", "--sustained-seconds", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "strong" and d["n_devices"] == 2
    assert d["sustained"]["steps"] >= 2 and d["sustained"]["seconds"] >= 1.0 and d["sustained"]["step_ms_max"] >= d["sustained"]["step_ms_min"] > 0
    assert d["sustained"]["shader_clock_ghz"] is None  # (the probe build exists for the GPU library only)
    assert [(r[0], r[2]) for r in d["ranks"]] == [(0, 3), (1, 2)]
    assert "parent" in d["cpu_baseline"]["measured_by"] and d["cpu_baseline"]["value"] > 0
    assert "GPU-free parent" in d["config"]["launched_by"]
    assert d["checks"]["decrypt"] is True
    assert abs(d["value"] - 5 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_eight_self_launched_ranks_on_emulator(emu_lib):
    """the shape of the driver's largest run -- `bench.py --gpus 8`, no launcher -- with gloo and the emulator build: eight ranks,
    a ragged contiguous split (19 = 3 + 3 + 3 + 2 x 5), every rank counted"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TFHE_BENCH_HANDOFF")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--lib", emu_lib, "--backend", "gloo",
                          "--total", "19", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--lwe-n", "8", "--sustained-seconds", "0"], capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["scaling"] == "strong"
    assert [(r[0], r[2]) for r in d["ranks"]] == [(0, 3), (1, 3), (2, 3), (3, 2), (4, 2), (5, 2), (6, 2), (7, 2)]
    assert d["config"]["total_per_step"] == 19 and d["checks"]["decrypt"] is True
    assert [r[1] for r in d["ranks"]] == list(range(8)) and d["n_devices"] == 8 and len({r[4] for r in d["ranks"]}) == 8  # rank r on emulated device r
    assert abs(d["value"] - 19 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["ms_per_step"] * 1e-3 >= max(r[3] for r in d["ranks"]) - 1e-3  # the MAX over ranks is what is reported
    pg, ev = d["per_gpu_bootstraps_per_s"], d["efficiency_vs"]
    assert 0 < pg["min"] <= pg["mean"] <= pg["max"] and ev["batch"] == 3 and ev["per_gpu_bootstraps_per_s"] > 0
    # (value <= sum of the ranks' own rates: the slowest rank sets the time)
    assert d["value"] <= sum(r[2] * d["steps"] / r[3] for r in d["ranks"]) * 1.01


def test_bench_never_reports_ranks_that_did_not_run(emu_lib):
    """--gpus 2 with the self-launch disabled and no launcher, and --gpus 2 under a one-rank world: non-zero exit, no line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TFHE_BENCH_HANDOFF")}
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lib", emu_lib, "--backend", "gloo", "--total", "5",
            "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--lwe-n", "24", "--sustained-seconds", "0"]
    out = subprocess.run(args + ["--no-self-launch"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in out.stderr
    out = subprocess.run(args, capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "WORLD_SIZE=1" in out.stderr


def test_bench_post_region_sections_on_emulator(emu_lib):
    """what the default one-GPU command measures after the timed region: BASELINE config 1 (latency through
    tfhe_amd_bootstrap) and config 2's literal schedule (one launch per CMux), both bit-compared with the headline outputs"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", emu_lib, "--batch", "2", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline", "--extras-reps", "1", "--latency-batches", "1,2", "--lwe-n", "24",
                          "--other-configs", "4,ring", "--sustained-seconds", "1"],  # (config 3's 2049-coefficient private key switch is minutes on the emulator)
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    lat = d["config1_latency"]
    assert lat["latency_batch1_ms"] > 0 and lat["latency_batch2_ms"] > 0 and lat["identical_to_headline"] is True
    st = d["config2_streamed"]
    assert st["launches"] == 24 + 3 and st["identical_to_persistent"] is True and st["hipgraph_bootstraps_per_s"] > 0
    c4 = d["config4_transforms"]["hbm_frac [reverse_int, reverse_torus64, direct_torus64, direct_torus32 | Real96: iFFT, FFT]"]
    assert any(k.startswith("N=2048") for k in c4) and any(k.startswith("N=1024") for k in c4) and any(k.startswith("Real96") for k in c4)
    assert "config3_circuit_bootstrap" not in d
    rg = d["ring_degrees"]["[bootstraps/s, blind-rotation TF/s, oracle_bit_identical] by N"]  # the generic kernels, from the same child
    assert set(rg) == {"512", "4096"} and all(v[0] > 0 and v[2] is True for v in rg.values())
    # the pool: one member, and two members sharing the device; host arrays in and out; identical to the headline outputs
    pc = d["pool_check"]
    rows = pc["[devices, samples per call, bootstraps/s]"]
    assert [(m[0], m[1]) for m in rows] == [([0], 2), ([0, 0], 2), ([0], 8)] and all(m[2] > 0 for m in rows) and pc["identical_to_headline"] is True
    su = d["sustained"]
    assert su["steps"] >= 2 and su["bootstraps_per_s"] > 0 and su["step_ms_max"] >= su["step_ms_min"] > 0
    f = json.load(open(os.path.join(ROOT, d["detail"])))  # the full record: every stage and line
    assert "child process" in f["other_configs_run"] and f["pool_check"]["pools"][1]["split"] == [1, 1]
    assert f["streamed_schedule"]["roofline"]["bound"] == "hbm" and f["streamed_schedule"]["roofline"]["algorithmic_bytes_per_launch"] == 2 * 16388 + 65536
    assert f["streamed_schedule"]["hipgraph"].get("identical_to_persistent") is True
    assert f["config1_latency"]["batch1_identical_to_headline_outputs"] is True
    assert any("execute_reverse_torus64 N=2048" in ln["workload"] for ln in f["config4_transforms"]["lines"])
    assert abs(d["value"] - 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # never inside `value`


def test_pool_over_every_visible_device_on_emulator(emu_lib):
    """`bench.py --pool-devices all`: ONE process driving every device the engine library shows (here the emulator's 8) through
    tfhe_amd_pool_* -- keys uploaded from host arrays to all eight, the batch cut into eight slices, outputs identical to the
    headline's; what a maintainer runs on an 8-GPU node beside `--gpus 8`"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", emu_lib, "--batch", "5", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--lwe-n", "6", "--headline-only", "--pool-devices", "all"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    rows = d["pool_check"]["[devices, samples per call, bootstraps/s]"]
    assert rows[-1][0] == list(range(8)) and rows[-1][1] == 20 and rows[-1][2] > 0 and d["pool_check"]["identical_to_headline"] is True
    f = json.load(open(os.path.join(ROOT, d["detail"])))
    last = f["pool_check"]["pools"][-1]
    assert last["split"] == [3, 3, 3, 3, 2, 2, 2, 2] and len(set(last["pci"])) == 8


def test_two_ranks_on_one_device_are_reported_as_such(emu_lib):
    """two gloo ranks that share ONE (emulated) device: the line says n_gpus 2 (ranks) but n_devices 1 and shows the same PCI
    bus id twice -- under RCCL the same situation exits non-zero (bench.py refuses to call two ranks on one chip two GPUs)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TFHE_BENCH_HANDOFF")}
    env["TFHE_EMU_DEVICES"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lib", emu_lib, "--backend", "gloo",
                          "--total", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--lwe-n", "8", "--sustained-seconds", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["n_devices"] == 1
    assert [(r[1], r[4]) for r in d["ranks"]] == [(0, "0000:e0:00.0"), (0, "0000:e0:00.0")]


def test_broadcast_keys_two_ranks_on_emulator(emu_lib):
    """--broadcast-keys: only rank 0 builds the keys; rank 1 receives the bytes of the device layout by torch.distributed
    broadcast (gloo here, RCCL on GPUs), imports them on ITS device (emulated device 1) and bootstraps real encryptions that
    must decrypt (every rank's own check) -- rank 0's outputs bit-identical to the oracle"""
    port = 29300 + (os.getpid() % 200)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--lib", emu_lib, "--backend", "gloo", "--batch", "18", "--steps", "1", "--warmup", "0",
                          "--cpu-seconds", "1", "--lwe-n", "6", "--sustained-seconds", "0", "--broadcast-keys"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    kb = d["key_broadcast"]
    assert kb["bytes"] == 6 * 4 * 2 * 1024 * 8 + 1024 * 8 * 4 * 7 * 4 and kb["world"] == 2 and kb["backend"] == "gloo"
    assert "broadcast" in d["config"]["parallelism"]
    assert d["checks"]["decrypt"] is True and d["checks"]["oracle_bit_identical"] is True and d["n_devices"] == 2


def test_smoke_logic_on_emulator(emu_lib, monkeypatch):
    """__graft_entry__.smoke() with the engine library swapped for the emulator build: the same
    calls, keys and oracle comparison the driver runs on cuda:0"""
    import importlib
    T = importlib.import_module("experimental-tfhe_amd")
    orig = T.load_library
    monkeypatch.setattr(T, "load_library", lambda path=None: orig(emu_lib))
    sys.path.insert(0, ROOT)
    G = importlib.import_module("__graft_entry__")
    G.smoke()
