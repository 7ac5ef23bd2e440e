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


def test_bench_line_on_emulator(emu_lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", emu_lib, "--batch", "2", "--steps", "1",
                          "--warmup", "0", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0 and d["higher_is_better"] is True
    assert d["unit"] == "bootstraps/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # 16,388 B per CMux per sample + 65,536 B key row per CMux per launch, n = 630 CMux
    assert r["algorithmic_bytes_per_launch"] == 2 * 630 * 16388 + 630 * 65536
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "bootstraps/s"
    assert d["decrypt_check"] is True
    assert abs(d["value"] - 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


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
