"""Static properties of the kernels' LDS layout (no GPU): the padded transpose indices are
permutations and, under the gfx950 banking rules of MI355X_MICROARCH.md, free of bank conflicts
for every ds_write_b64 / ds_read_b64 of both transposes in both directions."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft_transposes_are_conflict_free_permutations():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lds_conflicts.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "all conflict-free" in out.stdout


def test_lds_budgets_fit_160k():
    # mirrors BlindRotateLds / KsTiledLds of csrc/tfhe_kernels.h and the WAVES chosen in tfhe_amd.hip
    def br(torus_bytes, N, waves, twreg=False):
        NC = N // 2
        return (0 if twreg else 16 * 2 * NC) + waves * (torus_bytes * 2 * N + 8 * (NC + 64))
    assert br(4, 1024, 8) == 118784 <= 163840
    assert br(4, 1024, 4, True) <= 163840
    assert br(4, 2048, 4) <= 163840
    assert br(8, 1024, 4) <= 163840
    assert br(8, 2048, 3) <= 163840
    assert 16 * 5 * 128 * 4 <= 65536  # tiled key switch reduction buffer needs no raised limit
