"""Static properties of the kernels' LDS layout (no GPU): the padded transpose indices are
permutations and, under the gfx950 banking rules of MI355X_MICROARCH.md, free of bank conflicts
for every ds_write_b64 / ds_read_b64 (8-byte planes) and ds_write_b128 / ds_read_b128 (complex points,
N=1024) of both transposes in both directions."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft_transposes_are_conflict_free_permutations():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lds_conflicts.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "all conflict-free" in out.stdout


def test_lds_budgets_fit_160k():
    # mirrors BlindRotateLds of csrc/tfhe_kernels.h and the WAVES chosen in tfhe_amd.hip
    def br(torus_bytes, N, waves):
        NC = N // 2
        accreg = (N == 2048 and torus_bytes == 8)    # accumulator in registers, one scratch polynomial in LDS
        cplx = (N == 1024) or accreg                 # complex-point transposes
        xch = (16 if cplx else 8) * (NC + 64)
        sync = 64                                    # progress counters + SIMD ids (WaveLds::balance)
        if accreg:
            return 16 * 2 * NC + waves * max(torus_bytes * N, xch) + sync
        return 16 * 2 * NC + waves * (torus_bytes * 2 * N + xch) + sync
    assert br(4, 1024, 8) == 155648 + 64 <= 163840
    assert br(4, 2048, 4) <= 163840
    assert br(8, 1024, 4) <= 163840
    assert br(8, 2048, 4) == 32768 + 4 * 17408 + 64 <= 163840
    assert 2 * (2 * 8 * 4096) <= 163840  # k_ks_mfma: two workgroups per CU, double-buffered key slices of 8 K-steps


def test_generic_swizzle_of_the_tool_is_the_kernels():
    """tools/lds_conflicts.py checks ITS gen_sw: the expression must be the one in csrc/tfhe_kernels_generic.h"""
    import importlib.util
    import re
    src = open(os.path.join(ROOT, "experimental-tfhe_amd", "csrc", "tfhe_kernels_generic.h")).read()
    m = re.search(r"int gen_sw\(int j\) \{ return (.*?); \}", src)
    assert m, "gen_sw not found in the kernel header"
    spec = importlib.util.spec_from_file_location("lds_conflicts", os.path.join(ROOT, "tools", "lds_conflicts.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    expr = compile(m.group(1), "gen_sw", "eval")  # shifts, masks and XORs read the same in C and Python for j >= 0
    for j in range(1 << 14):
        assert eval(expr, {"j": j}) == tool.gen_sw(j), j
