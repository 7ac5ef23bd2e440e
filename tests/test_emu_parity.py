"""Kernel logic vs the oracle on the CPU emulation build (tests/emu).  Same checks as the GPU
parity tests, at sizes the fiber emulator finishes in seconds.  These do NOT count as GPU
parity; they exist to catch kernel bugs before a GPU call and to run the kernels under
sanitizers."""
import pytest

import parity_checks as P


@pytest.mark.parametrize("N", [1024, 2048])
def test_fft_plugin_emu(emu_lib, N):
    P.check_fft_plugin(emu_lib, N, count=3)


@pytest.mark.parametrize("N", [1024, 2048])
def test_fft_plugin_ragged_emu(emu_lib, N):
    """ragged last workgroup (count % waves per workgroup != 0)"""
    P.check_fft_plugin(emu_lib, N, count=5)


def test_gate_path_emu_n1024(emu_lib):
    P.check_gate_path(emu_lib, N=1024, n=6, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=3)


@pytest.mark.parametrize("B", [1, 3])
def test_gate_path_emu_both_blind_rotation_kernels(emu_lib, B):
    """the same inputs through the latency-shaped kernel (one ciphertext per 4-wave workgroup) and through the
    one-wave-per-ciphertext kernel: both bit-identical to the oracle"""
    P.check_gate_path(emu_lib, N=1024, n=5, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=B, check_export=False, br_split=1 << 30)
    P.check_gate_path(emu_lib, N=1024, n=5, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=B, check_export=False, br_split=0)


def test_gate_path_emu_eight_wave_workgroups(emu_lib):
    """batches above 1024 run one wave per ciphertext in 8-wave workgroups (two per SIMD, the issue balance between
    partner waves): 1027 samples of a 2-step blind rotation = 128 full workgroups and a ragged one, against the
    latency-shaped kernel on the same inputs and the oracle on a subset"""
    import importlib
    import numpy as np
    import oracle_py as O
    T = importlib.import_module("experimental-tfhe_amd")
    s = P.GateSetup(emu_lib, 1024, 2, 2, 10, 8, 2)
    try:
        rs = np.random.RandomState(77)
        x = rs.randint(-2 ** 31, 2 ** 31, size=(1027, 3)).astype(np.int32)
        s.eng.set_option(T.OPT_BR_SPLIT, 0)
        got = s.eng.bootstrap_woks(1 << 29, x)
        sub = [0, 7, 8, 1023, 1024, 1026]
        want = np.stack([O.bootstrap_woks32(1024, s.bk, 1 << 29, x[i], 2, 10) for i in sub])
        assert np.array_equal(got[sub], want)
        s.eng.set_option(T.OPT_BR_SPLIT, 1 << 30)
        assert np.array_equal(s.eng.bootstrap_woks(1 << 29, x[:40]), got[:40])
    finally:
        s.close()


def test_gate_path_emu_split_kernel_runtime_bgbit(emu_lib):
    """k_blind_rotate_split<0> (Bgbit read at run time) and <8> (the circuit bootstrap's output gadget)"""
    P.check_gate_path(emu_lib, N=1024, n=3, l=2, Bgbit=9, ks_t=8, ks_bb=2, B=2, check_export=False, seed=12, br_split=1 << 30)
    P.check_gate_path(emu_lib, N=1024, n=3, l=2, Bgbit=8, ks_t=8, ks_bb=2, B=2, check_export=False, seed=14, br_split=1 << 30)


def test_rounding_extremes_emu(emu_lib):
    P.check_rounding_extremes(emu_lib)


@pytest.mark.parametrize("N,l,Bgbit", [(1024, 4, 9), (2048, 4, 9)])
def test_rounding_extremes_torus64_emu(emu_lib, N, l, Bgbit):
    """the short Torus64 rounding sequence, its guard and the exact fallback (both blind-rotation shapes)"""
    P.check_rounding_extremes64(emu_lib, N=N, l=l, Bgbit=Bgbit)


def test_gate_path_emu_runtime_gadget(emu_lib):
    """l = 2 with a Bgbit the kernel has no compile-time instantiation for, and l = 4 (two digit pairs per polynomial)"""
    P.check_gate_path(emu_lib, N=1024, n=3, l=2, Bgbit=9, ks_t=8, ks_bb=2, B=3, check_export=False, seed=12)
    P.check_gate_path(emu_lib, N=1024, n=2, l=4, Bgbit=6, ks_t=8, ks_bb=2, B=2, check_export=False, seed=13)


def test_gate_path_emu_other_gadgets(emu_lib):
    P.check_gate_path(emu_lib, N=1024, n=3, l=3, Bgbit=7, ks_t=16, ks_bb=1, B=2, seed=5)
    P.check_gate_path(emu_lib, N=1024, n=2, l=1, Bgbit=12, ks_t=5, ks_bb=3, B=9, seed=6)  # ragged: 9 = 8 + 1 waves


def test_gate_path_emu_n2048(emu_lib):
    P.check_gate_path(emu_lib, N=2048, n=3, l=2, Bgbit=9, ks_t=4, ks_bb=3, B=5)


@pytest.mark.parametrize("n_out,t,bb,B", [(630, 8, 2, 17), (500, 6, 2, 3), (630, 16, 1, 2)])
def test_keyswitch_real_shapes_emu(emu_lib, n_out, t, bb, B):
    """gate key switch 8x2 and 16x1 (n=630) and the PoC's preKeySwitch 6x2 (n0=500): matrix-core + gather kernels"""
    P.check_keyswitch_shapes(emu_lib, 1024, n_out, t, bb, B)


def test_streamed_schedule_fixed_buffers_emu(emu_lib):
    P.check_streamed_graph(emu_lib, n=4, B=3)


def test_streamed_graph_without_prior_plain_call_emu(emu_lib):
    """graph mode as the FIRST thing a context does, at two batch sizes (on the GPU: two blind-rotation kernel classes)"""
    P.check_streamed_graph_batch_classes(emu_lib, n=2, big=20, small=3)


def test_streamed_graph_zero_step_class_emu(emu_lib):
    """the latency-shaped kernel for every batch (split limit 2^30), a batch of 3 and then of 1030: the 1-step launches keep their
    class, the 0-step launches move from the 4-wave to the 8-wave form, whose LDS attribute must be set outside the capture"""
    P.check_streamed_graph_batch_classes(emu_lib, n=2, big=1030, small=3, br_split=1 << 30, order=(3, 1030))


def test_gate_wide_batch_logic_emu(emu_lib):
    """the wide-batch check of the GPU suite (there: B = 1031, 8-wave workgroups) at a size the emulator finishes"""
    P.check_gate_wide_batch(emu_lib, l=3, Bgbit=7, B=30, n=2)


@pytest.mark.parametrize("bits,N,l,Bgbit,bound", [(32, 1024, 2, 10, 4), (64, 2048, 4, 9, 2 ** 32)])
def test_exact_external_product_emu(emu_lib, bits, N, l, Bgbit, bound):
    """FFT-free backend (poc:285-316) bit-exact vs the oracle; fp64 path within `bound` units of it"""
    P.check_exact_extprod(emu_lib, bits, N, l, Bgbit, B=2, fft_bound=bound)


def test_cmux_on_data_emu(emu_lib):
    P.check_cmux_data(emu_lib, B=5)


@pytest.mark.parametrize("d,B", [(3, 2), (10, 2), (12, 3)])
def test_lut_eval_emu(emu_lib, d, B):
    """vertical packing: fewer bits than log2 N (rotation only), exactly log2 N, and a 2-level CMux tree"""
    P.check_lut_eval(emu_lib, d=d, B=B)


def test_lut_eval_poc_gadget_emu(emu_lib):
    """the circuit bootstrap's output gadget (l1=2, Bgbit1=8, poc:70-85): bit-compare only"""
    P.check_lut_eval(emu_lib, l=2, Bgbit=8, d=11, B=2, decrypt_tol=None)


def test_circuit_bootstrap_emu(emu_lib):
    # l2 = 4, Bgbit2 = 9, privKS base 8 as in the PoC; short n0 / key-switch lengths and N2 = 1024 (every
    # readlane of the 32-sample private-key-switch tile is a fiber rendezvous here; the N2 = 2048 Torus64
    # path has its own test below, the full pipeline shape runs in the GPU suite)
    P.check_circuit_bootstrap(emu_lib, n0=2, N1=1024, N2=1024, l1=2, bg1=8, l2=4, bg2=9, t10=3, bb10=2, t21=2,
                              bb21=3, B=2)


def test_privks_wide_logic_emu(emu_lib):
    """the wide-count private-key-switch check of the GPU suite at a size the emulator finishes: 259 samples = one full
    256-sample tile + a ragged one, int64 inputs, base 8"""
    P.check_privks_wide(emu_lib, N2=1024, t21=2, bb21=3, counts=(259,), pipeline_B=0, n0=2, l2=3, bg2=10, t10=2, planes=(1,))


@pytest.mark.parametrize("N,l,Bgbit,B", [(2048, 4, 9, 4), (1024, 3, 10, 5)])
def test_torus64_path_emu(emu_lib, N, l, Bgbit, B):
    P.check_torus64_path(emu_lib, N=N, n=3, l=l, Bgbit=Bgbit, B=B)


def test_abi_edges_emu(emu_lib):
    """empty batches, calls in the wrong state, bad arguments (status codes of include/tfhe_amd.h)"""
    P.check_abi_edges(emu_lib)
