"""Staged GPU sanity run (manual tool, not a pytest module): each stage prints before it
starts, so a hang or fault can be attributed.  Usage: python tests/gpu_stage_check.py [max_stage]"""
import importlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle_py as O  # noqa: E402

O.lib()
T = importlib.import_module("experimental-tfhe_amd")
if os.environ.get("TFHE_AMD_TEST_LIB"):  # check an experiment build (tools/ab.py) instead
    T.DEFAULT_LIB = os.path.abspath(os.environ["TFHE_AMD_TEST_LIB"])
import parity_checks as P  # noqa: E402

max_stage = int(sys.argv[1]) if len(sys.argv) > 1 else 99


def say(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


say("device:", T.device_info(0))
say("stage 1: context + N=1024 ifft of 4 polynomials")
e = T.Engine(torus_bits=32, n=1, N=1024, l=2, Bgbit=10, ks_t=0)
rs = np.random.RandomState(0)
a = rs.randint(-2 ** 31, 2 ** 31, size=(4, 1024)).astype(np.int32)
got = e.ifft_int32(a)
say("   ifft bit-exact:", P.same_doubles(got, O.execute_reverse_int(1024, a)))
e.close()
if max_stage >= 2:
    say("stage 2: FFT plugin checks, N=1024 and 2048")
    P.check_fft_plugin(T.DEFAULT_LIB, 1024, count=5)
    P.check_fft_plugin(T.DEFAULT_LIB, 2048, count=5)
    say("   ok")
if max_stage >= 3:
    say("stage 3: gate path, n=4, B=3")
    P.check_gate_path(T.DEFAULT_LIB, N=1024, n=4, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=3)
    say("   ok")
if max_stage >= 4:
    say("stage 4: gate path, n=630, B=4, on the latency-shaped kernel and on one wave per ciphertext")
    P.check_gate_path(T.DEFAULT_LIB, N=1024, n=630, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=4, check_export=False, br_split=1 << 30)
    P.check_gate_path(T.DEFAULT_LIB, N=1024, n=630, l=2, Bgbit=10, ks_t=8, ks_bb=2, B=4, check_export=False, br_split=0)
    say("   ok")
if max_stage >= 5:
    say("stage 5: Torus64 N=2048 l=4, n=5, B=4")
    P.check_torus64_path(T.DEFAULT_LIB, N=2048, n=5, l=4, Bgbit=9, B=4)
    say("   ok")
if max_stage >= 6:
    say("stage 6: key switch shapes (matrix-core + gather), n_out=630 8x2 and n_out=500 6x2")
    P.check_keyswitch_shapes(T.DEFAULT_LIB, 1024, 630, 8, 2, 35)
    P.check_keyswitch_shapes(T.DEFAULT_LIB, 1024, 500, 6, 2, 9)
    say("   ok")
if max_stage >= 7:
    say("stage 7: run-time gadget instantiations and rounding extremes")
    P.check_gate_path(T.DEFAULT_LIB, N=1024, n=8, l=2, Bgbit=9, ks_t=8, ks_bb=2, B=9, check_export=False)
    P.check_gate_path(T.DEFAULT_LIB, N=1024, n=5, l=4, Bgbit=6, ks_t=8, ks_bb=2, B=5, check_export=False)
    P.check_rounding_extremes(T.DEFAULT_LIB)
    say("   ok")
if max_stage >= 8:
    say("stage 8: circuit bootstrap pipeline (private key switch), N2=2048")
    P.check_circuit_bootstrap(T.DEFAULT_LIB, n0=6, N1=1024, N2=2048, l1=2, bg1=8, l2=4, bg2=9, t10=6, bb10=2, t21=2,
                              bb21=3, B=5)
    say("   ok")
if max_stage >= 9:
    say("stage 9: CMux on data + LUT evaluation by vertical packing (d = 3, 10, 13)")
    P.check_cmux_data(T.DEFAULT_LIB, B=9)
    for d in (3, 10, 13):
        P.check_lut_eval(T.DEFAULT_LIB, d=d, B=4)
    say("   ok")
if max_stage >= 10:
    say("stage 10: ragged transform batches; streamed schedule replayed from a hipGraph")
    for N in (1024, 2048):
        P.check_fft_plugin(T.DEFAULT_LIB, N, count=11)
    P.check_streamed_graph(T.DEFAULT_LIB, n=24, B=9)
    say("   ok")
if max_stage >= 11:
    say("stage 11: exact (FFT-free) external product, Torus32 N=1024 and Torus64 N=2048")
    say("   fp64 vs exact, worst difference:", P.check_exact_extprod(T.DEFAULT_LIB, 32, 1024, 2, 10, B=5, fft_bound=4),
        P.check_exact_extprod(T.DEFAULT_LIB, 64, 2048, 4, 9, B=5, fft_bound=2 ** 32))
if max_stage >= 12:
    say("stage 12: Real96 high-precision transforms, N=2048")
    import test_hp_fft
    test_hp_fft.check_hp_kernels(T.DEFAULT_LIB, 2048, B=5)
    say("   ok")
say("done")
