#!/usr/bin/env python3
"""One purpose: the FFT-plugin parity checks (tests/parity_checks.check_fft_plugin, both ring sizes) on a chosen build of
the engine library -- the correctness half of an A/B experiment (tools/bench_configs.py fft --lib ... is the timing half).
Run ON THE GPU BOX:   python tools/fft_check_lib.py build/lib_variant.so"""
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import parity_checks as P
lib=sys.argv[1]
for N in (1024,2048):
    P.check_fft_plugin(lib, N, count=37)
print("fft plugin parity ok", lib)
