#!/usr/bin/env python3
"""One purpose: the PCIe-inclusive rate of the C ABI's host-pointer form (tfhe_amd_bootstrap_host: copy in, blind rotation +
key switch, copy out, sync) on 4096 gate bootstraps (run ON THE GPU BOX).  Never the bench's `value`."""
import importlib, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
T = importlib.import_module("experimental-tfhe_amd")
shard = importlib.import_module("experimental-tfhe_amd.shard")
cfg = shard.GateConfig()
job = shard.GateJob(cfg, 0x5446484500000001, device=0)
eng = job.eng
x = shard.synthetic_samples(cfg, 4096, seed=3)
out = eng.bootstrap_host(1 << 29, x)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = eng.bootstrap_host(1 << 29, x); ts.append(time.perf_counter() - t0)
print("tfhe_amd_bootstrap_host, 4096 samples: min %.3f ms median %.3f ms -> %.0f bootstraps/s" % (1e3*min(ts), 1e3*sorted(ts)[2], 4096/sorted(ts)[2]))
ref = eng.bootstrap(1 << 29, x)
print("identical to resident path:", bool(np.array_equal(out, ref)))
job.close()
