import importlib, sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("experimental-tfhe_amd")
shard = importlib.import_module("experimental-tfhe_amd.shard")
cfg = shard.GateConfig()
job = shard.GateJob(cfg, 0x5446484500000001, device=0)
eng, lib = job.eng, job.eng.lib
eng.set_option(T.OPT_BR_SPLIT, 1 << 30)
ref = {}
for B in (1, 64, 256, 512, 1024):
    x = shard.synthetic_samples(cfg, B, seed=99)
    x_d = eng.to_device(x); u_d = eng.alloc(B * (cfg.N + 1) * 4)
    e0, e1 = eng.event(), eng.event()
    for form in (2, 4, 2, 4):
        eng.set_option(7, form)
        ts = []
        for _ in range(6):
            eng.record(e0); eng._chk(lib.tfhe_amd_bootstrap_woks(eng.ctx, u_d.ptr, 1 << 29, x_d.ptr, B)); eng.record(e1); eng.sync()
            ts.append(eng.elapsed_ms(e0, e1))
        out = u_d.download(np.int32, (B, cfg.N + 1))
        same = np.array_equal(out, ref.setdefault(B, out))
        print("B=%4d form %d: blind rotation min %.3f ms  identical %s" % (B, form, min(ts[1:]), same), flush=True)
job.close()
