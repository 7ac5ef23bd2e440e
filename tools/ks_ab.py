import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle_py as O
O.lib()
import parity_checks as P
T = importlib.import_module("experimental-tfhe_amd")
for args in ((1024, 630, 8, 2, 67), (1024, 500, 6, 2, 33), (1024, 630, 16, 1, 16), (1024, 700, 4, 3, 5), (1024, 630, 8, 2, 300)):
    P.check_keyswitch_shapes(T.DEFAULT_LIB, *args); print(args, "ok", flush=True)
# timing
e = T.Engine(torus_bits=32, n=630, N=1024, l=2, Bgbit=10, ks_t=8, ks_basebit=2)
rs = np.random.RandomState(1)
ks = rs.randint(-2**31, 2**31, size=(1024, 8, 4, 631)).astype(np.int32)
e.load_keyswitch_key(ks)
for B in (1, 256, 4096, 16384):
    x = e.to_device(rs.randint(-2**31, 2**31, size=(B, 1025)).astype(np.int32)); o = e.alloc(B*631*4)
    e0, e1 = e.event(), e.event()
    outs = {}
    for mode, name in ((0, "mfma"), (1, "gather")):
        if mode == 1 and B > 4096: continue
        e.set_option(T.OPT_KS_GATHER, mode)
        ts = []
        for rep in range(6):
            e.record(e0); e._chk(e.lib.tfhe_amd_keyswitch(e.ctx, o.ptr, x.ptr, B)); e.record(e1); ts.append(e.elapsed_ms(e0, e1))
        outs[name] = o.download(np.int32, (B, 631))
        print(f"B={B} {name}: min {min(ts[1:]):.4f} ms", flush=True)
    assert all(np.array_equal(v, outs["mfma"]) for v in outs.values())
e.close()
