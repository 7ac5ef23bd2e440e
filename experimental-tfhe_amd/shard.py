"""Host-side job logic for batches of gate bootstraps on one or several GPUs.

The reference's only parallel construct is an independent-item loop
(parallel/src/test_parallel_multiplications.cpp:62 `#pragma omp parallel for`): ciphertexts
never interact.  The multi-GPU mapping is therefore: one process per GPU, the batch cut into
contiguous slices, keys REPLICATED (every rank regenerates the same keys from the same seed,
so not even a broadcast is needed), no collective on the data path.  torch.distributed is
used only for the timing barrier / max-over-ranks and for gathering results in tests.
"""
import importlib
from dataclasses import dataclass

import numpy as np

_T = importlib.import_module(__name__.rsplit(".", 1)[0])


@dataclass(frozen=True)
class GateConfig:
    """gate-bootstrap parameter set; defaults = BASELINE.json (n=630, N=1024, k=1, l=2) with
    Bgbit=10 (circuit-bootstrapping/misc/params-gb.html:124-131) and the key switch 8 x 2 bits"""
    N: int = 1024
    n: int = 630
    l: int = 2
    Bgbit: int = 10
    ks_t: int = 8
    ks_basebit: int = 2
    bk_stdev: float = 2.0 ** -25
    ks_stdev: float = 2.0 ** -15

    def describe(self):
        return (f"gate bootstrap n={self.n} N={self.N} k=1 l={self.l} Bgbit={self.Bgbit} "
                f"ks_t={self.ks_t} ks_basebit={self.ks_basebit} Torus32")


def shard_range(total, rank, world):
    """contiguous slice [lo, hi) of `total` items owned by `rank`; sizes differ by at most 1"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def synthetic_samples(cfg, count, seed):
    """uniformly random LWE samples ('synthetic random ciphertexts'): int32 [count][n+1]"""
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    return rs.randint(-2 ** 31, 2 ** 31, size=(count, cfg.n + 1), dtype=np.int64).astype(np.int32)


class GateJob:
    """keys + engine of one rank.  Secret keys, bootstrapping key and key-switch key are functions
    of (cfg, seed) only, so every rank holds identical replicas without communication."""

    def __init__(self, cfg, seed, device=0, lib_path=None):
        self.cfg, self.seed = cfg, seed
        self.lwe_key = _T.keygen_binary(cfg.n, seed, 1, lib_path=lib_path)
        self.tlwe_key = _T.keygen_binary(cfg.N, seed, 2, lib_path=lib_path)
        self.eng = _T.Engine(torus_bits=32, n=cfg.n, N=cfg.N, l=cfg.l, Bgbit=cfg.Bgbit, ks_t=cfg.ks_t,
                             ks_basebit=cfg.ks_basebit, device=device, lib_path=lib_path)
        bk = _T.keygen_bk_torus(32, self.lwe_key, self.tlwe_key, cfg.l, cfg.Bgbit, cfg.bk_stdev, seed, 1000,
                                lib_path=lib_path)
        self.gsw = self.eng.gsw_from_torus(bk)  # tGswToFFTConvert on the GPU
        self.eng.set_bootstrap_key(self.gsw)
        del bk
        ks = _T.keygen_ks32(self.tlwe_key, self.lwe_key, cfg.ks_t, cfg.ks_basebit, cfg.ks_stdev, seed, 100000,
                            lib_path=lib_path)
        self.eng.load_keyswitch_key(ks)
        self.lib_path = lib_path

    def encrypt(self, messages, stdev=2.0 ** -15, stream0=5000):
        return np.stack([_T.lwe_encrypt32(int(m), stdev, self.lwe_key, self.seed, stream0 + i, lib_path=self.lib_path)
                         for i, m in enumerate(messages)])

    def phase(self, ct):
        return _T.lwe_phase32(ct, self.lwe_key, lib_path=self.lib_path)

    def bootstrap(self, mu, x):
        return self.eng.bootstrap(mu, x)

    def close(self):
        self.eng.close()


def max_over_ranks(value, device):
    """MAX all-reduce of a python float (the bench contract's timing rule)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def rank_census(rank, device_index, batch, seconds, device):
    """who took part: (ranks_seen, [{rank, device, batch, seconds}...]).  ranks_seen is a SUM all-reduce of ones on the
    same kind of tensor max_over_ranks uses (the device tensor under RCCL), so a bench line cannot claim ranks
    that did not run; the table is an all_gather of each rank's (device ordinal, samples per step, timed seconds)."""
    import torch
    import torch.distributed as dist
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = torch.tensor([float(rank), float(device_index), float(batch), float(seconds)], dtype=torch.float64, device=device)
    rows = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, mine)
    table = [{"rank": int(r[0].item()), "device": int(r[1].item()), "batch": int(r[2].item()), "seconds": float(r[3].item())}
             for r in rows]
    return int(round(one.item())), table


def gather_rows(local_rows, total, rank, world, device):
    """all-gather ragged row blocks into the full [total][cols] array (tests / verification only)"""
    import torch
    import torch.distributed as dist
    local_rows = np.ascontiguousarray(local_rows)
    if world == 1:
        return local_rows
    cols = local_rows.shape[1]
    maxrows = -(-total // world)
    pad = np.zeros((maxrows, cols), dtype=local_rows.dtype)
    pad[:local_rows.shape[0]] = local_rows
    src = torch.from_numpy(pad).to(device)
    bufs = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(bufs, src)
    out = np.empty((total, cols), dtype=local_rows.dtype)
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        out[lo:hi] = bufs[r].cpu().numpy()[:hi - lo]
    return out
