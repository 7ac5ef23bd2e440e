"""Host-side job logic for batches of gate bootstraps on one or several GPUs.

The reference's only parallel construct is an independent-item loop
(parallel/src/test_parallel_multiplications.cpp:62 `#pragma omp parallel for`): ciphertexts
never interact.  The multi-GPU mapping is therefore: one process per GPU, the batch cut into
contiguous slices, keys REPLICATED, no collective on the data path.  Two ways to replicate:
  keys="seed"       every rank regenerates the same synthetic keys from the same seed (no communication at all;
                    only possible because the bench's keys are synthetic)
  keys="broadcast"  rank 0 alone holds host keys (as a client's key would arrive), uploads them, and ONE broadcast per key
                    (torch.distributed: RCCL on GPUs) hands the other ranks the bytes of the DEVICE layout -- SURVEY 8(e):
                    "replicate bkFFT + KS key on every GPU at setup (one hipMemcpy per device or one RCCL broadcast)"
torch.distributed is otherwise used only for the timing barrier / max-over-ranks and for gathering results in tests.
(One process driving several GPUs is the other form: tfhe_amd_pool in include/tfhe_amd.h, `Pool` in the binding.)
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


def host_keys(cfg, seed, lib_path=None):
    """(lwe_key, tlwe_key, bk in coefficient form [n][2l][2][N] int32, ks [N][t][base][n+1] int32): the synthetic key set"""
    lwe_key = _T.keygen_binary(cfg.n, seed, 1, lib_path=lib_path)
    tlwe_key = _T.keygen_binary(cfg.N, seed, 2, lib_path=lib_path)
    bk = _T.keygen_bk_torus(32, lwe_key, tlwe_key, cfg.l, cfg.Bgbit, cfg.bk_stdev, seed, 1000, lib_path=lib_path)
    ks = _T.keygen_ks32(tlwe_key, lwe_key, cfg.ks_t, cfg.ks_basebit, cfg.ks_stdev, seed, 100000, lib_path=lib_path)
    return lwe_key, tlwe_key, bk, ks


class GateJob:
    """keys + engine of one rank.  keys="seed": secret keys, bootstrapping key and key-switch key are functions of
    (cfg, seed) only, so every rank builds identical replicas without communication.  keys="broadcast": only rank 0
    builds and uploads them; the others receive the device-layout bytes (see the module docstring); `tensor_device` is the
    torch device collectives run on (cuda:<d> under nccl = RCCL, cpu under gloo)."""

    def __init__(self, cfg, seed, device=0, lib_path=None, keys="seed", tensor_device=None, keep_host_keys=False):
        self.cfg, self.seed = cfg, seed
        self.lwe_key = _T.keygen_binary(cfg.n, seed, 1, lib_path=lib_path)  # the rank's own decrypt checks (2.5 KB: never broadcast)
        self.tlwe_key = None
        self.eng = _T.Engine(torus_bits=32, n=cfg.n, N=cfg.N, l=cfg.l, Bgbit=cfg.Bgbit, ks_t=cfg.ks_t,
                             ks_basebit=cfg.ks_basebit, device=device, lib_path=lib_path)
        self.lib_path, self.keys, self.key_bytes_received = lib_path, keys, 0
        self.bk_host = self.ks_host = None
        rank0 = True
        if keys == "broadcast":
            import torch.distributed as dist
            rank0 = dist.get_rank() == 0
        if rank0:
            _, self.tlwe_key, bk, ks = host_keys(cfg, seed, lib_path)
            self.gsw = self.eng.gsw_from_torus(bk)  # tGswToFFTConvert on the GPU
            self.eng.load_keyswitch_key(ks)
            if keep_host_keys:
                self.bk_host, self.ks_host = bk, ks
            del bk, ks
        if keys == "broadcast":
            self.gsw = self._broadcast_keys(rank0, tensor_device)
        self.eng.set_bootstrap_key(self.gsw)

    def _broadcast_keys(self, rank0, tensor_device):
        """one broadcast per key, of the bytes of its DEVICE layout: rank 0 exports its resident copy into a torch buffer on
        the collective's device, the others import what arrives (a device-to-device copy under RCCL)"""
        import torch
        import torch.distributed as dist
        cfg, eng = self.cfg, self.eng
        bk_t = torch.empty(eng.gsw_packed_bytes(cfg.n), dtype=torch.uint8, device=tensor_device)
        ks_t = torch.empty(eng.keyswitch_key_bytes(), dtype=torch.uint8, device=tensor_device)
        if rank0:
            eng.gsw_export_packed(self.gsw, bk_t.data_ptr())
            eng.keyswitch_key_export(ks_t.data_ptr())
        dist.broadcast(bk_t, src=0)
        dist.broadcast(ks_t, src=0)
        if bk_t.is_cuda:
            torch.cuda.synchronize()
        self.key_bytes_received = bk_t.numel() + ks_t.numel()
        if rank0:
            return self.gsw
        gsw = eng.gsw_from_packed(bk_t.data_ptr(), cfg.n)
        eng.load_keyswitch_key_d(ks_t.data_ptr())
        return gsw

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


def pci_to_number(bus_id):
    """'dddd:bb:dd.f' -> domain << 16 | bus << 8 | device << 3 | function (exact in a float64); -1 when unknown"""
    try:
        dom, bus, rest = bus_id.split(":")
        dev, fn = rest.split(".")
        return (int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(dev, 16) << 3) | int(fn, 16)
    except (AttributeError, ValueError):
        return -1


def number_to_pci(v):
    v = int(v)
    return None if v < 0 else "%04x:%02x:%02x.%x" % (v >> 16, (v >> 8) & 0xFF, (v >> 3) & 0x1F, v & 7)


def rank_census(rank, device_index, batch, seconds, device, pci_bus_id=None):
    """who took part: (ranks_seen, [{rank, device, batch, seconds, pci}...]).  ranks_seen is a SUM all-reduce of ones on the
    same kind of tensor max_over_ranks uses (the device tensor under RCCL), so a bench line cannot claim ranks
    that did not run; the table is an all_gather of each rank's (device ordinal, samples per step, timed seconds, PCI bus
    id of its GPU).  The ordinal is per process -- under a launcher that shows every rank one device they are all 0 -- the
    bus id is what proves N distinct GPUs."""
    import torch
    import torch.distributed as dist
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    mine = torch.tensor([float(rank), float(device_index), float(batch), float(seconds), float(pci_to_number(pci_bus_id))],
                        dtype=torch.float64, device=device)
    rows = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, mine)
    table = [{"rank": int(r[0].item()), "device": int(r[1].item()), "batch": int(r[2].item()), "seconds": float(r[3].item()),
              "pci": number_to_pci(r[4].item())} for r in rows]
    return int(round(one.item())), table


def distinct_devices(table):
    """number of different GPUs in a census table (by PCI bus id; ranks whose id is unknown count as one device each)"""
    known = {r["pci"] for r in table if r.get("pci")}
    return len(known) + sum(1 for r in table if not r.get("pci"))


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
