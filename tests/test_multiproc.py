"""world_size-2 CPU (gloo) test of the multi-GPU path's host logic (bench.py / shard.py):
contiguous batch sharding, key replication by seed, max-over-ranks timing, gathered outputs
equal the single-rank result.  Compute runs on the tests/emu build (no GPU here)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, emu_lib, q, keys="seed", N=1024):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), TFHE_EMU_DEVICES="8")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    T = importlib.import_module("experimental-tfhe_amd")
    cfg = shard.GateConfig(N=N, n=4, l=2, Bgbit=10, ks_t=8, ks_basebit=2)
    total = 7  # ragged: 4 + 3
    lo, hi = shard.shard_range(total, rank, world)
    # keys replicated from the seed, or built by rank 0 alone and broadcast as device-layout bytes; rank r on emulated device 3 r
    job = shard.GateJob(cfg, seed=0x5446484500000001, device=3 * rank, lib_path=emu_lib, keys=keys, tensor_device=torch.device("cpu"))
    if keys == "broadcast":
        assert job.key_bytes_received == 4 * 4 * 2 * N * 8 + N * 8 * 4 * 5 * 4
        assert (job.tlwe_key is None) == (rank != 0)  # only rank 0 ever held the host keys
    x_all = shard.synthetic_samples(cfg, total, seed=77)
    out = job.bootstrap(1 << 29, x_all[lo:hi])
    t_local = 0.25 * (rank + 1)
    t_max = shard.max_over_ranks(t_local, device="cpu")
    gathered = shard.gather_rows(out, total, rank, world, device="cpu")
    if rank == 0:
        single = job.bootstrap(1 << 29, x_all)
        q.put((t_max, bool(np.array_equal(gathered, single)), (lo, hi)))
    job.close()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("keys,N", [("seed", 1024), ("broadcast", 1024), ("broadcast", 512)])
def test_two_rank_sharding(emu_lib, keys, N):
    """keys="broadcast": rank 1 never generates a key -- it bootstraps with the bytes rank 0 broadcast, on another
    (emulated) device, and the gathered outputs still equal the single-rank result.  N = 512: the same with the generic kernels'
    key layout as the broadcast bytes"""
    import torch.multiprocessing as mp  # imported lazily: collecting -m gpu tests must stay light
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if keys == "broadcast" else 0) + (13 if N != 1024 else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu_lib, q, keys, N)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    t_max, same, rng0 = q.get(timeout=10)
    assert t_max == 0.5
    assert same
    assert rng0 == (0, 4)


def test_shard_ranges_cover_batch():
    shard = importlib.import_module("experimental-tfhe_amd.shard")
    for total in (0, 1, 7, 4096, 2 ** 20):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
