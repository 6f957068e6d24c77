"""N > 1 host logic on CPU: world_size-2 gloo.  The multi-GPU mode shards INDEPENDENT SDPs over
ranks (no data-path collective); this test runs the same sharding + aggregation code bench.py uses,
with the oracle ADMM standing in for the GPU solver, and checks the sharded job equals the serial one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from nnsdp_amd import parallel


def test_shard_units_partitions():
    for n in (0, 1, 7, 8, 39):
        for world in (1, 2, 3, 8):
            parts = [parallel.shard_units(n, world, r) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_units(4, 2, 2)
    bins = parallel.bin_pack_by_cost([121 ** 3] * 4 + [83 ** 3] + [151 ** 3] * 3, 3)
    assert sorted(sum(bins, [])) == list(range(8))
    loads = [sum(([121 ** 3] * 4 + [83 ** 3] + [151 ** 3] * 3)[i] for i in b) for b in bins]
    assert max(loads) / min(loads) < 1.6


UNITS = [("W10-D5", 0), ("W10-D5", 3), ("W10-D10", 0)]


def _solve_unit(u, iters=150):
    from oracle import admm as oadmm, operator as oop
    q = helpers.oracle_query(helpers.load_problem(*UNITS[u]))
    r = oadmm.admm_solve(oop.build_operator(q, "single", normalize=True), oadmm.AdmmOptions(max_iters=iters))
    return r.objective


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = parallel.shard_units(len(UNITS), world, rank)
    res = torch.zeros(len(UNITS), dtype=torch.float64)
    for u in mine:
        res[u] = _solve_unit(u)
    dist.all_reduce(res, op=dist.ReduceOp.SUM)               # result gathering only, not on the data path
    rate = parallel.aggregate_rate(len(mine) * 150, 1.0 + rank, dist)
    if rank == 0:
        out.put((res.tolist(), rate))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_serial():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res, rate = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    serial = [_solve_unit(u) for u in range(len(UNITS))]
    assert np.allclose(res, serial, rtol=0, atol=0)           # same arithmetic, same answers
    assert abs(rate - (3 * 150) / 2.0) < 1e-9                 # sum of units / max of times
