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
    assert np.allclose(res, serial, rtol=1e-12, atol=0)       # same arithmetic, same answers (threaded BLAS may reorder a sum: 1e-16)
    assert abs(rate - (3 * 150) / 2.0) < 1e-9                 # sum of units / max of times


# ----------------------------------------------------------------------------- clique-sharded mode (one SDP over ranks)
def _product_plan(world):
    """blocks and rank ranges from the PRODUCT's own partition code (nnsdp_shard_plan: the host-only entry point
    nnsdp_solver::set_comm calls), not a re-derivation."""
    import nnsdp_amd as na
    q = helpers.product_query(helpers.load_problem("W10-D10", 0))
    return na.shardPlan(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp()), world)


def _sharded_worker(rank, world, port, out):
    """the HIP library's clique-sharded iteration restated on the oracle state: own cliques are projected
    locally, the consensus sum is all-reduced (gloo here, RCCL on the GPUs), the rest is replicated."""
    import scipy.linalg as sla
    from oracle import admm as oadmm, operator as oop
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q = helpers.oracle_query(helpers.load_problem("W10-D10", 0))
    P = oadmm.ScaledProblem(oop.build_operator(q, "single", normalize=True))
    S = oadmm.AdmmState(P, 0.1, 1.6)
    bn, st = _product_plan(world)
    assert bn == list(S.nk), (bn, S.nk)                      # the product solves the very blocks the oracle state holds
    own = range(st[rank], st[rank + 1])
    for _ in range(40):
        nu = S.nu
        w = np.zeros_like(nu)
        w[:S.ng] = np.maximum(nu[:S.ng], 0.0)
        h = np.zeros(P.pat.NE)
        for k in own:
            n = S.nk[k]
            V = nu[S.offs[k]:S.offs[k + 1]].reshape(n, n)
            Wk = oadmm.project_psd(V)
            w[S.offs[k]:S.offs[k + 1]] = Wk.ravel()
            R = 2.0 * Wk - 0.5 * (V + V.T)
            np.add.at(h, S.trilpos[k], R[S.tril[k]] * S.trilw[k])
        ht = torch.from_numpy(h)
        dist.all_reduce(ht, op=dist.ReduceOp.SUM)             # the one exchange step per iteration
        g = S.Dinv * (P.z0 / S.sigma + ht.numpy())
        p = 2.0 * w[:S.ng] - nu[:S.ng] - P.c
        ww = sla.cho_solve(S.Mfac, S.At @ g - p)
        x = g - S.Dinv * (S.A @ ww)
        new = nu.copy()
        new[:S.ng] = nu[:S.ng] + S.alpha * (p + ww + P.c - w[:S.ng])
        for k in own:
            sl = slice(S.offs[k], S.offs[k + 1])
            new[sl] = nu[sl] + S.alpha * (x[S.G[k]] * S.Wm[k] - w[sl])
        S.nu = new
    # assemble the distributed state: multiplier block is replicated, clique blocks live on their owners
    full = torch.zeros(S.N, dtype=torch.float64)
    for k in own:
        full[S.offs[k]:S.offs[k + 1]] = torch.from_numpy(S.nu[S.offs[k]:S.offs[k + 1]])
    dist.all_reduce(full, op=dist.ReduceOp.SUM)
    full[:S.ng] = torch.from_numpy(S.nu[:S.ng])
    if rank == 0:
        out.put((full.numpy(), st))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_plan_is_contiguous_balanced_and_complete():
    import nnsdp_amd as na
    q = helpers.product_query(helpers.load_problem("W40-D20", 0))
    for mode in (na.SingleDecomp(), na.DoubleDecomp()):
        o = na.AdmmSdpOptions(decomp_mode=mode)
        bn1, st1 = na.shardPlan(q, o, 1)
        assert st1 == [0, len(bn1)] and max(bn1) <= 128
        cost = np.array(bn1, dtype=float) ** 3
        for world in (2, 4, 8, 64):
            bn, st = na.shardPlan(q, o, world)
            assert bn == bn1 and st[0] == 0 and st[-1] == len(bn) and all(a <= b for a, b in zip(st, st[1:]))
            load = [cost[st[r]:st[r + 1]].sum() for r in range(world)]
            if world <= 8:
                # the heaviest rank is the critical path: optimal linear partition, within 20 % of the ideal share here
                assert min(load) > 0 and max(load) <= 1.2 * cost.sum() / world
    with pytest.raises(na._lib.NnsdpError):
        na.shardPlan(q, na.AdmmSdpOptions(), 0)


def test_clique_sharded_iteration_matches_serial_gloo():
    from oracle import admm as oadmm, operator as oop
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    nu_sharded, st = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert st[0] == 0 and 0 < st[1] < st[2] == 9              # 9 cliques of W10-D10 split over 2 ranks
    assert st == _product_plan(2)[1]
    q = helpers.oracle_query(helpers.load_problem("W10-D10", 0))
    S = oadmm.AdmmState(oadmm.ScaledProblem(oop.build_operator(q, "single", normalize=True)), 0.1, 1.6)
    for _ in range(40):
        S.step()
    assert np.abs(nu_sharded - S.nu).max() <= 1e-11 * max(1.0, np.abs(S.nu).max())


# ----------------------------------------------------------------------------- f4: (network, spec) pairs over ranks
def _oracle_safety_solver(query, opts):
    """stands in for runQuery on ranks without a GPU: the oracle ADMM on the same query (checker side of the tests)."""
    from oracle import admm as oadmm, nnet_io, operator as oop, qc as oqc
    net = nnet_io.FeedFwdNet(xdims=list(query.ffnet.xdims), Ms=query.ffnet.Ms)
    qs = query.qc_activs[1]
    qo = oqc.make_safety_query(net, np.asarray(query.qc_input.x1min), np.asarray(query.qc_input.x1max), qs.beta, query.qc_safety.S)
    r = oadmm.admm_solve(oop.build_operator(qo, "single", normalize=True), oadmm.AdmmOptions(max_iters=opts.max_iters))
    import nnsdp_amd as na
    ok = r.pres <= 1e-4 and r.dres <= 1e-4
    return na.QuerySolution(float(r.objective), {}, "OPTIMAL" if ok else "ITERATION_LIMIT", 1.0, 0.1, 0.9,
                            {"lambda_max": 1e-9 if ok else 1.0})


def _pairs():
    import nnsdp_amd as na
    from oracle import nnet_io
    spec = os.path.join(helpers.GOLDEN, "vnnlib", "prop_or_inputs.vnnlib")
    nets = []
    for seed, xd in ((1, [2, 6, 6, 2]), (2, [2, 8, 8, 8, 2]), (3, [2, 6, 6, 2]), (4, [2, 8, 8, 8, 2])):
        n = nnet_io.random_net(xd, seed=seed)
        nets.append((f"net{seed}", na.FeedFwdNet(xdims=list(n.xdims), Ms=n.Ms), "prop_or_inputs", spec))
    return nets


def _pair_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nnsdp_amd as na
    from nnsdp_amd import vnnlib as vl
    pairs = _pairs()
    mine = vl.shardPairs(pairs, 0, world, rank)
    rows, _ = vl.verifyPairs(mine, 0, na.AdmmSdpOptions(max_iters=1500), solve=_oracle_safety_solver)
    gathered = [None] * world
    dist.all_gather_object(gathered, [(r[0], r[2], r[3], r[4]) for r in rows])      # result table only
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_vnnlib_pairs_sharded_over_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_pair_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import nnsdp_amd as na
    from nnsdp_amd import vnnlib as vl
    serial, _ = vl.verifyPairs(_pairs(), 0, na.AdmmSdpOptions(max_iters=1500), solve=_oracle_safety_solver)
    merged = sorted(x for part in gathered for x in part)
    assert merged == sorted((r[0], r[2], r[3], r[4]) for r in serial)
    assert all(len(part) == 2 for part in gathered)                                   # one heavy + one light pair per rank
