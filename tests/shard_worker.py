"""Worker of tests/test_sharded_two_ranks.py: one rank of a clique-sharded solve on the GPU, the per-iteration exchange carried
by gloo through nnsdp_solver_set_comm_callback (RCCL refuses two ranks on one device, and the test box has one).
usage: python tests/shard_worker.py <rank> <world> <port> <fixture> <beta> <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, name, beta, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist
    import helpers
    import nnsdp_amd as na
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    calls = [0]

    def allreduce(a):
        calls[0] += 1
        dist.all_reduce(torch.from_numpy(a))      # in place on the library's host buffer

    ipc = name.endswith(":ipc")              # the device-side transport (hipIpc-mapped peer buffers) instead of the host callback
    if ipc:
        name = name.split(":")[0]
        import hashlib, time
        q = helpers.acas_shaped_query() if name == "acas-shape" else helpers.product_query(helpers.load_problem(name, beta))
        res = {"rank": rank}
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), proj_refine=0))
        s.set_comm_ipc(world, rank, allreduce)
        s.iterate(100)
        dist.barrier()
        t0 = time.perf_counter()
        s.iterate(400)
        res["us_per_iter_sharded_ipc"] = 1e6 * (time.perf_counter() - t0) / 400
        res["graph_launches"] = s.info(0)
        res["ipc_transport"] = s.info(6)
        res["after_500"] = list(s.residuals())
        res["mult501_digest"] = hashlib.sha256(s.raw_multipliers().tobytes()).hexdigest()
        s.close()
        calls_setup = calls[0]
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), eps_rel=1e-5, max_iters=200000, max_time=200))
        s.set_comm_ipc(world, rank, allreduce)
        sol = s.run()
        res["solve"] = dict(status=sol.termination_status, iters=int(sol.summary["iters"]), rho=float(sol.objective_value),
                            admm=float(sol.summary["objective_admm"]), lambda_max=float(sol.summary["lambda_max"]), solve_s=float(sol.solve_time),
                            gamma=np.concatenate([sol.values[k] for k in ("γin", "γout", "γac1", "γac2")]).tolist())
        s.close()
        res["host_allreduce_calls"] = calls[0] - calls_setup
        dist.barrier()
        dist.destroy_process_group()
        with open(out, "w") as fh:
            json.dump(res, fh)
        return
    light = name.endswith(":light")          # BASELINE configs 4 / 5 (W40-D40, the ACAS shape): the plain-iteration leg and a capped solve
    name = name.split(":")[0]
    q = helpers.acas_shaped_query() if name == "acas-shape" else helpers.product_query(helpers.load_problem(name, beta))
    res = {"rank": rank}
    if light:
        import hashlib
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), proj_refine=0))
        s.set_comm_callback(world, rank, allreduce)
        s.iterate(200)
        res["after_200"] = list(s.residuals())
        res["mult201_digest"] = hashlib.sha256(s.raw_multipliers().tobytes()).hexdigest()
        s.close()
        # the default configuration (stage on; blocks above 96: the tile-parallel pipeline once the stage carries the visits) to an
        # iteration cap: every decision collective, one certificate
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), eps_rel=1e-6, max_iters=2500, max_time=200))
        s.set_comm_callback(world, rank, allreduce)
        sol = s.run()
        res["capped"] = dict(status=sol.termination_status, iters=int(sol.summary["iters"]), rho=float(sol.objective_value),
                             lambda_max=float(sol.summary["lambda_max"]), pres=float(sol.summary["pres"]),
                             gamma=np.concatenate([sol.values[k] for k in ("γin", "γout", "γac1", "γac2")]).tolist())
        s.close()
        bn, st = na.shardPlan(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp()), world)
        res["blocks"] = [int(v) for v in bn]
        res["blocks_owned"] = [int(st[rank]), int(st[rank + 1])]
        dist.barrier()
        dist.destroy_process_group()
        with open(out, "w") as fh:
            json.dump(res, fh)
        return
    # (a) exactly 300 plain iterations, then one check iteration
    # (proj_refine=0 here and in the parent's serial run: the iterates are compared to 1e-8, and the refinement stage's accept /
    # reject thresholds would amplify the 1e-9 difference of the two summation orders; legs (b), (c) run the default)
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), proj_refine=0))
    s.set_comm_callback(world, rank, allreduce)
    import time
    t0 = time.perf_counter()
    s.iterate(300)
    res["us_per_iter_sharded_callback"] = 1e6 * (time.perf_counter() - t0) / 300
    # diagnostics of the REPLICATED state: the Woodbury core applied to a fixed vector and the multiplier block after
    # 300 plain iterations (no check iteration yet, so no resynchronisation has happened), as hex digests the parent compares
    import hashlib
    qfix = np.cos(np.arange(s.cp.ngamma) * 0.37)
    res["minv_digest"] = hashlib.sha256(s.apply_minv(qfix)[0].tobytes()).hexdigest()
    res["mult300_digest"] = hashlib.sha256(s.raw_multipliers().tobytes()).hexdigest()
    res["after_300"] = list(s.residuals())
    res["mult301_digest"] = hashlib.sha256(s.raw_multipliers().tobytes()).hexdigest()    # after a check iteration: resynchronised
    s.close()
    # (b) a whole solve: every stopping / penalty / tolerance decision is collective
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), eps_rel=1e-5, max_iters=200000, max_time=200))
    s.set_comm_callback(world, rank, allreduce)
    sol = s.run()
    res["solve"] = dict(status=sol.termination_status, iters=int(sol.summary["iters"]), rho=float(sol.objective_value),
                        admm=float(sol.summary["objective_admm"]), pres=float(sol.summary["pres"]), dres=float(sol.summary["dres"]),
                        lambda_max=float(sol.summary["lambda_max"]),
                        gamma=np.concatenate([sol.values[k] for k in ("γin", "γout", "γac1", "γac2")]).tolist())
    s.close()
    res["allreduce_calls"] = calls[0]
    # (c) the certified-gap stopping rule (cert_tol) in sharded mode: rank 0 polishes, the stop flag is all-reduced
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), eps_rel=1e-7, cert_tol=1e-3, max_iters=200000, max_time=200))
    s.set_comm_callback(world, rank, allreduce)
    sol = s.run()
    res["cert"] = dict(status=sol.termination_status, iters=int(sol.summary["iters"]), rho=float(sol.objective_value),
                       pres=float(sol.summary["pres"]), dres=float(sol.summary["dres"]), lambda_max=float(sol.summary["lambda_max"]),
                       gamma=np.concatenate([sol.values[k] for k in ("γin", "γout", "γac1", "γac2")]).tolist())
    s.close()
    # (d) the DEFAULT configuration (refinement stage on) at the level of the replicated state: 300 plain iterations + one check
    # iteration; the multiplier block is replicated, so after the check iteration's resynchronisation it is rank 0's bit for bit
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp()))
    s.set_comm_callback(world, rank, allreduce)
    s.iterate(300)
    res["refine_after_300"] = list(s.residuals())
    res["refine_mult301_digest"] = hashlib.sha256(s.raw_multipliers().tobytes()).hexdigest()
    s.close()
    bn, st = na.shardPlan(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp()), world)
    res["blocks_owned"] = [int(st[rank]), int(st[rank + 1])]
    dist.barrier()
    dist.destroy_process_group()
    with open(out, "w") as fh:
        json.dump(res, fh)


if __name__ == "__main__":
    main()
