"""The product's clique-sharded mode with TWO ranks on the GPU ("no process group larger than 1 has ever driven
nnsdp_solver_set_comm").  RCCL refuses two ranks on one device and the test box has one, so the two processes exchange through
gloo via nnsdp_solver_set_comm_callback; partition, owned-source gather, the kernels, the staging of the exchange and the
collective control decisions are the code RCCL mode runs (the RCCL call itself: test_clique_sharded_mode_single_rank_rccl)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers
import nnsdp_amd as na

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("name,beta", [("W10-D10", 0), ("W40-D20", 0)])
def test_two_rank_sharded_solve_matches_the_serial_solve(name, beta, tmp_path):
    port = _free_port()
    worker = os.path.join(helpers.ROOT, "tests", "shard_worker.py")
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, name, str(beta), outs[r]]) for r in range(2)]
    # the serial solves run in this process meanwhile (3 processes on the card)
    q = helpers.product_query(helpers.load_problem(name, beta))
    s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), proj_refine=0))
    import time
    t0 = time.perf_counter()
    s.iterate(300)
    us_serial = 1e6 * (time.perf_counter() - t0) / 300
    ref300 = s.residuals()
    s.close()
    try:
        ref = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), eps_rel=1e-5, max_iters=200000, max_time=200))
        refc = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), eps_rel=1e-7, cert_tol=1e-3, max_iters=200000, max_time=200))
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:            # a failure above must not leave rank processes behind on the card
            if p.poll() is None:
                p.kill()
                p.wait()
    r0, r1 = (json.load(open(o)) for o in outs)
    os.makedirs(os.path.join(helpers.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(helpers.ROOT, "gpurun_out", f"shard_workers_{name}.json"), "w") as fh:      # post-mortem material
        json.dump({"rank0": {k: v for k, v in r0.items() if k != "solve" and k != "cert"}, "solve0": {k: v for k, v in r0["solve"].items() if k != "gamma"},
                   "solve1": {k: v for k, v in r1["solve"].items() if k != "gamma"}, "cert0": {k: v for k, v in r0["cert"].items() if k != "gamma"},
                   "ref": [ref.termination_status, ref.summary["iters"], ref.summary["pres"], ref.summary["dres"]],
                   "refc": [refc.termination_status, refc.summary["iters"]]}, fh, indent=1)
    # diagnostics of the replicated state, kept beside the run (gpurun_out/ travels back from the GPU box)
    diag = {k: [r0[k], r1[k]] for k in ("minv_digest", "mult300_digest", "mult301_digest")}
    # control-flow cost of the sharded iteration (two ranks on ONE card, gloo through host memory: an upper bound, not a scaling figure)
    diag["us_per_iteration"] = {"serial_one_process": us_serial, "sharded_two_ranks_callback_transport": [r0["us_per_iter_sharded_callback"], r1["us_per_iter_sharded_callback"]]}
    print("us / iteration: serial", round(us_serial, 1), "sharded over two ranks on one card (callback transport)", [round(r["us_per_iter_sharded_callback"], 1) for r in (r0, r1)])
    print("replicated-state digests equal on both ranks:", {k: v[0] == v[1] for k, v in diag.items() if k.endswith("digest")})
    os.makedirs(os.path.join(helpers.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(helpers.ROOT, "gpurun_out", f"shard_diag_{name}.json"), "w") as fh:
        json.dump(diag, fh, indent=1)
    assert diag["mult301_digest"][0] == diag["mult301_digest"][1]   # after a check iteration the multiplier block IS rank 0's, bit for bit
    # (a) 300 plain iterations: the same iterate as the serial run (the consensus sum is associated differently: 1e-9)
    for r in (r0, r1):
        assert np.allclose(r["after_300"], ref300, rtol=1e-8, atol=1e-12), (r["after_300"], ref300)
    assert r0["after_300"] == r1["after_300"]                     # the control numbers are bit-identical on both ranks
    # (b) whole solve: both ranks take every decision identically and land on the serial certificate
    a, b = r0["solve"], r1["solve"]
    assert a["status"] == b["status"] == ref.termination_status == "OPTIMAL"
    assert a["iters"] == b["iters"]
    assert np.array_equal(np.array(a["gamma"]), np.array(b["gamma"]))
    assert abs(a["admm"] - ref.summary["objective_admm"]) <= 1e-4 * abs(ref.summary["objective_admm"])
    assert abs(a["rho"] - ref.objective_value) <= 1e-3 * abs(ref.objective_value)
    assert a["lambda_max"] <= 1e-6 and min(a["gamma"]) >= 0.0
    assert r0["blocks_owned"][1] == r1["blocks_owned"][0] and r0["blocks_owned"][0] == 0
    assert r0["allreduce_calls"] == r1["allreduce_calls"] >= a["iters"]
    # (c) the certified-gap rule under sharding: one decision (rank 0 polishes, the flag is all-reduced), one certificate
    ca, cb = r0["cert"], r1["cert"]
    assert ca["status"] == cb["status"] == refc.termination_status == "OPTIMAL"
    assert ca["iters"] == cb["iters"]
    assert ca["iters"] <= 2 * refc.summary["iters"] + 500          # (the rule fires at the serial solve's check point or one of the next few)
    assert np.array_equal(np.array(ca["gamma"]), np.array(cb["gamma"]))
    assert max(ca["pres"], ca["dres"]) > 1e-7                       # it was the certified-gap rule that stopped the solve
    assert abs(ca["rho"] - refc.objective_value) <= 1e-3 * abs(refc.objective_value)
    assert ca["lambda_max"] <= 1e-6 and min(ca["gamma"]) >= 0.0
    # (d) with the refinement stage on (the default): the control numbers and the resynchronised multiplier block are bit-identical
    # on both ranks (the stage's accept / reject decisions are taken by each block's owner alone; nothing replicated depends on them)
    assert r0["refine_after_300"] == r1["refine_after_300"]
    assert r0["refine_mult301_digest"] == r1["refine_mult301_digest"]


@pytest.mark.parametrize("name", ["W40-D40", "acas-shape"])
def test_two_rank_sharded_baseline_configs_4_and_5(name, tmp_path):
    """BASELINE config 4's network (bench/rand W=40 D=40: 39 blocks over two ranks) and config 5's shape (5-50x6-5 in the reference's
    Single cliques, 106 + 4 x 151: the packed variant / the tile-parallel pipeline on each rank's own blocks) through the product's
    sharded code with two processes on the card: 200 plain iterations equal to the serial iterate, the resynchronised multiplier
    block bit-identical on both ranks, and a capped solve in the default configuration returning ONE certificate."""
    port = _free_port()
    worker = os.path.join(helpers.ROOT, "tests", "shard_worker.py")
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, name + ":light", "0", outs[r]]) for r in range(2)]
    q = helpers.acas_shaped_query() if name == "acas-shape" else helpers.product_query(helpers.load_problem(name, 0))
    try:
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), proj_refine=0))
        s.iterate(200)
        ref200 = s.residuals()
        s.close()
        for p in procs:
            assert p.wait(timeout=900) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    r0, r1 = (json.load(open(o)) for o in outs)
    # (the consensus sum is associated differently in the two modes; on the 151-wide blocks the inexact sweeps of the first hundred
    # iterations - stopped at the adaptive tolerance, 1e-4 this early - turn that 1e-16 into a difference at their own level)
    # W40-D40: the serial run's M^-1 comes from this process's own rocSOLVER call, the ranks' from rank 0's - under three-process
    # contention those differ in the last bits one run in three (profiles/r03_contention_repro_minv_digest.log),
    # and 200 iterations of this collapsed network turn that into 4e-7; the two RANKS stay bit-identical (asserted below)
    rtol = 2e-3 if name == "acas-shape" else 1e-5
    for r in (r0, r1):
        assert np.allclose(r["after_200"], ref200, rtol=rtol, atol=1e-12), (r["after_200"], ref200)
    assert r0["after_200"] == r1["after_200"] and r0["mult201_digest"] == r1["mult201_digest"]
    a, b = r0["capped"], r1["capped"]
    assert a["status"] == b["status"] and a["iters"] == b["iters"]
    assert np.array_equal(np.array(a["gamma"]), np.array(b["gamma"]))
    assert a["lambda_max"] <= 1e-6 and min(a["gamma"]) >= 0.0          # (a certificate of the capped iterate: feasible, if not yet tight)
    assert r0["blocks_owned"][0] == 0 and r0["blocks_owned"][1] == r1["blocks_owned"][0] and r1["blocks_owned"][1] == len(r0["blocks"])
    if name == "acas-shape":
        assert len(r0["blocks"]) == 5 and max(r0["blocks"]) == 151 and min(r0["blocks"]) > 80      # (106 + 4 x 151 less the coordinates the normalisation eliminates)


@pytest.mark.parametrize("name", ["W40-D20", "W40-D40"])
def test_two_rank_sharded_solve_over_the_device_side_transport(name, tmp_path):
    """the sharded mode's per-iteration exchange as the library's own kernels over hipIpc-mapped peer buffers (nnsdp_solver_set_comm_ipc):
    two processes on the card, no host on the data path (the iterations between checks replay a hipGraph that contains the exchange),
    the same iterate as the serial run, identical bits on both ranks, a whole solve landing on the serial certificate; the cost of a
    sharded iteration beside the serial one is recorded (gpurun_out/shard_ipc_*.json)."""
    port = _free_port()
    worker = os.path.join(helpers.ROOT, "tests", "shard_worker.py")
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, name + ":ipc", "0", outs[r]]) for r in range(2)]
    q = helpers.product_query(helpers.load_problem(name, 0))
    try:
        import time
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), proj_refine=0))
        s.iterate(100)
        t0 = time.perf_counter()
        s.iterate(400)
        us_serial = 1e6 * (time.perf_counter() - t0) / 400
        ref500 = s.residuals()
        s.close()
        ref = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), eps_rel=1e-5, max_iters=200000, max_time=200))
        for p in procs:
            assert p.wait(timeout=900) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    r0, r1 = (json.load(open(o)) for o in outs)
    os.makedirs(os.path.join(helpers.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(helpers.ROOT, "gpurun_out", f"shard_ipc_{name}.json"), "w") as fh:
        json.dump({"us_per_iteration": {"serial_one_process_beside_two_ranks": us_serial, "sharded_two_ranks_ipc": [r0["us_per_iter_sharded_ipc"], r1["us_per_iter_sharded_ipc"]]},
                   "graph_launches": [r0["graph_launches"], r1["graph_launches"]], "host_allreduce_calls_in_the_solve": r0["host_allreduce_calls"],
                   "solve": {k: v for k, v in r0["solve"].items() if k != "gamma"}, "serial_solve": [ref.termination_status, ref.summary["iters"], ref.solve_time]}, fh, indent=1)
    print("us / iteration: serial", round(us_serial, 1), "sharded over two ranks on one card, hipIpc transport", [round(r["us_per_iter_sharded_ipc"], 1) for r in (r0, r1)])
    # sharded vs serial differ in the ORDER the consensus sum is taken in (per-rank partial sums, then rank order) and, under the
    # three-process contention of this test, in the rounding of the serial process's own M^-1 (section 6 of DESIGN.md); 500 early
    # iterations amplify that to 1e-9 on W40-D20 and to 2e-6 .. 1e-5 on W40-D40's 39 blocks (observed over the round's runs).  The
    # strong statements are the next line's (both ranks: identical bits) and the whole solve's below
    rtol = 1e-4 if name == "W40-D40" else 1e-7
    for r in (r0, r1):
        assert np.allclose(r["after_500"], ref500, rtol=rtol, atol=1e-12), (r["after_500"], ref500)
    assert r0["after_500"] == r1["after_500"] and r0["mult501_digest"] == r1["mult501_digest"]
    assert r0["graph_launches"] > 0 and r1["graph_launches"] > 0            # the exchange ran inside replayed graphs
    assert r0["ipc_transport"] == 2.0 and r1["ipc_transport"] == 2.0        # exchange buffers in fine-grained memory (coherent across devices while kernels run)
    a, b = r0["solve"], r1["solve"]
    assert a["status"] == b["status"] == ref.termination_status == "OPTIMAL"
    assert a["iters"] == b["iters"]
    assert np.array_equal(np.array(a["gamma"]), np.array(b["gamma"]))
    assert abs(a["rho"] - ref.objective_value) <= 1e-3 * abs(ref.objective_value)
    assert a["lambda_max"] <= 1e-6 and min(a["gamma"]) >= 0.0
    assert r0["host_allreduce_calls"] < a["iters"] // 10                    # the host collective only at set-up and check iterations
