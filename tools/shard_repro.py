"""is a two-rank sharded solve reproducible run to run?  Runs the (b) leg of tests/shard_worker.py three times per setting
usage: python tools/shard_repro.py"""
import json, os, subprocess, sys, socket, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, json
rank, world, port, refine, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
ROOT = sys.argv[6]
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import numpy as np, torch, torch.distributed as dist, helpers, nnsdp_amd as na
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
def allreduce(a): dist.all_reduce(torch.from_numpy(a))
q = helpers.product_query(helpers.load_problem("W40-D20", 0))
s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), eps_rel=1e-5, max_iters=200000, max_time=200, proj_refine=refine))
s.set_comm_callback(world, rank, allreduce)
sol = s.run()
json.dump(dict(status=sol.termination_status, iters=int(sol.summary["iters"]), pres=float(sol.summary["pres"]), dres=float(sol.summary["dres"]),
               admm=float(sol.summary["objective_admm"]), refine=sol.summary["refine_blocks"]), open(out, "w"))
s.close(); dist.barrier(); dist.destroy_process_group()
'''
wf = os.path.join(tempfile.gettempdir(), "shard_repro_worker.py")
open(wf, "w").write(code)
import itertools
for refine, dbg in ((1, 0), (0, 0)):
    for rep in range(12 if refine else 2):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = str(so.getsockname()[1])
        outs = [os.path.join(tempfile.gettempdir(), f"sr_{refine}_{rep}_{r}.json") for r in range(2)]
        ps = [subprocess.Popen([sys.executable, wf, str(r), "2", port, str(refine), outs[r], ROOT]) for r in range(2)]
        for p in ps:
            p.wait(timeout=400)
        r0, r1 = (json.load(open(o)) for o in outs)
        print("refine", refine, "dbg", dbg, "rep", rep, r0["status"], r0["iters"], repr(r0["pres"]), repr(r0["admm"]), r0["refine"], "ranks equal:", r0["pres"] == r1["pres"], flush=True)
