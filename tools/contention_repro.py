"""are plain (unsharded) solves reproducible when several processes share the card?  usage: python tools/contention_repro.py [nproc=3] [reps=6]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, json
refine, out, ROOT, graph = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
if graph == "0": os.environ["NNSDP_NO_GRAPH"] = "1"
if graph == "rr": os.environ["NNSDP_PROJ_ALG"] = "0"
sys.path.insert(0, os.path.join(ROOT, "nn-sdp_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers, nnsdp_amd as na
q = helpers.product_query(helpers.load_problem("W40-D20", 0))
import hashlib
sv = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), eps_rel=1e-5, max_iters=200000, max_time=200, proj_refine=refine))
dig = hashlib.sha256(sv.apply_minv(np.cos(np.arange(sv.cp.ngamma) * 0.37))[0].tobytes()).hexdigest()[:12]
sol = sv.run()
sv.close()
json.dump(dict(minv=dig, status=sol.termination_status, iters=int(sol.summary["iters"]), pres=float(sol.summary["pres"]), admm=float(sol.summary["objective_admm"]),
               refine=sol.summary["refine_blocks"]), open(out, "w"))
'''
wf = os.path.join(tempfile.gettempdir(), "contention_worker.py")
open(wf, "w").write(code)
nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
settings = ((1, "0"), (1, "1"), (0, "0"))
if len(sys.argv) > 3 and sys.argv[3] == "alg":
    settings = ((0, "1"), (0, "1"))
for refine, graph in settings:
    seen = {}
    for rep in range(reps):
        outs = [os.path.join(tempfile.gettempdir(), f"cr_{refine}_{rep}_{r}.json") for r in range(nproc)]
        ps = [subprocess.Popen([sys.executable, wf, str(refine), outs[r], ROOT, graph]) for r in range(nproc)]
        for p in ps:
            p.wait(timeout=400)
        for o in outs:
            r = json.load(open(o))
            key = (r["minv"], r["status"], r["iters"], repr(r["pres"]), repr(r["admm"]), tuple(r["refine"]))
            seen[key] = seen.get(key, 0) + 1
    print(f"refine {refine} graph {graph}: {nproc} concurrent processes x {reps} repetitions -> distinct results:", flush=True)
    for k, v in seen.items():
        print("   ", v, "x", k, flush=True)
