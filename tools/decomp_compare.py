import sys, os, time
sys.path.insert(0, "nn-sdp_amd"); sys.path.insert(0, "tests")
import numpy as np, helpers, nnsdp_amd as na
for name, beta in (("W40-D20", 0), ("W40-D20", 2), ("W10-D20", 0), ("W20-D30", 0)):
    try:
        q = helpers.product_query(helpers.load_problem(name, beta))
    except Exception as e:
        d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
        xd = [int(v) for v in d["xdims"]]
        net = na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])
        q, P, yc = na.ellipsoidQuery(net, [0.5, 0.5], [1.5, 1.5], beta)
    for mode in (na.SingleDecomp(), na.DoubleDecomp(), na.PathDecomp()):
        for kw in (dict(eps_rel=1e-6), dict(eps_rel=1e-6, cert_tol=1e-3)):
            s = na.runQuery(q, na.AdmmSdpOptions(decomp_mode=mode, max_iters=400000, max_time=60, **kw))
            print(name, beta, type(mode).__name__, kw, s.termination_status, "iters", s.summary["iters"], f"solve {s.solve_time:.2f}s rho {s.objective_value:.8g} blocks {s.summary['n_cliques']} max {s.summary['max_clique']}", flush=True)
