#!/usr/bin/env python3
"""Benchmark of the hot path: ADMM iterations/s of the chordal SDP solve on bench/rand W=40 D=20.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched with
torch.distributed.run, one rank per GPU.  One JSON line on rank 0.

  * step   = one ADMM iteration (per-clique PSD projection + consensus/multiplier updates) over the
             whole SDP, all data resident in HBM before the timed region.
  * N = 1  : workload = BASELINE.json configs[2], "bench/rand W=40 D=20, full chordal decomposition,
             all cliques batched on 1 x MI355X" (fixture tests/golden/problem_W40-D20_b0.npz).
  * N > 1  : `value` = the north star's clique-sharded figure: ONE W40-D20 SDP, its PSD blocks sharded over the N GPUs
             (contiguous ranges balanced by n_k^3), one RCCL all-reduce of the consensus sum per iteration ("strong"
             scaling).  The replica figure - N independent SDPs, one per GPU, no data-path collective ("weak") - is
             measured in the same run and reported beside it under "replicas".  torch.distributed (gloo) only launches,
             synchronises and takes the max over ranks; the data-path collective is the library's own ncclAllReduce.
             If the sharded leg cannot start (RCCL unavailable), `value` falls back to the replica figure and says so.
             See DESIGN.md "Multi-GPU".
  * roofline: the north star scores the eigendecomposition (projection) kernel against HBM bandwidth, so `bound` = "hbm":
             achieved = algorithmic bytes per launch / average duration of the projection launch(es) from HIP events recorded on
             the solver's stream inside the timed region.  Algorithmic bytes: SURVEY.md section 8(d)'s 2 x 8 x sum n_k^2 (nu read,
             w written) PLUS the persistent eigenbasis, read and written once per launch by design (another 2 x 8 x sum n_k^2 -
             it is the state the refinement stage updates); both figures are in the object.  The fp64 matrix-pipe figure
             (10 x sum n_k^3 flop against 78.6 TFLOP/s), which is what actually bounds the kernel, rides beside it under "mfma".
  * cpu_baseline: the C++/OpenMP port of the same ADMM iteration (oracle/c, LAPACK dsyevd per clique) timed on the host
             cores for a bounded sample, plus its estimated time to the certificates the GPU solves reached.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# the one-stream-per-SDP comparison of the batched leg needs more than the default 4 hardware queues (measured:
# 13 SDPs 9.9k -> 17.6k aggregate iterations/s with 16 queues).  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "nn-sdp_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = fp64 matrix peak (half the fp32 vector peak of 157.3)
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(workload: str, beta: int, seconds: float, gpu_iters_to_cert=None):
    """Same-box CPU restatement: the oracle's ADMM iteration in C++ / OpenMP with LAPACK dsyevd per
    clique (oracle/c/admm_cpu.cpp, one thread per clique) on the host cores - a reported baseline, not the target.  The
    reference's own CPU path (Julia + MOSEK) cannot run here.  Falls back to the numpy oracle if the C++ port is not built."""
    import helpers
    from oracle import operator as oop, admm as oadmm
    cores = min(16, os.cpu_count() or 1)
    d = helpers.load_problem(workload, beta)
    q = helpers.oracle_query(d)
    out = {"unit": "ADMM iters/s", "cores": cores, "kind": "port"}
    try:
        from oracle import admm_c
        rates = {}
        for mode in ("single", "double"):
            P = oadmm.ScaledProblem(oop.build_operator(q, mode, normalize=True))
            S = admm_c.CpuAdmm(P, 0.1, 1.6, threads=cores)
            S.step(3)
            n, t0 = 0, time.time()
            budget = seconds * (0.7 if mode == "single" else 0.3)
            while time.time() - t0 < budget:
                S.step(10)
                n += 10
            rates[mode] = (n, time.time() - t0)
        n, dt = rates["single"]
        out.update(value=n / dt, sample=f"{n} iterations of the C++/OpenMP port of the ADMM iteration (oracle/c/admm_cpu.cpp: LAPACK dsyevd per clique, "
                                        f"{cores} threads, explicit M^-1 as on the GPU) on {workload} beta={beta}, Single decomposition, in {dt:.1f} s")
        n2, dt2 = rates["double"]
        out["double_decomp_iters_per_s"] = n2 / dt2
        if gpu_iters_to_cert:
            # the port runs the very iteration of the GPU solver (iterates agree to 1e-11, tests/test_cpu_port.py).  Its time to the
            # certified-gap certificate is MEASURED: the port's loop (checks every 50 iterations, the oracle's penalty schedule) run
            # for the iteration count the GPU's rule stopped at; for the 1e-6 rule (5x longer) the figure stays an estimate
            need = gpu_iters_to_cert.get("certified_gap_1e-3")
            if need:
                Pd = oadmm.ScaledProblem(oop.build_operator(q, "double", normalize=True))
                C = admm_c.CpuAdmm(Pd, 0.1, 1.6, threads=cores)
                t1, it, next_adapt = time.time(), 0, 50
                while it < need:
                    r = C.step(50)
                    it += 50
                    if it >= next_adapt:
                        next_adapt = max(it + 100, it * 3 // 2)
                        ratio = np.sqrt(max(r["pres"], 1e-300) / max(r["dres"], 1e-300))
                        if ratio > 1.5 or ratio < 0.67:
                            C.S.nu, C.S.sigma = C.nu.copy(), C.sigma
                            C.S.set_sigma(C.sigma * min(max(ratio, 0.2), 5.0))
                            C.nu[:] = C.S.nu
                            C.sigma = C.S.sigma
                out["time_to_cert_s"] = {"certified_gap_1e-3": time.time() - t1}
                out["time_to_cert_note"] = (f"measured: the C++/OpenMP port's loop (Double decomposition, {cores} threads) run for the {it} iterations after which the "
                                            f"GPU's certified-gap rule stopped; iterate at the end: pres {r['pres']:.2e} dres {r['dres']:.2e} objective {r['objective']:.8g} "
                                            "(the port has no certificate polish: 10 ms on the GPU)")
            out["time_to_cert_estimate_s"] = {k: v / (n2 / dt2) for k, v in gpu_iters_to_cert.items() if k != "certified_gap_1e-3" or not need}
        return out
    except Exception as e:      # C++ port not built: numpy oracle
        out["port_error"] = repr(e)
    L = oop.build_operator(q, "single", normalize=True)
    P = oadmm.ScaledProblem(L)
    S = oadmm.AdmmState(P, 0.1, 1.6)
    for _ in range(3):
        S.step()
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        S.step()
        n += 1
    dt = time.time() - t0
    out.update(value=n / dt, sample=f"{n} iterations of the numpy oracle ADMM (oracle/admm.py, LAPACK eigh per clique) on {workload} beta={beta} in {dt:.1f} s")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--burn-in", type=int, default=2000,
                    help="untimed ADMM iterations before the warmup so that the timed window sits in the solver's steady state "
                         "(a solve takes 16k-59k iterations; the first few hundred need 2-3 Jacobi sweeps per projection, the rest ~1)")
    ap.add_argument("--late-burn-in", type=int, default=18000,
                    help="further untimed iterations before a second timed window deep in the solve (reported as late_window; 0 = skip)")
    ap.add_argument("--workload", default="W40-D20")
    ap.add_argument("--beta", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["auto", "replica", "shard"], default="auto",
                    help="N > 1: auto/shard = ONE SDP, cliques sharded over the GPUs, RCCL all-reduce of the consensus sum per iteration "
                         "(strong; the replica figure is reported beside it); replica = one independent SDP per GPU only (weak)")
    ap.add_argument("--batch", type=int, default=13, help="independent SDPs solved side by side in the batched leg (0/1 = skip)")
    ap.add_argument("--cert-seconds", type=float, default=30.0, help="time cap of the time-to-certificate solve (0 = skip)")
    ap.add_argument("--wide-burn-in", type=int, default=8000,
                    help="iterations before the timed window of the wide-block leg (ACAS-Xu shaped network, the reference's 151-wide cliques; 0 = skip)")
    ap.add_argument("--no-torch", action="store_true",
                    help="N = 1 only: do not import torch (device synchronisation through the HIP runtime directly); a diagnostic that separated "
                         "torch from the library when rocprofv3 --pmc died at start-up (the cause was neither: the load order, see below)")
    args = ap.parse_args()

    if os.environ.get("NNSDP_BENCH_WATCHDOG"):      # (diagnostic) Python stack of every thread after that many seconds, then exit: where does a run hang?
        import faulthandler
        faulthandler.enable()
        faulthandler.dump_traceback_later(float(os.environ["NNSDP_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("NNSDP_BENCH_ONE_DEVICE"):    # rehearsal of the N > 1 control flow on a one-GPU box (all ranks on device 0)
        local_rank = 0
    if args.no_torch:
        if world > 1:
            raise SystemExit("--no-torch is for N = 1 (the launcher and the control plane of N > 1 are torch.distributed)")
        # the library first, the HIP runtime calls after it: under rocprofv3 --pmc a code object registered AFTER the runtime is up
        # hangs or kills the tool on this pool (found with NNSDP_BENCH_WATCHDOG: the stack ends in ctypes.CDLL of the library)
        from nnsdp_amd import _lib as _nl
        _nl.load()
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so.7")
        ndev = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(ndev)) != 0 or ndev.value < 1:
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        hip.hipSetDevice(local_rank)

        def dev_sync():
            if hip.hipDeviceSynchronize() != 0:
                raise RuntimeError("hipDeviceSynchronize failed")
    else:
        import torch
        from nnsdp_amd import _lib as _nl
        _nl.load()              # (before the first call that brings the HIP runtime up: see above)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev_sync = torch.cuda.synchronize
    dist = None
    if world > 1:
        # control plane only (launch, barrier, max over ranks, the 128-byte RCCL id): gloo, so that the one RCCL
        # communicator on the GPUs is the library's own
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")

    import helpers
    import nnsdp_amd as na
    d = helpers.load_problem(args.workload, args.beta)
    q = helpers.product_query(d)
    opts = na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9, device=local_rank)

    def barrier():
        dev_sync()
        if dist is not None:
            dist.barrier()
        dev_sync()

    def all_max(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_leg(solver):
        """burn-in + W warmup + exactly K timed iterations (barrier + synchronize on both sides, MAX over ranks)"""
        if args.burn_in > 0:
            solver.advance(args.burn_in)      # regular solve loop (checks, sigma / tolerance adaptation), no stopping
        solver.iterate(args.warmup, time_eig=True)
        barrier()
        t0 = time.perf_counter()
        eig_ms = solver.iterate(args.steps, time_eig=True)   # eager launches + HIP events around the projection kernel
        barrier()
        return all_max(time.perf_counter() - t0), eig_ms

    shard = world > 1 and args.mode in ("auto", "shard")
    transport = None
    shard_error = None
    replicas = None
    if world > 1:
        # replica leg: one independent SDP per GPU, no data-path collective
        rs = na.Solver(q, opts)
        dt_r, _ = timed_leg(rs)
        rs.close()
        replicas = {"value": world * args.steps / dt_r, "unit": "ADMM iters/s", "scaling": "weak", "ms_per_step": 1e3 * dt_r / args.steps,
                    "note": f"{world} independent W40-D20 SDPs, one per GPU, no data-path collective"}
    solver = na.Solver(q, opts)           # setup: pattern, generators, factorisation; everything now in HBM
    if shard:
        ok = 1.0
        try:
            ub = [na.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ub, src=0)
            solver.set_comm(world, rank, ub[0])
        except Exception as e:      # RCCL missing / communicator failure: report, fall back to the replica figure
            shard_error = repr(e)
            ok = 0.0
        transport = "rccl"
        if all_max(1.0 - ok) > 0.0:
            # RCCL could not start (absent, or ranks sharing a device): the library's own device-side transport carries the same
            # clique-sharded iteration (hipIpc-mapped peer buffers); only if that fails too does `value` fall back to the replicas
            shard_error = shard_error or "another rank failed to join the RCCL communicator"
            solver.close()
            solver = na.Solver(q, opts)
            import torch as _t0
            ok2, err2 = 1.0, None
            try:
                solver.set_comm_ipc(world, rank, lambda a_: dist.all_reduce(_t0.from_numpy(a_)))
            except Exception as e:
                ok2, err2 = 0.0, repr(e)
            if all_max(1.0 - ok2) > 0.0:
                shard = False
                shard_error += " | hipIpc transport: " + (err2 or "another rank could not map its peers' buffers")
                solver.close()
                solver = na.Solver(q, opts)
            else:
                transport = "hipIpc"
    dt, eig_ms = timed_leg(solver)
    pres, dres, pobj, dobj = solver.residuals()
    soln = solver.finish()
    sm = soln.summary

    # hipGraph replay rate of the same iteration (what nnsdp_solve itself uses): the graph is captured and instantiated by an untimed
    # call first, and at least 256 iterations are timed whatever --steps is (a 20-step window used to contain the capture)
    solver.iterate(64)                  # every rank runs these too (sharded mode: collective inside)
    dev_sync()
    g_steps = max(args.steps, 2048)
    tg0 = time.perf_counter()
    solver.iterate(g_steps)
    dev_sync()
    graph_ips = g_steps / (time.perf_counter() - tg0)
    # the same measurement deep in the solve: a solve to residuals 1e-6 takes 59 000 iterations in this decomposition, and from about
    # 10 000 on the projection kernel's refinement stage carries nearly every block (the sweeps are the early-phase path)
    late = None
    if args.late_burn_in > 0 and world == 1:
        solver.advance(args.late_burn_in)
        solver.iterate(args.warmup, time_eig=True)
        dev_sync()
        tl0 = time.perf_counter()
        ms_l = solver.iterate(args.steps, time_eig=True)
        dev_sync()
        dtl = time.perf_counter() - tl0
        solver.iterate(64)
        dev_sync()
        tl1 = time.perf_counter()
        solver.iterate(g_steps)
        dev_sync()
        late = {"after_iters": args.burn_in + args.late_burn_in, "iters_per_s": args.steps / dtl, "kernel_avg_us": 1e3 * ms_l / args.steps,
                "graph_replay_iters_per_s": g_steps / (time.perf_counter() - tl1)}
    solver.close()

    ipc_leg = None
    if shard and world > 1 and transport == "rccl":
        # the same clique-sharded SDP over the library's own device-side transport (hipIpc-mapped peer buffers: one-shot all-gather +
        # local reduce in rank order, no library collective in the iteration) - reported beside the RCCL figure, never instead of it
        import torch as _t
        def _ar(a_):
            dist.all_reduce(_t.from_numpy(a_))
        ok, err = 1.0, None
        s_ipc = na.Solver(q, opts)
        try:
            s_ipc.set_comm_ipc(world, rank, _ar)
        except Exception as e:
            ok, err = 0.0, repr(e)
        if all_max(1.0 - ok) > 0.0:
            ipc_leg = {"error": err or "another rank could not map its peers' buffers"}
        else:
            try:
                dt_i, _ = timed_leg(s_ipc)
                ipc_leg = {"value": args.steps / dt_i, "unit": "ADMM iters/s", "ms_per_step": 1e3 * dt_i / args.steps,
                           "note": "nnsdp_solver_set_comm_ipc: eager launches with per-launch events like `value`"}
            except Exception as e:
                ipc_leg = {"error": repr(e)}
        s_ipc.close()

    out = None
    if rank == 0:
        # HBM bytes of the projection kernel per launch from the PMC counters: these need their own rocprofv3 --pmc passes
        # (tools/collect_profiles.sh), so the figure is read from the committed summary of the SAME workload and build round
        # and labelled as such; absent or for another workload it stays null
        traffic, traffic_source = None, None
        pmc_file = next((f for f in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_counters_{args.workload}.json") for r in range(9, 0, -1)) if os.path.exists(f)), "")
        if args.beta == 0 and os.path.exists(pmc_file):
            try:
                pmc = json.load(open(pmc_file))
                kj = [v for k, v in pmc.items() if "k_proj_jacobi" in k][0]
                # FETCH_SIZE / WRITE_SIZE are KB; gfx950 FETCH_SIZE under-reports wide streaming reads by 2x
                # (MI355X_MICROARCH.md, HBM section); our loads are 8 B/lane, for which the guide gives no calibration,
                # so the corrected value is an upper bound
                traffic = (2.0 * kj["FETCH_SIZE"]["mean_per_launch"] + kj["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
                traffic_source = f"profiles/{os.path.basename(pmc_file)} (separate rocprofv3 --pmc passes of this command, offline)"
            except Exception:
                traffic = None
        eig_avg_s = eig_ms * 1e-3 / args.steps
        flops = float(sm["eig_flops_per_iter"])
        byts = float(sm["eig_bytes_per_iter"])
        if shard:
            # the timed kernel is rank 0's launch over ITS blocks: algorithmic work of those only
            bn, st = na.shardPlan(q, opts, world)
            flops = float(sum(10 * n ** 3 for n in bn[st[0]:st[1]]))
            byts = float(sum(16 * n ** 2 for n in bn[st[0]:st[1]]))
        ach_tf = flops / eig_avg_s / 1e12
        byts_basis = 2.0 * byts          # + the persistent eigenbasis: read and written once per launch
        out = {
            "metric": "ADMM iters/sec + wall-clock to eps-cert, bench/rand W=40 D=20",
            "value": (1 if shard or world == 1 else world) * args.steps / dt,
            "unit": "ADMM iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "burn_in_iters": args.burn_in,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if shard else "weak",
            "replicas": replicas, "shard_error": shard_error, "ipc_transport": ipc_leg,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "reference bench/rand random network (fixture), CROWN-sliced intervals and sampled ellipsoid precomputed on the host",
            "config": {"workload": f"bench/rand scale-I2-O2-{args.workload} beta={args.beta}, findEllipsoid on [0.5,1.5]^2, "
                                   f"chordal SingleDecomp, {sm['n_cliques']} PSD blocks (max n={sm['max_clique']}) on 1 GPU"
                                   + ("" if world == 1 else (f"; ONE SDP, cliques sharded over {world} GPUs, one exchange per iteration ({transport})" if shard
                                                             else f"; {world} independent SDPs, one per GPU")),
                       "parallelism": ((f"clique-sharded, 1 all-reduce/iteration ({transport})") if shard else "1 SDP per GPU, cliques batched in one launch")},
            "roofline": {"bound": "hbm", "achieved": byts_basis / eig_avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": byts_basis / eig_avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "k_proj_jacobi (refinement stage + ping-pong sweeps, one launch per step; with blocks above 96 the five k_pipe_* launches in front of it are inside the same events)",
                         "kernel_avg_us": eig_avg_s * 1e6,
                         "algorithmic_bytes_per_launch": byts_basis,
                         "algorithmic_bytes_note": "2 x 8 x sum n_k^2 (nu read, w written: SURVEY section 8d) + 2 x 8 x sum n_k^2 (persistent eigenbasis read and written)",
                         "survey_8d_bytes_per_launch": byts, "survey_8d_frac": byts / eig_avg_s / 1e9 / HBM_PEAK_GBS,
                         "mfma": {"achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP64_PEAK_TFLOPS,
                                  "algorithmic_flops_per_launch": flops,
                                  "note": "what bounds the kernel: one SDP's 19 blocks occupy 19 of 256 CUs and live on those CUs' fp64 matrix pipes (DESIGN.md section 4)"}},
            "eig_share_of_step": eig_avg_s / (dt / args.steps), "avg_jacobi_sweeps": sm["avg_sweeps"],
            "graph_replay": {"value": graph_ips, "unit": "ADMM iters/s", "steps": g_steps, "ms_per_step": 1e3 / graph_ips,
                             "note": "the mode nnsdp_solve itself runs in (7 iterations per hipGraph replay + the check iteration as a graph of its own, no per-launch events); `value` above is the eager rate with HIP events around every projection launch"},
            "graph_replay_iters_per_s": graph_ips, "graph_replay_steps_timed": g_steps,
            "refine_blocks_until_window": sm.get("refine_blocks"),
            "iterate": {"pres": pres, "dres": dres, "objective": pobj, "dual_objective": dobj},
        }
        if late:
            fl = flops / (late["kernel_avg_us"] * 1e-6) / 1e12
            late.update(roofline_frac=byts_basis / (late["kernel_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, mfma_frac=fl / FP64_PEAK_TFLOPS, achieved_TFLOPs=fl,
                        note="second timed window of the same handle, same K steps and per-launch events; `value` and `roofline` above are the first window")
            out["late_window"] = late
    if rank == 0 and world == 1 and not shard:
        # round 1's default window (5 warmup + 20 timed iterations of a FRESH solver, no burn-in), kept for a like-for-like
        # comparison with BENCH_r01.json: the first iterations of a solve are not the regime it spends its time in
        s0 = na.Solver(q, opts)
        s0.iterate(5, time_eig=True)
        dev_sync()
        t1 = time.perf_counter()
        ms0 = s0.iterate(20, time_eig=True)
        dev_sync()
        dt0 = time.perf_counter() - t1
        s0.close()
        out["round1_window"] = {"warmup": 5, "steps": 20, "burn_in_iters": 0, "iters_per_s": 20 / dt0, "kernel_avg_us": 1e3 * ms0 / 20,
                                "note": "the window BENCH_r01.json used; same-box comparison of the two trees: profiles/r02_vs_r01_same_box.log "
                                        "(round-1 tree 4 974 it/s / 161.0 us, this tree 5 421 it/s / 152.3 us in this window)"}
    if rank == 0 and world == 1 and args.cert_seconds > 0 and not shard:
        # wall-clock to certificate on fresh solves (setup + ADMM to eps_rel = 1e-6 + feasibility polish)
        out["time_to_cert"] = {}
        for mode in (na.SingleDecomp(), na.DoubleDecomp(), na.PathDecomp()):   # Single / Double: chordal_sdp.jl:8-9; Path: finer cliques, an extension
            for rule, kw in (("residual_1e-6", dict(eps_rel=1e-6)), ("certified_gap_1e-3", dict(eps_rel=1e-6, cert_tol=1e-3))):
                o2 = na.AdmmSdpOptions(decomp_mode=mode, max_iters=500000, max_time=args.cert_seconds, **kw)
                t1 = time.perf_counter()
                s2 = na.runQuery(q, o2)
                out["time_to_cert"][f"{type(mode).__name__}/{rule}"] = {
                    "wall_s": time.perf_counter() - t1, "setup_s": s2.setup_time, "solve_s": s2.solve_time,
                    "status": s2.termination_status, "iters": s2.summary["iters"], "rho": s2.objective_value,
                    "rho_admm_iterate": s2.summary["objective_admm"], "polish_shift": s2.summary["polish_shift"],
                    "pres": s2.summary["pres"], "dres": s2.summary["dres"], "lambda_max": s2.summary["lambda_max"],
                    "blocks": s2.summary["n_cliques"], "max_block": s2.summary["max_clique"]}
        # BASELINE config 4's network (W40-D40) on this one GPU, the certified-gap rule (a warm-up solve first: the first solve of a new size
        # pays rocSOLVER's kernel loading)
        try:
            q40 = helpers.product_query(helpers.load_problem("W40-D40", 0))
            o40 = na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=500000, max_time=args.cert_seconds, eps_rel=1e-6, cert_tol=1e-3)
            for rep in range(2):
                t1 = time.perf_counter()
                s40 = na.runQuery(q40, o40)
                w40 = time.perf_counter() - t1
            out["time_to_cert_W40-D40"] = {"DoubleDecomp/certified_gap_1e-3": {
                "wall_s": w40, "setup_s": s40.setup_time, "solve_s": s40.solve_time, "status": s40.termination_status, "iters": s40.summary["iters"],
                "rho": s40.objective_value, "lambda_max": s40.summary["lambda_max"], "blocks": s40.summary["n_cliques"], "max_block": s40.summary["max_clique"]}}
        except Exception as e:      # (fixture missing: report, do not fail the bench line)
            out["time_to_cert_W40-D40"] = {"error": repr(e)}
        out["time_to_cert"]["eps"] = ("residual_1e-6: ADMM to pres,dres <= 1e-6 relative, then the feasibility polish. "
                                      "certified_gap_1e-3: stop as soon as the polished (exactly feasible) objective is within 1e-3 of the "
                                      "ADMM primal/dual estimates. rho = objective of the polished point; lambda_max = eigmax(Z(gamma)) "
                                      "in the reference's coordinates (reference acceptance: 1e-6 .. 1e-4)")
    if rank == 0 and world == 1 and args.batch > 1 and not shard:
        # the same kernels with `batch` independent SDPs in lockstep on one GPU: batch handle (one launch per stage for
        # all SDPs), and for comparison the older form with one HIP stream + hipGraph per SDP
        sb = na.SolverBatch([q] * args.batch, opts)
        sb.advance(args.burn_in)
        rates = {}
        for name, fn in (("fused", sb.iterate), ("streams", sb.iterate_streams)):
            fn(max(args.warmup, 16))
            dev_sync()
            tb = time.perf_counter()
            fn(args.steps)
            dev_sync()
            rates[name] = args.batch * args.steps / (time.perf_counter() - tb)
        sb.close()
        agg = rates["fused"]
        out["batched"] = {"sdps": args.batch, "aggregate_iters_per_s": agg, "per_sdp_iters_per_s": agg / args.batch,
                          "aggregate_iters_per_s_one_stream_per_sdp": rates["streams"],
                          "eig_TFLOPs_if_same_share": agg * float(sm["eig_flops_per_iter"]) / 1e12,
                          "eig_frac_of_fp64_peak_if_same_share": agg * float(sm["eig_flops_per_iter"]) / 1e12 / FP64_PEAK_TFLOPS,
                          "note": "independent SDPs (beta sweep of experiments/scale.jl:28, hyperplanes of findReach2Dpoly, ACAS sub-queries) in lockstep on one GPU, "
                                  "nnsdp_batch_*: one launch per stage for all SDPs, hipGraph replay; 13 x 19 blocks = 247 of the 256 CUs"}
    if rank == 0 and world == 1 and args.wide_burn_in > 0 and not shard:
        # BASELINE config 5's shape on ONE GPU: an ACAS-Xu shaped network (5-50x6-5, random weights), reach-hyperplane query, plain interval
        # arithmetic (no neuron stable), the reference's Single cliques = 106 + 4 x 151: the packed-triangle variant of the projection kernel
        from nnsdp_amd import frontend as F
        netw = na.randomNetwork([5] + [50] * 6 + [5], seed=1234)
        x0 = np.full(5, 0.3)
        lo, hi = x0 - 0.05, x0 + 0.05
        xi, acx = F.intervalsWorstCase(lo, hi, netw)
        nrm = np.zeros(5); nrm[0] = 1.0
        qw = na.ReachQuery(ffnet=netw, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=nrm),
                           qc_activs=F.makeQcActivsIntvs(netw, xi, acx, 0))
        sw = na.Solver(qw, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9))
        sw.advance(args.wide_burn_in)
        sw.iterate(16, time_eig=True)
        dev_sync()
        # 2 000 steps: a launch in which one 151-block falls back to the packed sweeps lasts ~1 ms against ~0.19 ms when every block takes
        # the refinement step, and on this network about one visit in ten still falls back - a 200-step window (rounds 3 and 4 until
        # the last day) read 193 us or 1 051 us depending on where it fell
        wsteps = 2000
        t1 = time.perf_counter()
        msw = sw.iterate(wsteps, time_eig=True)
        dev_sync()
        dtw = time.perf_counter() - t1
        smw = sw.finish().summary
        sw.close()
        out["wide_blocks"] = {"workload": "ACAS-Xu shaped 5-50x6-5 (random weights), reach hyperplane, interval arithmetic, SingleDecomp",
                              "blocks": smw["n_cliques"], "max_block": smw["max_clique"], "burn_in_iters": args.wide_burn_in, "steps": wsteps,
                              "iters_per_s": wsteps / dtw, "kernel_avg_us": 1e3 * msw / wsteps, "refine_blocks": smw["refine_blocks"],
                              "note": "one GPU, eager launches with per-launch events like `value` (the events bracket the five k_pipe_* launches + k_proj_jacobi); not BASELINE's metric config"}
        if args.cert_seconds > 0:
            # whole solves of the same query to residuals 1e-5: the reference's cliques, and the decomposition AutoDecomp picks (path cliques 6 x 101)
            for label, mode in (("single_solve", na.SingleDecomp()), ("auto_decomp_solve", na.AutoDecomp())):
                t2 = time.perf_counter()
                s3 = na.runQuery(qw, na.AdmmSdpOptions(decomp_mode=mode, max_iters=300000, eps_rel=1e-5, max_time=args.cert_seconds))
                out["wide_blocks"][label] = {"wall_s": time.perf_counter() - t2, "solve_s": s3.solve_time, "status": s3.termination_status, "iters": s3.summary["iters"],
                                             "bound": s3.objective_value, "blocks": s3.summary["n_cliques"], "max_block": s3.summary["max_clique"],
                                             "lambda_max": s3.summary["lambda_max"], "us_per_iter": 1e6 * s3.solve_time / max(s3.summary["iters"], 1)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        need = None
        if "time_to_cert" in out:
            need = {k.split("/")[1]: v["iters"] for k, v in out["time_to_cert"].items() if isinstance(v, dict) and k.startswith("DoubleDecomp/")}
        out["cpu_baseline"] = cpu_baseline(args.workload, args.beta, args.cpu_seconds, need)
    elif rank == 0:
        out["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
