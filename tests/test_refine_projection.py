"""The GEMM-only refinement stage of the PSD-projection kernel (nnsdp_options.proj_refine) against LAPACK, through the warm test entry
nnsdp_project_psd_warm: the kernel exactly as the solver runs it from the second iteration on, with the eigenbasis of a PREVIOUS matrix
as its starting point.  Reference of the arithmetic: numpy.linalg.eigh (the oracle's project_psd)."""
import numpy as np
import pytest

import helpers  # noqa: F401
import nnsdp_amd as na
from oracle import admm as oadmm

pytestmark = pytest.mark.gpu


def _sym(rng, n, spectrum):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (Q * spectrum) @ Q.T, Q


def _perturb(rng, A, eta):
    D = rng.standard_normal(A.shape)
    D = 0.5 * (D + D.T)
    return A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D


@pytest.mark.parametrize("n", [27, 41, 57, 68, 85, 96])
@pytest.mark.parametrize("eta", [1e-4, 1e-6])
def test_one_refinement_step_meets_its_tolerance(n, eta):
    """a slowly moving matrix (relative change eta, well separated spectrum): the stage takes its one-step path, and the projection is
    within the level it promises - 30 x tol |A| - of LAPACK's; the sweeps-only kernel on the same input is the control"""
    rng = np.random.default_rng(n)
    spec = np.concatenate([np.linspace(0.2, 2.0, n // 2), -np.linspace(0.1, 1.5, n - n // 2)])
    A0, Q0 = _sym(rng, n, spec)
    A1 = _perturb(rng, A0, eta)
    tol = 0.3 * eta
    # (a converged 57-block rides along so that the launch uses the 1024-thread ping-pong variant, as a solver's launch with blocks of
    # 27 and 68 does; the refinement stage lives in that variant)
    C, QC = _sym(rng, 57, np.linspace(-1.0, 1.0, 57))
    W, V, oc, _ = na.project_psd_warm([A1, C], [Q0, QC], tol, refine=True)
    Wx = oadmm.project_psd(A1)
    assert oc == [1, 1, 0, 0, 0], oc
    assert np.linalg.norm(W[0] - Wx) <= 30 * tol * np.linalg.norm(A1)
    assert np.linalg.norm(V[0].T @ V[0] - np.eye(n)) <= 1e-3                 # orthogonal to second order in the rotation
    Wj, Vj, ocj, _ = na.project_psd_warm([A1], [Q0], tol, refine=False)
    assert ocj == [0, 0, 0, 0, 0]
    assert np.linalg.norm(Wj[0] - Wx) <= tol * np.linalg.norm(A1)


def test_persistent_basis_over_many_steps_stays_accurate():
    """the basis is never recomputed exactly: 40 consecutive small moves, each projection from the basis the previous call returned"""
    rng = np.random.default_rng(7)
    ns = [57, 68, 85]
    mats, bases = [], []
    for n in ns:
        spec = np.concatenate([np.linspace(0.05, 2.0, n - n // 3), -np.linspace(0.05, 1.0, n // 3)])
        A, Q = _sym(rng, n, spec)
        mats.append(A)
        bases.append(Q)
    eta, tol = 3e-5, 1e-5
    steps = 0
    for it in range(40):
        mats = [_perturb(rng, A, eta) for A in mats]
        W, bases, oc, _ = na.project_psd_warm(mats, bases, tol, refine=True)
        steps += oc[1]
        for A, Wk, Vk in zip(mats, W, bases):
            assert np.linalg.norm(Wk - oadmm.project_psd(A)) <= 30 * tol * np.linalg.norm(A), it
            assert np.linalg.norm(Vk.T @ Vk - np.eye(len(A))) <= 1e-3
    assert steps >= 100          # nearly every visit took the one-step path (3 blocks x 40 calls)


@pytest.mark.parametrize("case", ["near_zero_cluster", "degenerate_pairs", "large_move", "non_orthogonal_basis", "already_converged"])
def test_refinement_edge_cases_fall_back_to_exact_sweeps(case):
    rng = np.random.default_rng(11)
    n = 68
    tol = 1e-7
    if case == "near_zero_cluster":
        # many eigenvalues at the 1e-9 .. 1e-7 level around zero (the late-solve picture: rank X + rank Z = n): the rotations among
        # them are large but their couplings are below the tolerance - the stage leaves them alone and still takes its step
        spec = np.concatenate([np.linspace(0.3, 2.0, 30), 1e-8 * rng.standard_normal(18), -np.linspace(0.2, 1.0, 20)])
        A0, Q0 = _sym(rng, n, spec)
        A1 = _perturb(rng, A0, 1e-6)
        expect_step = True
    elif case == "degenerate_pairs":
        # exactly repeated LARGE eigenvalues: first order cannot resolve rotations inside the eigenspaces, nor does it have to
        spec = np.concatenate([np.repeat(np.linspace(0.5, 2.0, 17), 2), -np.repeat(np.linspace(0.3, 1.0, 17), 2)])
        A0, Q0 = _sym(rng, n, spec)
        A1 = _perturb(rng, A0, 1e-6)
        expect_step = None          # either path; the result must be right
    elif case == "large_move":
        spec = np.concatenate([np.linspace(0.2, 2.0, 34), -np.linspace(0.1, 1.5, 34)])
        A0, Q0 = _sym(rng, n, spec)
        A1 = _perturb(rng, A0, 0.2)
        expect_step = False
    elif case == "non_orthogonal_basis":
        spec = np.concatenate([np.linspace(0.2, 2.0, 34), -np.linspace(0.1, 1.5, 34)])
        A0, Q0 = _sym(rng, n, spec)
        A1 = _perturb(rng, A0, 0.05)                                      # too far for one step: the sweeps run ...
        Q0 = Q0 @ (np.eye(n) + 1e-3 * rng.standard_normal((n, n)))       # ... from a basis that must first be re-orthogonalised
        expect_step = False
    else:
        spec = np.concatenate([np.linspace(0.2, 2.0, 34), -np.linspace(0.1, 1.5, 34)])
        A1, Q0 = _sym(rng, n, spec)
        expect_step = None
    W, V, oc, _ = na.project_psd_warm([A1], [Q0], tol, refine=True)
    Wx = oadmm.project_psd(A1)
    lim = (30 * tol if oc[1] or oc[4] else tol) * np.linalg.norm(A1) + 1e-12
    assert np.linalg.norm(W[0] - Wx) <= lim, (case, oc, np.linalg.norm(W[0] - Wx) / np.linalg.norm(A1))
    if expect_step is True:
        assert oc[1] == 1, oc
    if expect_step is False:
        assert oc[2] == 1, oc
        assert np.linalg.norm(V[0].T @ V[0] - np.eye(n)) <= 1e-9          # the sweeps got (and kept) an orthogonal basis
    W2, V2, oc2, _ = na.project_psd_warm([A1], [Q0], tol, refine=2)      # with the checked form allowed: same bound
    assert np.linalg.norm(W2[0] - Wx) <= (30 * tol if oc2[1] or oc2[4] else tol) * np.linalg.norm(A1) + 1e-12, (case, oc2)
    if case == "already_converged":
        assert oc[0] == 1, oc


def test_gram_product_is_skipped_three_visits_in_four_and_the_defect_estimate_holds():
    """with the per-block state carried between calls (as a solver carries it between iterations) the stage measures V'V on one visit
    in four and runs the others with R = 0; the estimate of |I - V'V|_F it keeps must stay ABOVE the defect numpy measures on the
    returned basis (it is what decides when to measure again), and the projections must stay inside the promised level"""
    rng = np.random.default_rng(21)
    ns = [57, 68, 85]
    mats, bases = [], []
    for n in ns:
        spec = np.concatenate([np.linspace(0.05, 2.0, n - n // 3), -np.linspace(0.05, 1.0, n // 3)])
        A, Q = _sym(rng, n, spec)
        mats.append(A)
        bases.append(Q)
    eta, tol = 3e-5, 1e-5
    state = np.zeros(4 * len(ns), dtype=np.int32)
    credits, steps = [], 0
    for it in range(24):
        mats = [_perturb(rng, A, eta) for A in mats]
        before = state.reshape(-1, 4)[:, 0].copy()
        W, bases, oc, _ = na.project_psd_warm(mats, bases, tol, refine=True, state=state)
        steps += oc[1]
        words = state.reshape(-1, 4)
        credits.append([(int(w) >> 24) & 15 for w in words[:, 0]])
        est = words[:, 2:4].copy().view(np.float64).ravel()
        for k, (A, Wk, Vk) in enumerate(zip(mats, W, bases)):
            n = len(A)
            assert np.linalg.norm(Wk - oadmm.project_psd(A)) <= 30 * tol * np.linalg.norm(A), (it, k)
            defect = np.linalg.norm(Vk.T @ Vk - np.eye(n))
            assert defect <= max(est[k], 1e-13) * 1.01 + 1e-13, (it, k, defect, est[k])
            assert est[k] <= 0.03 * 30 * tol + 1e-3                       # far below anything that matters; a larger one forces a measurement
            if ((int(before[k]) >> 24) & 15) > 0 and oc[1] == len(ns):     # a visit that ran on credit used one up
                assert credits[-1][k] == ((int(before[k]) >> 24) & 15) - 1
    assert steps >= 60
    assert credits[0] == [3, 3, 3]                   # the first visit (zero state) measured and granted three visits of credit
    assert credits[1] == [2, 2, 2] and credits[2] == [1, 1, 1] and credits[3] == [0, 0, 0] and credits[4] == [3, 3, 3]


@pytest.mark.parametrize("n", [57, 85, 101, 128, 151])
def test_one_unresolvable_pair_across_zero_is_rotated_exactly(n, monkeypatch):
    """the late-solve rejection: ONE pair of eigenvalues on either side of zero whose coupling is far above their gap (first order cannot
    resolve it, and it counts whichever side the projection is rebuilt from).  The stage rotates that pair exactly and takes its step;
    without the rotation (NNSDP_REFINE_PIVOTS=0) the same input goes on to the sweeps.  Both within their bounds of LAPACK."""
    rng = np.random.default_rng(n)
    h = n // 2
    spec = np.concatenate([np.linspace(0.2, 2.0, h - 1), [1e-4, -1e-4], -np.linspace(0.1, 1.5, n - h - 1)])
    Q0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    E = 1e-7 * rng.standard_normal((n, n))
    E = 0.5 * (E + E.T)
    E[h - 1, h] = E[h, h - 1] = 1e-3
    A1 = Q0 @ (np.diag(spec) + E) @ Q0.T
    A1 = 0.5 * (A1 + A1.T)
    tol = 1e-7
    assert 1e-3 > 30 * tol * np.linalg.norm(A1)            # the coupling alone is above the level a step may leave behind
    Wx = oadmm.project_psd(A1)
    W, V, oc, _ = na.project_psd_warm([A1], [Q0], tol, refine=True)
    assert oc[1] == 1, oc
    assert np.linalg.norm(W[0] - Wx) <= 30 * tol * np.linalg.norm(A1)
    assert np.linalg.norm(V[0].T @ V[0] - np.eye(n)) <= 1e-6
    # the basis it returns diagonalises the matrix: a second visit finds the block converged (nothing left of the pair)
    W2, V2, oc2, _ = na.project_psd_warm([A1], [V[0]], 30 * tol, refine=True)
    assert oc2[0] == 1, oc2
    monkeypatch.setenv("NNSDP_REFINE_PIVOTS", "0")
    W0, V0, oc0, _ = na.project_psd_warm([A1], [Q0], tol, refine=True)
    assert oc0[2] == 1, oc0
    assert np.linalg.norm(W0[0] - Wx) <= tol * np.linalg.norm(A1)


@pytest.mark.parametrize("n", [97, 101, 103, 128, 129, 151, 160])      # (97 .. 128: the variant a solver's warm iterations take with the stage on)
@pytest.mark.parametrize("eta", [1e-4, 1e-6])
def test_packed_variant_takes_the_refinement_step(n, eta):
    """blocks 129 .. 160 (the reference's 151-wide cliques of width-50 networks): the stage in its packed form - antisymmetric K in the
    packed triangle, X = I + K + K^2 / 2 and V X streamed through the LDS panel - takes the step on a slowly moving matrix and meets
    the level it promises; the packed sweeps on the same input are the control"""
    rng = np.random.default_rng(n)
    spec = np.concatenate([np.linspace(0.2, 2.0, n // 2), -np.linspace(0.1, 1.5, n - n // 2)])
    A0, Q0 = _sym(rng, n, spec)
    A1 = _perturb(rng, A0, eta)
    tol = 0.3 * eta
    W, V, oc, _ = na.project_psd_warm([A1], [Q0], tol, refine=True)
    Wx = oadmm.project_psd(A1)
    assert oc == [0, 1, 0, 0, 0], oc
    assert np.linalg.norm(W[0] - Wx) <= 30 * tol * np.linalg.norm(A1)
    assert np.linalg.norm(V[0].T @ V[0] - np.eye(n)) <= 1e-3
    Wj, Vj, ocj, _ = na.project_psd_warm([A1], [Q0], tol, refine=False)
    assert ocj == [0, 0, 0, 0, 0]
    assert np.linalg.norm(Wj[0] - Wx) <= tol * np.linalg.norm(A1)


def test_packed_variant_persistent_basis_and_fall_back():
    """the packed stage over 20 consecutive small moves with its state carried (Gram product on one visit in four), then a large move
    that must go on to the sweeps, then a basis with a defect that the Gram visit has to remove (Newton-Schulz pass + congruence again)"""
    rng = np.random.default_rng(5)
    ns = [151, 106, 160, 101, 128]
    mats, bases = [], []
    for n in ns:
        spec = np.concatenate([np.linspace(0.05, 2.0, n - n // 3), -np.linspace(0.05, 1.0, n // 3)])
        A, Q = _sym(rng, n, spec)
        mats.append(A)
        bases.append(Q)
    eta, tol = 3e-5, 1e-5
    state = np.zeros(4 * len(ns), dtype=np.int32)
    steps = 0
    for it in range(20):
        mats = [_perturb(rng, A, eta) for A in mats]
        W, bases, oc, _ = na.project_psd_warm(mats, bases, tol, refine=True, state=state)
        steps += oc[1]
        for A, Wk, Vk in zip(mats, W, bases):
            assert np.linalg.norm(Wk - oadmm.project_psd(A)) <= 30 * tol * np.linalg.norm(A), it
            assert np.linalg.norm(Vk.T @ Vk - np.eye(len(A))) <= 1e-6
    assert steps >= 55
    big = [_perturb(rng, A, 0.2) for A in mats]
    W, V2, oc, _ = na.project_psd_warm(big, bases, 1e-7, refine=True)
    assert oc[2] == len(ns), oc
    for A, Wk in zip(big, W):
        assert np.linalg.norm(Wk - oadmm.project_psd(A)) <= 1e-7 * np.linalg.norm(A) + 1e-12
    bent = [Q @ (np.eye(len(Q)) + 1e-5 * rng.standard_normal(Q.shape)) for Q in bases]
    mats = [_perturb(rng, A, 1e-6) for A in mats]
    W, V3, oc, _ = na.project_psd_warm(mats, bent, 1e-6, refine=True)
    for A, Wk, Vk, Qb in zip(mats, W, V3, bent):
        assert np.linalg.norm(Wk - oadmm.project_psd(A)) <= 30 * 1e-6 * np.linalg.norm(A)
        before = np.linalg.norm(Qb.T @ Qb - np.eye(len(A)))
        assert before >= 1e-3
        assert np.linalg.norm(Vk.T @ Vk - np.eye(len(A))) <= before * before          # the pass squares the defect (3/8 R^2)


def test_solver_with_and_without_refinement_agree():
    """a whole solve: same optimum, the refinement stage carries most block visits late in the solve"""
    q = helpers.product_query(helpers.load_problem("W40-D20", 0))
    on = na.runQuery(q, na.AdmmSdpOptions(max_iters=60000, eps_rel=1e-6, decomp_mode=na.DoubleDecomp()))
    off = na.runQuery(q, na.AdmmSdpOptions(max_iters=60000, eps_rel=1e-6, decomp_mode=na.DoubleDecomp(), proj_refine=0))
    assert on.termination_status == off.termination_status == "OPTIMAL"
    assert abs(on.summary["objective_admm"] - off.summary["objective_admm"]) <= 2e-5 * abs(off.summary["objective_admm"])
    assert abs(on.summary["iters"] - off.summary["iters"]) <= 0.25 * off.summary["iters"]
    print("solve seconds with / without the refinement stage:", on.solve_time, off.solve_time)
    rb = on.summary["refine_blocks"]
    assert off.summary["refine_blocks"] == [0, 0, 0, 0, 0]
    assert rb[1] > 0 and sum(rb) > 0
    print("refinement block visits [converged, one step, sweeps, skipped, checked step]:", rb, "iters", on.summary["iters"], off.summary["iters"])


# ---- the tile-parallel pipeline (csrc/refine_pipe.hpp): refine = 4 of the warm entry puts its five launches in front of the kernel ----

@pytest.mark.parametrize("n", [41, 57, 68, 85, 96, 97, 101, 103, 121, 128, 129, 151, 160])
@pytest.mark.parametrize("neg_share", [1 / 3, 0.7])
def test_pipeline_one_step_meets_its_tolerance(n, neg_share):
    """the same step as the kernel's stage, spread over the chip: a slowly moving matrix takes it and the projection is within the level
    it promises of LAPACK's, rebuilt from the positive side (few positive eigenvalues... or many: the negative side, W = sym(nu) - sum),
    the result exactly symmetric, the basis orthogonal to second order; several blocks of mixed sizes in one launch"""
    rng = np.random.default_rng(n)
    m = max(1, int(n * neg_share))
    spec = np.concatenate([np.linspace(0.2, 2.0, n - m), -np.linspace(0.1, 1.5, m)])
    eta, tol = 1e-6, 3e-7
    base = [_sym(rng, n, spec), _sym(rng, max(n - 13, 17), np.linspace(-1.0, 1.3, max(n - 13, 17))), _sym(rng, n, spec[::-1].copy())]
    mats = [_perturb(rng, A, eta) for A, _ in base]
    W, V, oc, _ = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=4)
    assert oc == [0, 3, 0, 0, 0], oc
    for A, Wk, Vk in zip(mats, W, V):
        assert np.linalg.norm(Wk - oadmm.project_psd(A)) <= 30 * tol * np.linalg.norm(A)
        assert np.abs(Wk - Wk.T).max() == 0.0
        assert np.linalg.norm(Vk.T @ Vk - np.eye(len(A))) <= 1e-3
    # the one-CU stage on the same input takes the same decision and lands on the same projection
    W1, V1, oc1, _ = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=1)
    if n > 40:
        assert oc1 == oc
    for Wa, Wb, A in zip(W, W1, mats):
        assert np.linalg.norm(Wa - Wb) <= 30 * tol * np.linalg.norm(A)


@pytest.mark.parametrize("n", [68, 85, 101, 151])
def test_pipeline_converged_blocks_and_fall_back(n):
    """a block that arrives converged is rebuilt from its basis as it is (no step); a block that moved too far is left to the kernel
    launched behind the pipeline (its sweeps: exact); both in one launch"""
    rng = np.random.default_rng(100 + n)
    spec = np.concatenate([np.linspace(0.2, 2.0, n - n // 3), -np.linspace(0.1, 1.5, n // 3)])
    A0, Q0 = _sym(rng, n, spec)
    A1, Q1 = _sym(rng, n, spec)
    far = _perturb(rng, A1, 3e-2)
    tol = 1e-7
    W, V, oc, _ = na.project_psd_warm([A0, far], [Q0, Q1], tol, refine=4)
    assert oc[0] == 1 and oc[2] == 1, oc
    assert np.linalg.norm(W[0] - oadmm.project_psd(A0)) <= tol * np.linalg.norm(A0)
    assert np.linalg.norm(W[1] - oadmm.project_psd(far)) <= tol * np.linalg.norm(far)
    assert np.linalg.norm(V[1].T @ V[1] - np.eye(n)) <= 1e-9


def test_pipeline_persistent_basis_over_many_steps():
    """40 consecutive small moves through the pipeline with the state carried, each from the basis the previous call returned: the Gram
    product on one visit in four, the defect estimate above the measured defect, every projection inside the promised level"""
    rng = np.random.default_rng(31)
    ns = [57, 85, 106, 151]
    mats, bases = [], []
    for n in ns:
        spec = np.concatenate([np.linspace(0.05, 2.0, n - n // 3), -np.linspace(0.05, 1.0, n // 3)])
        A, Q = _sym(rng, n, spec)
        mats.append(A)
        bases.append(Q)
    eta, tol = 3e-5, 1e-5
    state = np.zeros(4 * len(ns), dtype=np.int32)
    steps, credits = 0, []
    for it in range(40):
        mats = [_perturb(rng, A, eta) for A in mats]
        W, bases, oc, _ = na.project_psd_warm(mats, bases, tol, refine=4, state=state)
        steps += oc[1]
        words = state.reshape(-1, 4)
        credits.append([(int(w) >> 24) & 15 for w in words[:, 0]])
        est = words[:, 2:4].copy().view(np.float64).ravel()
        for k, (A, Wk, Vk) in enumerate(zip(mats, W, bases)):
            assert np.linalg.norm(Wk - oadmm.project_psd(A)) <= 30 * tol * np.linalg.norm(A), (it, k)
            assert np.linalg.norm(Vk.T @ Vk - np.eye(len(A))) <= max(est[k], 1e-13) * 1.01 + 1e-13
    assert steps >= 150
    assert credits[0] == [3] * 4 and credits[1] == [2] * 4 and credits[3] == [0] * 4 and credits[4] == [3] * 4


@pytest.mark.parametrize("pipe", ["3", "1"])
def test_solver_with_the_pipeline_agrees(pipe, monkeypatch):
    """whole solves with the pipeline (NNSDP_PIPE = 3: always on, blocks up to 85; 1: the default rule on a width-50 network in the Path
    decomposition, blocks of 101) and without it: same optimum, comparable iteration counts, the stage carries the block visits"""
    if pipe == "3":
        q = helpers.product_query(helpers.load_problem("W40-D20", 0))
        opts = dict(max_iters=60000, eps_rel=1e-6, decomp_mode=na.DoubleDecomp())
    else:
        from nnsdp_amd import frontend as F
        from oracle import nnet_io
        onet = nnet_io.random_net([5] + [50] * 6 + [5], seed=1)
        net = na.FeedFwdNet(xdims=onet.xdims, Ms=onet.Ms)
        lo, hi = np.full(5, 0.25), np.full(5, 0.35)
        xi, acx = F.intervalsWorstCase(lo, hi, net)
        nrm = np.zeros(5); nrm[0] = 1.0
        q = na.ReachQuery(ffnet=net, qc_input=na.QcInputBox(x1min=lo, x1max=hi), qc_reach=na.QcReachHplane(normal=nrm), qc_activs=F.makeQcActivsIntvs(net, xi, acx, 0))
        opts = dict(max_iters=60000, eps_rel=1e-5, decomp_mode=na.PathDecomp())
    monkeypatch.setenv("NNSDP_PIPE", "0")
    off = na.runQuery(q, na.AdmmSdpOptions(**opts))
    monkeypatch.setenv("NNSDP_PIPE", pipe)
    on = na.runQuery(q, na.AdmmSdpOptions(**opts))
    assert on.termination_status == off.termination_status == "OPTIMAL"
    lvl = 20 * opts["eps_rel"]      # (two runs to the same residual level agree in the objective at a small multiple of that level)
    assert abs(on.summary["objective_admm"] - off.summary["objective_admm"]) <= lvl * abs(off.summary["objective_admm"])
    assert abs(on.objective_value - off.objective_value) <= lvl * abs(off.objective_value)
    assert abs(on.summary["iters"] - off.summary["iters"]) <= 0.25 * off.summary["iters"]
    assert on.summary["lambda_max"] <= 1e-7
    print("solve seconds with / without the pipeline:", on.solve_time, off.solve_time, "block visits", on.summary["refine_blocks"])


@pytest.mark.parametrize("n", [57, 85, 96])
def test_two_workgroups_per_block_give_identical_bits(n, monkeypatch):
    """ProjArgs::split (off by default: measured slower): the warm-start congruence shared between two workgroups
    per block - helper's tiles through L2 behind a release / acquire pair - must reproduce the one-workgroup launch bit for bit
    (same instruction sequence per tile), on the step path, the converged path and the sweeps"""
    rng = np.random.default_rng(n)
    def sym():
        spec = np.concatenate([np.linspace(0.2, 2.0, n - n // 3), -np.linspace(0.1, 1.5, n // 3)])
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        return (Q * spec) @ Q.T, Q
    base = [sym() for _ in range(7)]
    for eta, tol, refine in ((1e-6, 3e-7, True), (0.0, 1e-7, True), (3e-2, 1e-6, True), (1e-6, 3e-7, False)):
        mats = []
        for A, _ in base:
            D = rng.standard_normal(A.shape); D = 0.5 * (D + D.T)
            mats.append(A + eta * np.linalg.norm(A) / np.linalg.norm(D) * D)
        out = []
        for split in ("0", "1"):
            monkeypatch.setenv("NNSDP_SPLIT_WARM", split)
            out.append(na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=refine))
        for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
            assert np.array_equal(a, b)
        assert list(out[0][2]) == list(out[1][2])


def test_solver_iterates_do_not_depend_on_the_second_workgroup(monkeypatch):
    """a solver whose launches hold a block above 80 shares the warm-start congruence of every block with a second workgroup
    (NNSDP_SPLIT, default on there): 1 500 iterations of W40-D20 Single (blocks up to 85, checks, penalty changes, the refinement
    stage coming in) must leave the same bits in the multiplier block with and without it"""
    import hashlib
    q = helpers.product_query(helpers.load_problem("W40-D20", 0))
    dig = []
    for split in ("0", "1"):
        monkeypatch.setenv("NNSDP_SPLIT", split)
        s = na.Solver(q, na.AdmmSdpOptions(decomp_mode=na.SingleDecomp(), max_iters=10 ** 9))
        s.advance(1500)
        res = s.residuals()
        dig.append((hashlib.sha256(s.raw_multipliers().tobytes()).hexdigest(), res))
        s.close()
    assert dig[0] == dig[1]


@pytest.mark.parametrize("n", [85, 151])
def test_pipeline_second_pass_takes_near_misses(n, monkeypatch):
    """a move whose predicted error misses the accepted level by a small factor: with one pass the pipeline hands the block to the
    kernel behind it (sweeps), with the second pass (default) the block takes the step blind, is analysed again on its own measured
    numbers and is carried - inside the level the stage promises either way"""
    rng = np.random.default_rng(500 + n)
    spec = np.concatenate([np.linspace(0.2, 2.0, n - n // 3), -np.linspace(0.1, 1.5, n // 3)])
    base = [_sym(rng, n, spec) for _ in range(3)]
    tol = 3e-7
    found = False
    for eta in (2e-4, 4e-4, 8e-4, 1.6e-3):
        mats = [_perturb(rng, A, eta) for A, _ in base]
        out = {}
        for near in ("0", "10"):
            monkeypatch.setenv("NNSDP_PIPE_NEAR", near)
            out[near] = na.project_psd_warm(mats, [Q for _, Q in base], tol, refine=4)
            for A, Wk, Vk in zip(mats, out[near][0], out[near][1]):
                assert np.linalg.norm(Wk - oadmm.project_psd(A)) <= 30 * tol * np.linalg.norm(A), (eta, near)
                assert np.linalg.norm(Vk.T @ Vk - np.eye(n)) <= 1e-6
        if out["0"][2][1] == 0 and out["10"][2][1] == 3:      # one pass: no block stepped; two passes: all three did
            found = True
    assert found
