"""End-to-end parity with EVERY kind of published row of the reference (dump/scale/*.csv -> tests/golden/dump_scale.csv):
NnSdp.findEllipsoid on [0.5,1.5]^2 (experiments/scale.jl:26-27,70) through the product path only (native CROWN intervals,
sampled ellipsoid, Double decomposition, residuals 1e-6, certificate polish), 21 (net, beta) rows with three OPTIMAL
published values each, W in {10, 20}, D from 10 to 100, beta in {0, 3, 7} - including W20-D100 beta = 0 and 7, the reference's
headline rows (dump/scale/chordalsdp2-scale-I2-O2-W20-D100.nnet.csv:2,9).

ONE stated tolerance:   |rho - nearest published| <= 1e-3 |rho| + 1e-9.
Rows outside it are xfail(strict=True) with the measured relative distance (profiles/r02_parity_cause.csv) - they are not
omitted.  DESIGN.md section 7 names the cause (the published objectives are interior-point iterates accepted at MOSEK's
relaxed tolerance, up to 1.3 % above the optimum of the very LMI they solve) and the evidence; the second test below states
what does hold on EVERY row without exception:

    sampled maximum of |invP y - yc|^2   <=   rho_certified   <=   (1 + 1e-3) x smallest published value

i.e. the certificate is sound (never below what 2e4 forward passes reach), exactly feasible for the reference's LMI
(eigmax(Z(gamma)) <= 1e-7 in its own coordinates, gamma >= 0), and never looser than anything the reference published.
"""
import os

import numpy as np
import pytest

import helpers
import nnsdp_amd as na
from nnsdp_amd import frontend as F

pytestmark = pytest.mark.gpu

TOL = 1e-3
# (net, beta, measured relative distance to the nearest published value when outside TOL, else None)
ROWS = [
    ("W10-D10", 0, None), ("W10-D10", 3, None), ("W10-D10", 7, None), ("W10-D20", 0, None),
    ("W10-D30", 0, 5.1e-3), ("W10-D30", 7, None), ("W10-D50", 0, 3.3e-3), ("W10-D60", 0, None), ("W10-D70", 0, None),
    ("W10-D80", 0, 7.3e-3),
    ("W20-D10", 0, 8.4e-3), ("W20-D10", 3, 9.4e-3), ("W20-D10", 7, 1.33e-2), ("W20-D20", 0, 1.7e-3),
    ("W20-D30", 0, 6.2e-3), ("W20-D30", 7, 1.5e-3), ("W20-D40", 0, 5.4e-3), ("W20-D50", 0, 2.3e-3), ("W20-D70", 0, 4.5e-3),
    ("W20-D100", 0, 4.8e-3), ("W20-D100", 7, 3.9e-3),
]
_cache = {}


def _net(name):
    d = np.load(os.path.join(helpers.GOLDEN, "nets", f"scale-I2-O2-{name}.npz"))
    xd = [int(v) for v in d["xdims"]]
    return na.FeedFwdNet(xdims=xd, Ms=[np.array(d[f"M{k}"]) for k in range(len(xd) - 1)])


def _solve_all():
    """all rows, solved in batches of independent SDPs (the batch handle keeps the GPU busy; every SDP stops on its own rule)"""
    if _cache:
        return _cache
    opts = na.AdmmSdpOptions(decomp_mode=na.DoubleDecomp(), max_iters=600000, max_time=240, eps_rel=1e-6)
    qs = {}
    for name, beta, _ in ROWS:
        q, P, yc = na.ellipsoidQuery(_net(name), [0.5, 0.5], [1.5, 1.5], beta)
        qs[(name, beta)] = q
    keys = list(qs)
    for i in range(0, len(keys), 7):
        chunk = keys[i:i + 7]
        for k, s in zip(chunk, na.runQueries([qs[k] for k in chunk], opts)):
            _cache[k] = (qs[k], s)
    return _cache


@pytest.mark.parametrize("name,beta,measured", [
    pytest.param(n, b, m, marks=[pytest.mark.xfail(strict=True, reason=f"{m:.1e} from the nearest published value (> {TOL:g}): MOSEK "
                                                                       "accepted a relaxed-tolerance iterate, see DESIGN.md section 7")] if m else [])
    for n, b, m in ROWS])
def test_published_objective_within_the_stated_tolerance(name, beta, measured):
    q, s = _solve_all()[(name, beta)]
    pub = helpers.published_rho(name, beta)
    assert len(pub) == 3 and s.termination_status == "OPTIMAL"
    rel = min(abs(s.objective_value - p) for p in pub)
    assert rel <= TOL * abs(s.objective_value) + 1e-9, (name, beta, s.objective_value, pub, rel / abs(s.objective_value))


@pytest.mark.parametrize("name,beta,measured", [(n, b, m) for n, b, m in ROWS if m])
def test_rows_outside_the_tolerance_sit_where_they_were_recorded(name, beta, measured):
    """ what IS claimed about the 14 rows above, as assertions that pass: the certified value is BELOW every published
    value (one-signed: ours is the tighter bound) and its relative distance to the nearest one is the recorded one within a band -
    so that a change which moves a row shows up here as a failure with numbers, not only as an unexpected pass of a strict xfail"""
    q, s = _solve_all()[(name, beta)]
    pub = helpers.published_rho(name, beta)
    rho = s.objective_value
    assert s.termination_status == "OPTIMAL"
    assert rho < min(pub), (name, beta, rho, pub)
    rel = min(abs(rho - p) for p in pub) / abs(rho)
    assert abs(rel - measured) <= 0.15 * measured + 2e-4, (name, beta, rel, measured)


@pytest.mark.parametrize("name,beta", [(n, b) for n, b, _ in ROWS])
def test_certificate_is_sound_feasible_and_never_looser_than_the_reference(name, beta):
    q, s = _solve_all()[(name, beta)]
    pub = helpers.published_rho(name, beta)
    rho = s.objective_value
    # exactly feasible for the reference's LMI, in its coordinates, with its interval bounds
    assert s.summary["lambda_max"] <= 1e-7
    Z = s.values["Z"]
    assert np.linalg.eigvalsh(0.5 * (Z + Z.T))[-1] <= 1e-7
    for k in ("γin", "γout", "γac1", "γac2"):
        assert np.min(s.values[k]) >= 0.0
    assert abs(float(s.values["γout"][0]) - rho) == 0.0
    # sound: |invP y - yc|^2 <= rho for sampled trajectories (the set the reference's QcReachEllipsoid certifies, output.jl:91-93)
    rng = np.random.default_rng(7)
    X = 0.5 + rng.random((2, 20000))
    Y = F.evalFeedFwdNet(q.ffnet, X)
    val = np.sum((q.qc_reach.invP @ Y - q.qc_reach.yc[:, None]) ** 2, axis=0).max()
    assert val <= rho * (1 + 1e-9) + 1e-12, (name, beta, val, rho)
    # never looser than the reference: at or below its smallest published value (1e-3 slack for its sampling noise)
    assert rho <= min(pub) * (1 + TOL), (name, beta, rho, pub)
    # ... and within 1.5 % of it (the worst published row, W20-D10 beta=7, sits 1.33 % above)
    assert rho >= min(pub) * (1 - 1.5e-2)


@pytest.mark.parametrize("name,beta", [("W10-D10", 0), ("W20-D10", 0)])
def test_hip_solver_lands_on_the_independent_interior_point_optimum(name, beta):
    """the optimum of the reference's LMI by a different algorithm on a different formulation (oracle/ipm.py: dense primal-dual
    interior point, one cone, the reference's coordinates, literal assembly; tests/golden/ipm_optimum.json) - including W20-D10,
    the row where the published values are furthest (8.4e-3) from ours.  The product's inputs differ from the fixture's only by
    float32 summation order in the CROWN bounds (1e-7)."""
    import json
    g = json.load(open(os.path.join(helpers.GOLDEN, "ipm_optimum.json")))[f"{name}_b{beta}"]
    q, s = _solve_all()[(name, beta)]
    assert abs(s.summary["objective_admm"] - g["rho"]) <= 2e-5 * g["rho"], (s.summary["objective_admm"], g["rho"])
    assert g["lower"] * (1 - 1e-6) <= s.objective_value <= g["rho"] * (1 + 1e-3)      # certified value: an upper bound, within 1e-3
    assert all(g["rho"] < p for p in g["published"])


@pytest.mark.parametrize("name,within,above", [("W10-D30", 3e-4, 4e-3), ("W10-D20", 3e-4, 1e-4)])
def test_deep_row_sits_inside_the_interior_point_bracket(name, within, above):
    """DEEP published rows through the independent interior point: W10-D30 beta = 0, Zdim 303, and W10-D20 (Zdim 203:
    299 iterations, gap 2.5e-7, dual feasible to 1.5e-14, lower bound 1.3463120 with the published values 6e-4 above it).  W10-D30:  The method does not
    converge to 1e-9 there (no Slater point; it stops with a numerical error after 814 iterations at a relative gap of 2e-6), but its dual
    iterate is feasible to 1.5e-12, so its dual objective IS a lower bound of the optimum of the reference's LMI
    (tests/golden/ipm_unconverged.json, profiles/r03_ipm_trace_W10-D30_b0.csv).  Our certified rho - a feasible point, an upper bound - is
    within 3e-4 of that lower bound: the optimum is enclosed, and the three published values lie more than 4e-3 above the enclosure."""
    import json
    g = json.load(open(os.path.join(helpers.GOLDEN, "ipm_unconverged.json")))[f"{name}_b0"]
    assert g["dinf"] <= 1e-10 and g["gap"] <= 1e-5
    q, s = _solve_all()[(name, 0)]
    rho = s.objective_value
    assert g["lower"] * (1 - 1e-6) <= rho, (g["lower"], rho)
    assert rho - g["lower"] <= within * rho, (g["lower"], rho)
    assert abs(s.summary["objective_admm"] - g["rho"]) <= 1e-4 * rho          # the ADMM iterate and the interior point's primal iterate
    assert min(g["published"]) >= rho * (1 + above), (min(g["published"]), rho)
