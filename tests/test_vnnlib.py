"""Row f4 of SURVEY.md section 8: VNNLIB specifications -> CNF of safety queries, the clause-by-clause driver with
early exit, its CSV tables and the load-balanced split of (network, spec) pairs over ranks.  CPU tests use a stub
solver (the driver logic is host code); the GPU test verifies a safe and an unsafe property end to end.
The reference publishes no ACAS result table and its checkout holds no ACAS / VNNLIB file:
the specs under tests/golden/vnnlib are synthetic files in the same dialect."""
import csv
import os

import numpy as np
import pytest

import helpers
import nnsdp_amd as na
from nnsdp_amd import vnnlib as vl
from oracle import nnet_io, qc as oqc

SPEC = os.path.join(helpers.GOLDEN, "vnnlib")


def _net(xdims, seed=3):
    n = nnet_io.random_net(xdims, seed=seed)
    return na.FeedFwdNet(xdims=list(n.xdims), Ms=n.Ms)


def test_parser_single_box_single_bound():
    (box, specs), = vl.read_vnnlib(os.path.join(SPEC, "prop_bound.vnnlib"), 2, 2)
    assert np.array_equal(box[0], [0.5, 0.5]) and np.array_equal(box[1], [1.5, 1.5])
    (A, b), = specs
    assert np.array_equal(A, [[-1.0, 0.0]]) and np.array_equal(b, [-10.0])       # Y_0 >= 10  <=>  -Y_0 <= -10


def test_parser_conjunction_and_disjunctions():
    (box, specs), = vl.read_vnnlib(os.path.join(SPEC, "prop_minimal.vnnlib"), 3, 3)
    assert np.allclose(box[0], [-0.25, 0.4, -0.2]) and np.allclose(box[1], [0.25, 0.5, -0.1])
    (A, b), = specs
    assert np.array_equal(A, [[1, -1, 0], [1, 0, -1]]) and np.array_equal(b, [0, 0])
    (box, specs), = vl.read_vnnlib(os.path.join(SPEC, "prop_or_outputs.vnnlib"), 2, 3)
    assert len(specs) == 2                                                       # same box: merged into one case
    assert np.array_equal(specs[0][0], [[-1, 0, 1], [0, -1, 1]]) and np.array_equal(specs[0][1], [0, 0])
    assert np.array_equal(specs[1][0], [[-1, 0, 0], [0, 1, -1], [0, 0, 1]]) and np.array_equal(specs[1][1], [-3.5, 0, 0.25])   # 0.25 >= Y_2
    cases = vl.read_vnnlib(os.path.join(SPEC, "prop_or_inputs.vnnlib"), 2, 2)
    assert len(cases) == 2 and sorted(float(c[0][0][0]) for c in cases) == [0.5, 1.0]
    assert all(np.array_equal(c[1][0][0], [[-1.0, 0.0]]) for c in cases)


@pytest.mark.parametrize("text,msg", [
    ("(assert (<= X_0 1.0))", "lower and an upper"),
    ("(assert (<= X_0 1.0)) (assert (>= X_0 2.0))", "empty interval"),
    ("(assert (<= X_5 1.0))", "out of range"),
    ("(assert (<= X_0 X_1))", "compare X_i with a number"),
    ("(assert (not (<= Y_0 1.0)))", "unsupported assertion"),
    ("(assert (<= Y_0 1.0)", "unbalanced"),
])
def test_parser_rejects_malformed(text, msg):
    with pytest.raises(ValueError) as ei:
        vl.read_vnnlib(text, 2, 2)
    assert msg in str(ei.value)


def test_hplane_S_matches_oracle_and_cnf_shapes():
    net = _net([2, 6, 6, 3])
    S = vl.hplaneS([1.0, -2.0, 0.5], 0.3, net)
    assert np.array_equal(S, oqc.hplane_S([1.0, -2.0, 0.5], 0.3, nnet_io.FeedFwdNet(xdims=net.xdims, Ms=net.Ms)))
    cnf = vl.loadVnnlibCnf(os.path.join(SPEC, "prop_or_outputs.vnnlib"), net)
    assert [len(c) for c in cnf] == [2, 3]
    qin, qs = cnf[1][0]                                                           # row  -Y_0 <= -3.5  ->  Y_0 <= 3.5 - eps
    assert np.array_equal(qs.S[2:5, -1], [1.0, 0.0, 0.0]) and qs.S[-1, -1] == pytest.approx(-2 * (3.5 - 1e-4))
    queries = vl.loadReluQueriesCnf(net, os.path.join(SPEC, "prop_or_outputs.vnnlib"), 1)
    assert [len(c) for c in queries] == [2, 3]
    assert all(q.qc_activs is queries[0][0].qc_activs for c in queries for q in c)   # one pre-processing per input box


class _Stub:
    """stands in for runQuery: certifies the literals whose (clause, literal) index is listed."""

    def __init__(self, good):
        self.good, self.calls = set(good), []

    def __call__(self, query, opts):
        key = tuple(float(v) for v in query.qc_safety.S[:, -1])
        self.calls.append(key)
        ok = key in self.good
        return na.QuerySolution(objective_value=0.0, values={}, termination_status="OPTIMAL" if ok else "ITERATION_LIMIT",
                                total_time=0.5, setup_time=0.1, solve_time=0.4, summary={"lambda_max": 1e-9 if ok else 0.3})


def test_driver_early_exit_semantics(tmp_path):
    net = _net([2, 6, 6, 3])
    spec = os.path.join(SPEC, "prop_or_outputs.vnnlib")
    cnf = vl.loadReluQueriesCnf(net, spec, 0)
    key = lambda c, i: tuple(float(v) for v in cnf[c][i].qc_safety.S[:, -1])
    # clause 1: the second literal holds; clause 2: the first holds -> 3 queries run of 5, safe
    stub = _Stub({key(0, 1), key(1, 0)})
    solns, nq, status = vl.verifyAcasSpec(net, spec, 0, na.AdmmSdpOptions(), solve=stub)
    assert (nq, len(solns), status) == (5, 3, "safe")
    # clause 1 has no certified literal -> unsafe after its 2 queries, clause 2 is never tried
    stub = _Stub({key(1, 0)})
    solns, nq, status = vl.verifyAcasSpec(net, spec, 0, na.AdmmSdpOptions(), solve=stub)
    assert (nq, len(solns), status) == (5, 2, "unsafe")
    # a non-OPTIMAL solution with lambda_max <= 1e-4 still counts (experiments/acas.jl:76-79)
    s = na.QuerySolution(0.0, {}, "SLOW_PROGRESS", 1.0, 0.1, 0.9, {"lambda_max": 5e-5})
    assert vl.isSolutionGood(s) and not vl.isSolutionGood(na.QuerySolution(0.0, {}, "SLOW_PROGRESS", 1.0, 0.1, 0.9, {"lambda_max": 2e-4}))
    # tables (experiments/acas.jl:146-185)
    out = str(tmp_path / "acas.csv")
    rows, qrows = vl.verifyPairs([("net_a", net, "prop_or_outputs", spec)], 0, na.AdmmSdpOptions(), saveto=out,
                                 solve=_Stub({key(0, 1), key(1, 0)}))
    got = list(csv.reader(open(out)))
    assert got[0] == ["acas", "spec", "verif_status", "num_queries", "queries_ran", "avg_query_time", "total_time"]
    assert got[1][:5] == ["net_a", "prop_or_outputs", "safe", "5", "3"] and float(got[1][5]) == 0.5 and float(got[1][6]) == 1.5
    q = list(csv.reader(open(out + "-qdf.csv")))
    assert q[0] == ["acas", "spec", "qnum", "num_queries", "time", "status", "eigmax"] and len(q) == 4
    assert [r[5] for r in q[1:]] == ["ITERATION_LIMIT", "OPTIMAL", "OPTIMAL"]


def test_pairs_are_balanced_over_ranks():
    small, big = _net([2, 6, 6, 3]), _net([2, 12, 12, 12, 3])
    pairs = [("s", small, "p7", os.path.join(SPEC, "prop_or_outputs.vnnlib"))] * 3 + \
            [("b", big, "p7", os.path.join(SPEC, "prop_or_outputs.vnnlib"))] * 2
    cs, cb = vl.pairCost(small, pairs[0][3], 0), vl.pairCost(big, pairs[0][3], 0)
    assert cb > cs > 0
    shards = [vl.shardPairs(pairs, 0, 2, r) for r in range(2)]
    assert sorted(p[0] for s in shards for p in s) == ["b", "b", "s", "s", "s"]
    assert all(sum(1 for p in s if p[0] == "b") == 1 for s in shards)          # the two heavy pairs land on different ranks


@pytest.mark.gpu
def test_verify_safe_and_unsafe_property_on_gpu(tmp_path):
    """W10-D5 reference network on the box [0.5,1.5]^2 (experiments/scale.jl:26-27): the outputs stay far below 10
    (certified: safe); 'first output never below 10' is false at every point of the box (unsafe: no certificate)."""
    d = helpers.load_problem("W10-D5", 0)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    y = na.evalFeedFwdNet(net, np.array([1.0, 1.0]))
    assert abs(y[0]) < 5
    opts = na.AdmmSdpOptions(max_iters=20000, eps_rel=1e-5)
    logs = []
    solns, nq, status = vl.verifyAcasSpec(net, os.path.join(SPEC, "prop_bound.vnnlib"), 1, opts, log=logs.append)
    assert (nq, len(solns), status) == (1, 1, "safe") and vl.isSolutionGood(solns[0]) and len(logs) == 1
    Z = solns[0].values["Z"]
    assert np.linalg.eigvalsh(0.5 * (Z + Z.T))[-1] <= vl.NSD_TOL and all(np.all(solns[0].values[k] >= 0) for k in ("γin", "γac1", "γac2"))
    unsafe = open(os.path.join(SPEC, "prop_bound.vnnlib")).read().replace("(assert (>= Y_0 10.0))", "(assert (<= Y_0 10.0))")
    solns, nq, status = vl.verifyAcasSpec(net, unsafe, 1, na.AdmmSdpOptions(max_iters=3000, eps_rel=1e-5))
    assert status == "unsafe" and not vl.isSolutionGood(solns[0])
    # union of two boxes: two clauses, both certified
    rows, qrows = vl.verifyPairs([("W10-D5", net, "prop_or_inputs", os.path.join(SPEC, "prop_or_inputs.vnnlib"))], 1, opts,
                                 saveto=str(tmp_path / "t.csv"))
    assert rows[0][2:5] == ["safe", 2, 2]


@pytest.mark.gpu
def test_clause_literals_in_one_batch_on_gpu():
    """a clause with several literals (ACAS prop_7 style): sequential early exit and the batched form agree on the verdict;
    the batched form runs every literal of the clauses it tries."""
    d = helpers.load_problem("W10-D5", 0)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    spec = """
    (assert (>= X_0 0.5)) (assert (<= X_0 1.5)) (assert (>= X_1 0.5)) (assert (<= X_1 1.5))
    (assert (or (and (<= Y_0 10.0) (>= Y_1 20.0)) (and (>= Y_0 15.0) (>= Y_1 -50.0))))
    """
    opts = na.AdmmSdpOptions(max_iters=4000, eps_rel=1e-5)
    s1, nq, st1 = vl.verifyAcasSpec(net, spec, 1, opts)
    s2, nq2, st2 = vl.verifyAcasSpec(net, spec, 1, opts, batch_clause=True)
    # clause 1: "Y_0 <= 10" cannot be refuted (it is true), "Y_1 >= 20" can; clause 2: "Y_0 >= 15" can
    assert (nq, nq2, st1, st2) == (4, 4, "safe", "safe")
    assert len(s1) == 3 and len(s2) == 4


def test_reach_form_of_a_safety_literal_cpu():
    """reachForm / safetyFromReach: the hyperplane literal becomes a reach query on the same normal over a network whose
    output bias is shifted by a lower bound h0 of normal' y (the offset multiplier of a reach query is >= 0); a reach
    solution turns into the safety certificate Z_safety = Z_reach - 2 (h - rho) e_a e_a', rho = objective + h0."""
    net = _net([2, 6, 6, 3])
    (qin, qsafe), = vl.loadVnnlibCnf("(assert (>= X_0 0))(assert (<= X_0 1))(assert (>= X_1 0))(assert (<= X_1 1))(assert (>= Y_1 2.5))", net)[0]
    qa = na.makeQcActivs(net, qin.x1min, qin.x1max, 0)
    sq = na.SafetyQuery(ffnet=net, qc_input=qin, qc_safety=qsafe, qc_activs=qa)
    rq, h, h0 = vl.reachForm(sq)
    assert np.array_equal(rq.qc_reach.normal, [0.0, 1.0, 0.0]) and h == pytest.approx(2.5 - 1e-4)
    X = np.random.default_rng(0).random((2, 4000))
    y1 = na.evalFeedFwdNet(net, X)[1]
    assert h0 <= y1.min() and np.allclose(na.evalFeedFwdNet(rq.ffnet, X)[1], y1 - h0) and rq.ffnet is not net
    assert np.array_equal(rq.ffnet.Ms[0], net.Ms[0]) and np.array_equal(rq.ffnet.Ms[-1][:, :-1], net.Ms[-1][:, :-1])
    n = sum(net.xdims) + 1
    vals = {"γin": np.ones(2), "γout": np.array([2.0]), "γac1": np.ones(12), "γac2": np.ones(36), "Z": -np.eye(n)}
    s = vl.safetyFromReach(na.QuerySolution(2.0 - h0, dict(vals), "OPTIMAL", 1.0, 0.1, 0.9, {"lambda_max": -1.0}), h, h0)
    assert s.termination_status == "OPTIMAL" and vl.isSolutionGood(s) and "γout" not in s.values
    assert s.values["Z"][-1, -1] == pytest.approx(-1.0 - 2 * (h - 2.0)) and s.summary["margin"] == pytest.approx(h - 2.0)
    assert s.objective_value == pytest.approx(2 + 12 + 36)
    bad = vl.safetyFromReach(na.QuerySolution(4.0 - h0, dict(vals), "OPTIMAL", 1.0, 0.1, 0.9, {"lambda_max": -1.0}), h, h0)
    assert bad.termination_status == "INFEASIBLE" and bad.summary["margin"] < 0 and not vl.isSolutionGood(bad)
    Sc = qsafe.S.copy(); Sc[0, 0] = 1.0
    with pytest.raises(ValueError):
        vl.reachForm(na.SafetyQuery(ffnet=net, qc_input=qin, qc_safety=na.QcSafety(S=Sc), qc_activs=qa))


@pytest.mark.gpu
def test_via_reach_agrees_and_fails_fast_on_gpu():
    d = helpers.load_problem("W10-D5", 0)
    net = na.FeedFwdNet(xdims=[int(v) for v in d["xdims"]], Ms=helpers.problem_Ms(d))
    spec = os.path.join(SPEC, "prop_bound.vnnlib")
    opts = na.AdmmSdpOptions(max_iters=100000, eps_rel=1e-6, cert_tol=1e-3)
    s1, _, st1 = vl.verifyAcasSpec(net, spec, 1, opts)
    s2, _, st2 = vl.verifyAcasSpec(net, spec, 1, opts, via_reach=True)
    assert st1 == st2 == "safe" and s2[0].summary["margin"] > 0
    Z = s2[0].values["Z"]
    assert np.linalg.eigvalsh(0.5 * (Z + Z.T))[-1] <= vl.NSD_TOL and all(np.all(s2[0].values[k] >= 0) for k in ("γin", "γac1", "γac2"))
    # the certified bound of the reach form is the threshold of the safety form: just above it the property is certified,
    # just below it is not - and the reach form says so in bounded time
    bound = s2[0].summary["reach_bound"]
    box = "(assert (>= X_0 0.5))(assert (<= X_0 1.5))(assert (>= X_1 0.5))(assert (<= X_1 1.5))"
    ok, _, st_ok = vl.verifyAcasSpec(net, box + f"(assert (>= Y_0 {bound + 0.01}))", 1, opts, via_reach=True)
    no, _, st_no = vl.verifyAcasSpec(net, box + f"(assert (>= Y_0 {bound - 0.02 * max(1.0, abs(bound))}))", 1, opts, via_reach=True)
    assert (st_ok, st_no) == ("safe", "unsafe") and no[0].termination_status == "INFEASIBLE" and no[0].summary["margin"] < 0
    au, _, st_au = vl.verifyAcasSpec(net, spec, 1, opts, via_reach="auto")        # Y_0 <= 10: comfortable -> feasibility form
    au2, _, st_au2 = vl.verifyAcasSpec(net, box + f"(assert (>= Y_0 {bound - 0.02 * max(1.0, abs(bound))}))", 1, opts, via_reach="auto")   # not certifiable -> reach form
    assert st_au == "safe" and "margin" not in au[0].summary and au[0].summary["iters"] <= 500
    assert st_au2 == "unsafe" and len(au2) == 1 and au2[0].summary["margin"] < 0
    assert no[0].summary["iters"] == ok[0].summary["iters"]
    # the bound is a genuine maximum estimate (this network's first output is negative on the box: without the bias shift the
    # reach form could only return gamma_out = 0)
    X = 0.5 + np.random.default_rng(0).random((2, 20000))
    y0 = na.evalFeedFwdNet(net, X)[0]
    assert y0.max() <= bound <= y0.max() + 0.5 * (y0.max() - y0.min()) + 0.05
