"""Oracle self-consistency: the structured (per-entry) LMI assembly against the literal dense
restatement of src/Qc/*.jl, clique construction against src/Methods/chordal_cliques.jl and the
sizes tabulated in SURVEY.md section 8, and the committed golden vectors."""
import numpy as np
import pytest

import helpers
from oracle import qc, operator as oop


@pytest.mark.parametrize("name,beta", [("W10-D5", 0), ("W10-D5", 3), ("W10-D10", 2)])
@pytest.mark.parametrize("mode", ["single", "double", "dense"])
def test_structured_matches_literal(name, beta, mode):
    q = helpers.oracle_query(helpers.load_problem(name, beta))
    L = oop.build_operator(q, mode)
    rng = np.random.default_rng(3)
    R = qc.make_R(q.net)
    for _ in range(2):
        g = rng.random(q.ngamma)
        Z1 = qc.assemble_Z_literal(q, g, R)
        Z2 = L.Z_dense(g)
        assert np.abs(Z1 - Z2).max() <= 1e-12 * max(1.0, np.abs(Z1).max())


@pytest.mark.parametrize("name,beta", [("W10-D5", 0), ("W10-D5", 3), ("W10-D10", 2)])
def test_golden_vectors(name, beta):
    q = helpers.oracle_query(helpers.load_problem(name, beta))
    g = helpers.load_golden(name, beta)
    L = oop.build_operator(q, "dense")
    for gam, Z in zip(g["gammas"], g["Zs"]):
        assert np.abs(L.Z_dense(gam) - Z).max() <= 1e-12 * np.abs(Z).max()
    # adjoint probes: <G_i, X>
    A = L.A.toarray() / L.pat.svec_scale()[:, None]
    for X, adj in zip(g["Xs"], g["adj"]):
        xv = X[L.pat.rows, L.pat.cols] * np.where(L.pat.rows == L.pat.cols, 1.0, 2.0)
        assert np.abs(A.T @ xv - adj).max() <= 1e-11 * max(1.0, np.abs(adj).max())
    for mode in ("single", "double"):
        cl = qc.clique_index_sets(q.net, beta, mode)
        ptr, idx = g[f"cl_{mode}_ptr"], g[f"cl_{mode}_idx"]
        assert [idx[ptr[k]:ptr[k + 1]].tolist() for k in range(len(ptr) - 1)] == cl


def _net(xdims):
    from oracle.nnet_io import random_net
    return random_net(xdims, seed=5)


def test_clique_sizes_match_survey_table():
    # W10-D5 -> 23,31,31,31 ; W40-D20 -> 83, 121 x 18 ; W40-D40 -> 83, 121 x 38
    assert [len(c) for c in qc.clique_index_sets(_net([2] + [10] * 5 + [2]), 0, "single")] == [23, 31, 31, 31]
    s = [len(c) for c in qc.clique_index_sets(_net([2] + [40] * 20 + [2]), 0, "single")]
    assert s == [83] + [121] * 18
    s = [len(c) for c in qc.clique_index_sets(_net([2] + [40] * 40 + [2]), 0, "single")]
    assert s == [83] + [121] * 38
    # general rule: first xdims[0]+2W+beta+W+1... middle 3W+1+beta, last 3W+1; Double splits into 2W+beta+1, W+beta+1
    for beta in (0, 2, 7):
        W = 12
        net = _net([3] + [W] * 9 + [2])
        s = [len(c) for c in qc.clique_index_sets(net, beta, "single")]
        assert s[0] == 3 + W + beta + W + 1 and s[-1] == 3 * W + 1
        assert all(v == 3 * W + 1 + beta for v in s[1:-1])
        d = [len(c) for c in qc.clique_index_sets(net, beta, "double")]
        assert d[1] == 2 * W + beta + 1 and d[2] == W + beta + 1


def test_selectors_and_generator_structure():
    assert np.array_equal(qc.E(1, [2, 3, 1]), np.array([[0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 0]], dtype=float))
    assert np.array_equal(qc.Ec([0, 2], 4), np.array([[1, 0, 0, 0], [0, 0, 1, 0]], dtype=float))
    with pytest.raises(AssertionError):
        qc.Ec([2, 0], 4)
    q = helpers.oracle_query(helpers.load_problem("W10-D5", 3))
    L = oop.build_operator(q, "dense")
    # every generator has rank <= 2 and Z(gamma) is supported on the union of clique blocks
    rng = np.random.default_rng(0)
    for i in rng.choice(q.ngamma, 25, replace=False):
        e = np.zeros(q.ngamma)
        e[i] = 1.0
        G = L.Z_dense(e) - L.Z_dense(np.zeros(q.ngamma))
        assert np.linalg.matrix_rank(G, tol=1e-10) <= 2
    Z = L.Z_dense(rng.random(q.ngamma))
    mask = np.zeros_like(Z, dtype=bool)
    for c in qc.clique_index_sets(q.net, 3, "single"):
        mask[np.ix_(c, c)] = True
    assert np.abs(Z[~mask]).max() == 0.0


def test_gamma_layout_and_cost():
    d = helpers.load_problem("W10-D5", 3)
    q = helpers.oracle_query(d)
    nin, nout, n1, n2 = q.gamma_dims()
    acdim = 50
    assert (nin, nout, n1) == (2, 1, acdim)
    assert n2 == (3 + 1) * acdim - 3 * 4 // 2 + 2 * acdim      # lambda_dim + 2 acdim (activ_sector.jl:18-20)
    c = q.cost()
    assert c[nin] == 1.0 and c.sum() == 1.0
    qs = helpers.oracle_query(d, out="safety", S=qc.hplane_S([1.0, 0.0], 5.0, q.net))
    assert qs.cost().sum() == qs.ngamma and qs.gamma_dims()[1] == 0


def test_normalising_congruence_is_a_congruence():
    """solver coordinates: Z~(gamma) = T' Z(gamma) T for every gamma (oracle/operator.py Congruence)."""
    q = helpers.oracle_query(helpers.load_problem("W10-D5", 0))
    Lf = oop.build_operator(q, "dense", normalize=False)
    Ln = oop.build_operator(q, "dense", normalize=True)
    cg = oop.make_congruence(q)
    Zdim = q.net.Zdim
    T = np.zeros((Zdim, cg.nred))
    for i in range(Zdim - 1):
        if cg.newpos[i] >= 0:
            T[i, cg.newpos[i]] = cg.h[i]
        T[i, cg.nred - 1] = cg.m[i]
    T[Zdim - 1, cg.nred - 1] = 1.0
    g = np.random.default_rng(1).random(q.ngamma)
    assert np.abs(T.T @ Lf.Z_dense(g) @ T - Ln.Z_dense(g)).max() <= 1e-11
