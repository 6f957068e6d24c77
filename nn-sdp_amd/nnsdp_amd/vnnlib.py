"""VNNLIB specifications and the CNF safety driver, host side.

  read_vnnlib       the "simple" VNNLIB subset of ACAS-Xu style properties   exts/vnnlib_parser.jl:3-216
  hplaneS           safety set  normal' y <= h  as the S matrix              src/Utils/qc.jl:27-37
  loadVnnlibCnf     negated DNF property -> CNF of (input box, safety QC)    experiments/vnnlib_utils.jl:18-55
  loadReluQueriesCnf  ... -> CNF of SafetyQuery                              experiments/vnnlib_utils.jl:58-74
  verifyAcasSpec    clause-by-clause verification with early exit            experiments/acas.jl:76-137
  verifyPairs       (network, spec) table, dump CSV layout                   experiments/acas.jl:139-187
  shardPairs        load-balanced split of the pairs over ranks (multi-GPU: independent units, no collective)

A VNNLIB file states the NEGATION of the property in disjunctive normal form:
    NOT phi = OR_box OR_(A,b) (x in box AND A y <= b).
The property holds iff every (box, A, b) case is refuted, and a case is refuted as soon as ONE row i is shown
impossible on the box:  -A_i y <= -b_i - eps for all x in the box (eps = 1e-4, vnnlib_utils.jl:40-44).  So
phi = AND_cases OR_rows safety(box, normal = -A_i, offset = -b_i - eps): a conjunction of disjunctive clauses, each
sub-query one SDP through runQuery.
"""
from __future__ import annotations

import csv
import dataclasses
import os
import re
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import methods as M
from . import frontend as F
from . import parallel as P

NSD_TOL = 1e-4          # experiments/acas.jl:71
SPEC_EPS = 1e-4         # experiments/vnnlib_utils.jl:41


# ----------------------------------------------------------------------------- parsing
def _statements(text: str) -> List[Any]:
    """s-expressions of the file (comments start at ';'), as nested lists of tokens."""
    text = "\n".join(line.split(";", 1)[0] for line in text.splitlines())
    toks = re.findall(r"\(|\)|[^\s()]+", text)
    out, stack = [], []
    for t in toks:
        if t == "(":
            stack.append([])
        elif t == ")":
            if not stack:
                raise ValueError("vnnlib: unbalanced ')'")
            done = stack.pop()
            (stack[-1] if stack else out).append(done)
        else:
            if not stack:
                raise ValueError(f"vnnlib: token {t!r} outside of a statement")
            stack[-1].append(t)
    if stack:
        raise ValueError("vnnlib: unbalanced '('")
    return out


def _var(tok: str) -> Optional[Tuple[str, int]]:
    m = re.fullmatch(r"([XY])_(\d+)", tok)
    return (m.group(1), int(m.group(2))) if m else None


@dataclass
class _Case:
    lo: np.ndarray
    hi: np.ndarray
    rows: List[np.ndarray]
    rhs: List[float]

    def copy(self):
        return _Case(self.lo.copy(), self.hi.copy(), [r.copy() for r in self.rows], list(self.rhs))


def _apply(case: _Case, cmp: Sequence[Any], nin: int, nout: int) -> None:
    """one comparison (op a b): X bounds tighten the box, Y comparisons add a row  row' y <= rhs
    (update_rv_tuple!, exts/vnnlib_parser.jl:46-96)."""
    if len(cmp) != 3 or cmp[0] not in ("<=", ">=") or not all(isinstance(t, str) for t in cmp):
        raise ValueError(f"vnnlib: unsupported comparison {cmp!r}")
    op, a, b = cmp
    va, vb = _var(a), _var(b)
    if va and va[0] == "X":
        if vb is not None:
            raise ValueError("vnnlib: input constraints must compare X_i with a number")
        i = va[1]
        if not 0 <= i < nin:
            raise ValueError(f"vnnlib: X_{i} out of range")
        v = float(b)
        if op == "<=":
            case.hi[i] = min(case.hi[i], v)
        else:
            case.lo[i] = max(case.lo[i], v)
        if case.lo[i] > case.hi[i]:
            raise ValueError(f"vnnlib: empty interval for X_{i}")
        return
    if vb and vb[0] == "X":
        raise ValueError("vnnlib: input constraints must have the variable first")
    if op == ">=":
        a, b, va, vb = b, a, vb, va
    row, rhs = np.zeros(nout), 0.0
    for v in (va, vb):
        if v and not 0 <= v[1] < nout:
            raise ValueError(f"vnnlib: Y_{v[1]} out of range")
    if va and vb:
        row[va[1]] += 1.0
        row[vb[1]] -= 1.0
    elif va:
        row[va[1]] = 1.0
        rhs = float(b)
    elif vb:
        row[vb[1]] = -1.0
        rhs = -float(a)
    else:
        raise ValueError(f"vnnlib: comparison without a variable {cmp!r}")
    case.rows.append(row)
    case.rhs.append(rhs)


def read_vnnlib(path_or_text: str, num_inputs: int, num_outputs: int):
    """-> list of (box, specs): box = (lo, hi) arrays, specs = list of (A, b) with the unsafe set {y: A y <= b}.
    Cases with the same input box are merged (read_vnnlib_simple, exts/vnnlib_parser.jl:106-216)."""
    text = open(path_or_text).read() if os.path.exists(path_or_text) else path_or_text
    cases = [_Case(np.full(num_inputs, -np.inf), np.full(num_inputs, np.inf), [], [])]
    for st in _statements(text):
        if not st:
            continue
        if st[0] == "declare-const":
            continue
        if st[0] != "assert" or len(st) != 2 or not isinstance(st[1], list):
            raise ValueError(f"vnnlib: unsupported statement {st!r}")
        body = st[1]
        if body and body[0] in ("<=", ">="):
            for c in cases:
                _apply(c, body, num_inputs, num_outputs)
            continue
        if not body or body[0] != "or":
            raise ValueError(f"vnnlib: unsupported assertion {body!r}")
        alts = []
        for alt in body[1:]:
            if isinstance(alt, list) and alt and alt[0] == "and":
                alts.append(alt[1:])
            elif isinstance(alt, list) and alt and alt[0] in ("<=", ">="):
                alts.append([alt])
            else:
                raise ValueError(f"vnnlib: unsupported disjunct {alt!r}")
        new = []
        for c in cases:
            for alt in alts:
                cc = c.copy()
                for cmp in alt:
                    _apply(cc, cmp, num_inputs, num_outputs)
                new.append(cc)
        cases = new
    merged: Dict[bytes, Tuple[Tuple[np.ndarray, np.ndarray], list]] = {}
    for c in cases:
        if not (np.all(np.isfinite(c.lo)) and np.all(np.isfinite(c.hi))):
            raise ValueError("vnnlib: every input needs a lower and an upper bound")
        key = c.lo.tobytes() + c.hi.tobytes()
        A = np.array(c.rows).reshape(len(c.rows), num_outputs)
        merged.setdefault(key, ((c.lo, c.hi), []))[1].append((A, np.array(c.rhs)))
    return list(merged.values())


# ----------------------------------------------------------------------------- queries
def hplaneS(normal, h: float, net: M.FeedFwdNet) -> np.ndarray:
    """S of the safety set {normal' y <= h} on (x_1, y, 1): S23 = normal, S33 = -2h (src/Utils/qc.jl:27-37)."""
    d1, dK = net.xdims[0], net.xdims[-1]
    normal = np.asarray(normal, dtype=np.float64)
    if normal.shape != (dK,):
        raise ValueError("normal must have xdims[K] entries")
    S = np.zeros((d1 + dK + 1, d1 + dK + 1))
    S[d1:d1 + dK, -1] = normal
    S[-1, d1:d1 + dK] = normal
    S[-1, -1] = -2.0 * float(h)
    return S


def scaleS(S, alphas, net: M.FeedFwdNet) -> np.ndarray:
    """Qc.scaleS (src/Qc/output.jl:109-124): the safety matrix for a network whose weights were scaled by alphas
    (loadFromFileScaled), y' = prod(alphas) y: the blocks that touch y are divided by alpha (once per factor of y)."""
    alphas = np.asarray(alphas, dtype=np.float64)
    if len(alphas) != net.K:
        raise ValueError("one alpha per layer")
    a = float(np.prod(alphas))
    d1, dK = net.xdims[0], net.xdims[-1]
    S = np.asarray(S, dtype=np.float64)
    if S.shape != (d1 + dK + 1, d1 + dK + 1):
        raise ValueError("S must be (xdims[1] + xdims[end] + 1) square")
    D = np.concatenate([np.ones(d1), np.full(dK, 1.0 / a), [1.0]])
    out = S * D[:, None] * D[None, :]
    # the reference rebuilds the lower blocks from the upper ones (S12', S13', S23'): the result is symmetric by construction
    iu = np.triu_indices(len(D), 1)
    out[(iu[1], iu[0])] = out[iu]
    return out


def loadVnnlibCnf(spec, net: M.FeedFwdNet, alphas=None):
    """CNF of (QcInputBox, QcSafety): one disjunctive clause per unsafe polytope, one literal per row.  `alphas`: the
    scaling factors of loadFromFileScaled, applied through scaleS as experiments/vnnlib_utils.jl:45-47 does."""
    cnf = []
    for (lo, hi), specs in read_vnnlib(spec, net.xdims[0], net.xdims[-1]):
        qin = M.QcInputBox(x1min=lo, x1max=hi)
        for A, b in specs:
            if len(b) == 0:
                raise ValueError("vnnlib: a case without output constraints is trivially violated")
            lits = []
            for i in range(len(b)):
                S = hplaneS(-A[i], -b[i] - SPEC_EPS, net)
                lits.append((qin, M.QcSafety(S=scaleS(S, alphas, net) if alphas is not None else S)))
            cnf.append(lits)
    return cnf


def loadReluQueriesCnf(net: M.FeedFwdNet, spec, beta: int):
    """CNF of SafetyQuery; the interval pre-processing runs once per input box (vnnlib_utils.jl:58-74)."""
    if beta < 0:
        raise ValueError("beta must be >= 0")
    cache: Dict[bytes, Any] = {}
    out = []
    for clause in loadVnnlibCnf(spec, net):
        qs = []
        for qin, qsafe in clause:
            key = np.asarray(qin.x1min).tobytes() + np.asarray(qin.x1max).tobytes()
            if key not in cache:
                cache[key] = F.makeQcActivs(net, qin.x1min, qin.x1max, beta)
            qs.append(M.SafetyQuery(ffnet=net, qc_input=qin, qc_safety=qsafe, qc_activs=cache[key]))
        out.append(qs)
    return out


def isSolutionGood(soln: M.QuerySolution) -> bool:
    """experiments/acas.jl:71-79 accepts status == "OPTIMAL" OR eigmax(Z) <= NSD_TOL.  There OPTIMAL is MOSEK's
    interior-point certificate; here it only says that the ADMM residuals are small in the solver's coordinates, which does
    not bound eigmax(Z) in the reference's.  So the certificate itself is always checked: multipliers >= 0 and
    eigmax(Z(gamma)) <= NSD_TOL, whatever the status (NUMERICAL_ERROR excepted: never good)."""
    if soln.termination_status == "NUMERICAL_ERROR":
        return False
    lam = soln.summary.get("lambda_max") if soln.summary else None
    if lam is None:
        Z = np.asarray(soln.values["Z"])
        lam = float(np.linalg.eigvalsh(0.5 * (Z + Z.T))[-1])
    gmin = min((float(np.min(soln.values[k])) for k in ("γin", "γout", "γac1", "γac2") if k in soln.values and len(soln.values[k])), default=0.0)
    return bool(np.isfinite(lam) and lam <= NSD_TOL and gmin >= 0.0)


def reachForm(query: M.SafetyQuery, ybounds=None):
    """hyperplane safety literal  normal' y <= h  ->  (ReachQuery, h, h0): the reach query bounds normal' y - h0 on a copy
    of the network whose output bias is shifted along the normal (the LMI is the same quadratic form).  By the
    S-procedure the safety LMI is feasible iff the optimal offset rho' satisfies rho' + h0 <= h; the reach form has
    cost-free multipliers, so the first-order solver eliminates fixed neurons, stops on a certified gap and returns
    the margin.  h0 is a lower bound of normal' y over the box (interval arithmetic on the output bounds), so that the
    optimum is nonnegative - the offset multiplier of a reach query is constrained to gamma_out >= 0."""
    net = query.ffnet
    d1, dK = net.xdims[0], net.xdims[-1]
    S = np.asarray(query.qc_safety.S)
    if np.any(S[:d1 + dK, :d1 + dK] != 0) or np.any(S[:d1, -1] != 0):
        raise ValueError("reach form needs a hyperplane safety set (S = hplaneS(normal, h))")
    normal, h = S[d1:d1 + dK, -1].copy(), -0.5 * float(S[-1, -1])
    nn_ = float(normal @ normal)
    if nn_ == 0.0:
        raise ValueError("reach form needs a nonzero normal")
    if ybounds is None:
        xi, _ = F.makeIntervalsInfo(query.qc_input.x1min, query.qc_input.x1max, net)
        ybounds = xi[-1]
    ylo, yhi = ybounds
    h0 = float(np.maximum(normal, 0) @ ylo + np.minimum(normal, 0) @ yhi)
    h0 -= 1e-6 * max(1.0, abs(h0))
    Ms = [np.array(Mk, dtype=np.float64, copy=True) for Mk in net.Ms]
    Ms[-1][:, -1] -= (h0 / nn_) * normal
    shifted = M.FeedFwdNet(xdims=list(net.xdims), Ms=Ms)
    rq = M.ReachQuery(ffnet=shifted, qc_input=query.qc_input, qc_reach=M.QcReachHplane(normal=normal), qc_activs=query.qc_activs)
    return rq, h, h0


def safetyFromReach(soln: M.QuerySolution, h: float, h0: float = 0.0) -> M.QuerySolution:
    """the safety certificate a reach-hyperplane solution implies: same multipliers, Z_safety = Z_reach - 2 (h - rho) e_a e_a'
    with rho = objective + h0 (NSD whenever Z_reach is and rho <= h).  The literal is certified iff gamma >= 0 and
    eigmax(Z_safety) <= NSD_TOL (termination_status "OPTIMAL", else "INFEASIBLE"); margin = h - rho."""
    rho = float(soln.objective_value) + h0
    Z = np.array(soln.values["Z"], dtype=np.float64, copy=True)
    Z[-1, -1] -= 2.0 * (h - rho)
    lam = float(np.linalg.eigvalsh(0.5 * (Z + Z.T))[-1])
    vals = {k: v for k, v in soln.values.items() if k != "γout"}
    gmin = min((float(np.min(vals[k])) for k in ("γin", "γac1", "γac2") if len(vals[k])), default=0.0)
    ok = bool(np.isfinite(lam) and lam <= NSD_TOL and gmin >= 0.0 and rho <= h)
    vals["Z"] = Z
    summ = dict(soln.summary)
    summ.update(lambda_max=lam, reach_bound=rho, offset=h, margin=h - rho, reach_status=soln.termination_status)
    obj = float(sum(np.sum(vals[k]) for k in ("γin", "γac1", "γac2")))
    return M.QuerySolution(objective_value=obj, values=vals, termination_status="OPTIMAL" if ok else "INFEASIBLE",
                           total_time=soln.total_time, setup_time=soln.setup_time, solve_time=soln.solve_time, summary=summ)


def verifyAcasSpec(net: M.FeedFwdNet, spec, beta: int, opts: M.AdmmSdpOptions,
                   solve: Callable[[Any, M.AdmmSdpOptions], M.QuerySolution] = None, log: Callable[[str], None] = None,
                   batch_clause: bool = False, via_reach: bool = False):
    """Goes through the conjunction; a clause holds as soon as one of its sub-queries is certified, the spec fails as
    soon as a clause has none (experiments/acas.jl:87-137).  -> (solutions tried, number of queries, status).
    batch_clause: the sub-queries of a clause are independent SDPs on one network - solve them in lockstep through the
    batch handle (runQueries) instead of one after the other; every literal of a tried clause then counts as run.
    via_reach: True = decide every literal through the equivalent reach-hyperplane query (reachForm / safetyFromReach);
    "auto" = first give the feasibility form 500 iterations per literal (failed attempts are not recorded).  The
    feasibility form 'min sum(gamma)' of the reference needs an interior-point solver to FAIL quickly; the reach form
    ends in bounded time either way and reports the margin."""
    solve = solve or M.solveQuery
    cnf = loadReluQueriesCnf(net, spec, beta)
    num_queries = sum(len(c) for c in cnf)
    solns, status = [], "safe"
    for ci, clause in enumerate(cnf):
        holds = False
        auto = via_reach == "auto"
        if auto:
            # a comfortably certifiable literal is decided by the feasibility form within a few hundred iterations;
            # everything else goes to the reach form, which ends in bounded time either way
            quick = dataclasses.replace(opts, max_iters=min(int(opts.max_iters), 500), cert_tol=0.0)
            for qi, q in enumerate(clause):
                s = solve(q, quick)
                if isSolutionGood(s):
                    solns.append(s)
                    holds = True
                    if log:
                        log(f"conj {ci + 1}/{len(cnf)} subquery {qi + 1}/{len(clause)}: certified by the feasibility form in {s.summary.get('iters')} iterations")
                    break
            if holds:
                continue
        forms = [reachForm(q) for q in clause] if via_reach else [(q, None, None) for q in clause]
        if batch_clause and len(clause) > 1:
            got = M.runQueries([f[0] for f in forms], opts)
            got = [safetyFromReach(s, f[1], f[2]) if via_reach else s for s, f in zip(got, forms)]
            solns.extend(got)
            holds = any(isSolutionGood(s) for s in got)
            if log:
                log(f"conj {ci + 1}/{len(cnf)}: {len(clause)} subqueries in one batch, certified: {[isSolutionGood(s) for s in got]}")
        else:
            for qi, (q, h, h0) in enumerate(forms):
                s = solve(q, opts)
                if via_reach:
                    s = safetyFromReach(s, h, h0)
                solns.append(s)
                good = isSolutionGood(s)
                if log:
                    log(f"conj {ci + 1}/{len(cnf)} subquery {qi + 1}/{len(clause)}: {s.termination_status} "
                        f"time {s.total_time:.3f}s lambda_max {s.summary.get('lambda_max', float('nan')):.3e} good={good}")
                if good:
                    holds = True
                    break
        if not holds:
            status = "unsafe"
            break
    return solns, num_queries, status


PAIR_COLUMNS = ["acas", "spec", "verif_status", "num_queries", "queries_ran", "avg_query_time", "total_time"]
QUERY_COLUMNS = ["acas", "spec", "qnum", "num_queries", "time", "status", "eigmax"]


def verifyPairs(pairs: Sequence[Tuple[str, M.FeedFwdNet, str, Any]], beta: int, opts: M.AdmmSdpOptions, saveto: str = None,
                solve=None, log=None, batch_clause: bool = False, via_reach: bool = False):
    """pairs: (network name, network, spec name, spec path or text).  Writes the reference's two tables
    (experiments/acas.jl:146-185): `saveto` and `saveto + "-qdf.csv"`, re-saved after every pair."""
    rows, qrows = [], []
    for name, net, sname, spec in pairs:
        solns, nq, status = verifyAcasSpec(net, spec, beta, opts, solve=solve, log=log, batch_clause=batch_clause, via_reach=via_reach)
        good = [s for s in solns if isSolutionGood(s)]
        avg = sum(s.total_time for s in good) / len(good) if good else float("inf")
        rows.append([name, sname, status, nq, len(solns), avg, sum(s.total_time for s in solns)])
        for i, s in enumerate(solns):
            qrows.append([name, sname, i + 1, nq, s.total_time, s.termination_status, s.summary.get("lambda_max", float("nan"))])
        if saveto:
            for path, cols, data in ((saveto, PAIR_COLUMNS, rows), (saveto + "-qdf.csv", QUERY_COLUMNS, qrows)):
                with open(path, "w", newline="") as f:
                    w = csv.writer(f)
                    w.writerow(cols)
                    w.writerows(data)
    return rows, qrows


def pairCost(net: M.FeedFwdNet, spec, beta: int, decomp_mode=M.SingleDecomp) -> float:
    """work estimate of a (network, spec) pair: queries x sum of n_k^3 over the PSD blocks of one query."""
    nq = sum(len(c) for c in loadVnnlibCnf(spec, net))
    return float(nq) * float(sum(len(c) ** 3 for c in M.makeCliques(net.xdims, beta, decomp_mode)))


def shardPairs(pairs, beta: int, world: int, rank: int, decomp_mode=M.SingleDecomp):
    """the pairs this rank verifies: longest-processing-time bins over `world` ranks (BASELINE configs[4]: 'load-balanced
    bins across 8 GPUs'); pairs are independent, so there is no data-path collective."""
    costs = [pairCost(net, spec, beta, decomp_mode) for _, net, _, spec in pairs]
    return [pairs[i] for i in P.bin_pack_by_cost(costs, world)[rank]]
