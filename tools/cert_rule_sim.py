import re, sys, json
opt = json.load(open('/root/repo/tests/golden/oracle_optimum.json'))
OPT = {"W40-D20 0": opt["W40-D20_b0"]["rho"], "W40-D20 2": opt["W40-D20_b2"]["rho"]}
pat = re.compile(r"it\s+(\d+) t [\d.]+ polished rho ([\d.e+-]+) \(ok (\d) shift ([\d.e+-]+)\) admm ([\d.e+-]+) dobj ([\d.e+-]+) pres ([\d.e+-]+) dres ([\d.e+-]+)")
def runs(path):
    cur, out = None, {}
    for ln in open(path):
        if ln.startswith("== "):
            cur = ln[3:].strip(); out.setdefault(cur, [])
        m = pat.search(ln)
        if m and cur:
            out[cur].append(tuple(float(x) for x in m.groups()))
    return out
def sim(rows, rule, tol=1e-3):
    hist = []
    next_cert = 0
    for (it, o, ok, sh, p, d, pr, dr) in rows:
        hist.append((p, d))
        if max(pr, dr) > 1e-3 or it < next_cert: continue
        ref = max(abs(p), abs(d))
        if rule == "old":
            # polish at geometric schedule x1.25, current estimates
            next_cert = max(it + 250, it * 5 / 4)
            if o - min(p, d) <= tol * ref and abs(p - d) <= tol * ref: return it, o
        elif rule == "new":
            if abs(p - d) > tol * ref: continue
            next_cert = max(it + 100, it * 1.1)
            if o - min(p, d) <= tol * ref: return it, o
        elif rule.startswith("win"):
            w = int(rule[3:])
            if abs(p - d) > tol * ref: continue
            lb = min(min(a, b) for a, b in hist[-w:])
            next_cert = max(it + 100, it * 1.1)
            if o - lb <= tol * ref: return it, o
    return None, None
for path in sys.argv[1:]:
    for name, rows in runs(path).items():
        key = " ".join(name.split()[:2])
        truth = OPT.get(key)
        if truth is None:
            truth = min(r[1] for r in rows[-20:]) * (1 - 3e-4)   # rough: final polished minus typical excess
        first_ok = next((r[0] for r in rows if r[1] - truth <= 1e-3 * truth), None)
        print(name, "optimum", truth, "first iteration with polished within 1e-3:", first_ok)
        for rule in ("old", "new", "win5", "win10", "win20"):
            it, o = sim(rows, rule)
            print(f"   {rule:6s} stop {it} certified excess {((o - truth) / truth) if o else None}")

print("\n=== safety: over ALL check points where a rule's conditions hold (schedule ignored): first such iteration, max certified excess")
def cond(rule, row, hist, tol=1e-3):
    (it, o, ok, sh, p, d, pr, dr) = row
    ref = max(abs(p), abs(d))
    if max(pr, dr) > 1e-3: return False
    if rule == "old" or rule == "new":
        return o - min(p, d) <= tol * ref and abs(p - d) <= tol * ref
    if rule.startswith("res"):
        f = float(rule[3:])
        return max(pr, dr) <= f * tol and o - min(p, d) <= tol * ref and abs(p - d) <= tol * ref
    if rule.startswith("win"):
        w = int(rule[3:]); lb = min(min(a, b) for a, b in hist[-w:])
        return abs(p - d) <= tol * ref and o - lb <= tol * ref
for path in sys.argv[1:]:
    for name, rows in runs(path).items():
        key = " ".join(name.split()[:2]); truth = OPT.get(key)
        if truth is None: truth = min(r[1] for r in rows[-20:]) * (1 - 3e-4)
        print(name)
        for rule in ("new", "res0.3", "res0.2", "res0.1", "win5", "win20"):
            hist, hits = [], []
            for r in rows:
                hist.append((r[4], r[5]))
                if cond(rule, r, hist): hits.append((r[0], (r[1] - truth) / truth))
            print(f"   {rule:7s} first {hits[0][0] if hits else None}  max excess {max(h[1] for h in hits) if hits else None:.2e}  hits {len(hits)}" if hits else f"   {rule:7s} none")
