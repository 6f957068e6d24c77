"""CPU oracle for the Chordal-DeepSDP hot path (TEST INFRASTRUCTURE ONLY).

This package is a numpy / plain-C restatement of the reference algorithm
(AntonXue/nn-sdp: src/MyMath.jl, src/Qc/*.jl, src/Methods/chordal_cliques.jl,
src/Methods/chordal_sdp.jl) plus a CPU statement of the ADMM iteration that the
HIP library runs.  It is the *checker*: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path
(nn-sdp_amd/) never imports anything from here and fails loudly when the HIP
library is missing.

Parity status: the LMI assembly and clique construction are pinned against the
reference's only published numbers for this path (dump/scale/*.csv objective
values, 1e-3 relative; see tests/golden/dump_scale.csv and
tests/test_oracle_vs_dump.py).  The reference itself (Julia + MOSEK) cannot be
executed in this environment, so there are no reference-generated vectors.
"""
