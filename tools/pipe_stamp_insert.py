"""(diagnostic helper) insert / strip the wall-clock stamp calls of the -DNNSDP_STAMPS build in nn-sdp_amd/csrc/refine_pipe.hpp.
The shipped source carries no stamp calls; `python tools/pipe_stamp_insert.py insert` writes refine_pipe.hpp with them in place
(entry, loads issued, loads landed, chain done, epilogue done, stores drained), `strip` removes them again."""
import re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, "nn-sdp_amd", "csrc", "refine_pipe.hpp")
s = open(p).read()
def strip(s):
    s = re.sub(r"  PST\(\d, 0\)\n", "", s)
    s = re.sub(r"  PST\(\d, 1\)\n  PWAIT\(\);\n  PST\(\d, 2\)\n  d4_t c = pipe_chain<KSQ>\(av, bv, ksq\);\n  \{ double c0_ = c\[0\]; PDEP\(c0_\); c\[0\] = c0_; \}\n  PST\(\d, 3\)\n", "  const d4_t c = pipe_chain<KSQ>(av, bv, ksq);\n", s)
    return re.sub(r"\n  PST\(\d, 4\)\n  PWAIT\(\);\n  PST\(\d, 5\)", "", s)
s = strip(s)
if sys.argv[1:] == ["insert"]:
    for ki, kn in enumerate(["k_pipe_T", "k_pipe_B", "k_pipe_X", "k_pipe_V", "k_pipe_W"]):
        a = s.index("void %s(PipeArgs a) {" % kn)
        b = s.index("\n}\n", a)
        body = s[a:b]
        body = body.replace("  if (b < 0) return;\n", "  if (b < 0) return;\n  PST(%d, 0)\n" % ki, 1)
        body = body.replace("  const d4_t c = pipe_chain<KSQ>(av, bv, ksq);\n", "  PST(%d, 1)\n  PWAIT();\n  PST(%d, 2)\n  d4_t c = pipe_chain<KSQ>(av, bv, ksq);\n  { double c0_ = c[0]; PDEP(c0_); c[0] = c0_; }\n  PST(%d, 3)\n" % (ki, ki, ki), 1)
        body += "\n  PST(%d, 4)\n  PWAIT();\n  PST(%d, 5)" % (ki, ki)
        s = s[:a] + body + s[b:]
open(p, "w").write(s)
