"""nnsdp_amd: MI355X-native Chordal-DeepSDP LMI assembler + ADMM solver (host-side mirror of the
reference's Methods/Qc API over the C ABI in include/nnsdp.h)."""
from .methods import (  # noqa: F401
    FeedFwdNet, QcInputBox, QcSafety, QcReachHplane, QcReachCircle, QcReachEllipsoid,
    QcActivBounded, QcActivSector, SafetyQuery, ReachQuery, AdmmSdpOptions, QuerySolution,
    SingleDecomp, DoubleDecomp, DoubleRelaxDecomp, PathDecomp, AutoDecomp, DenseCone, Solver, SolverBatch,
    runQuery, runQueries, solveQuery, makeZ, adjoint, makeCliques, project_psd_batched, project_psd_warm, comm_unique_id, shardPlan,
)
from .frontend import (  # noqa: F401
    read_nnet, evalFeedFwdNet, evalFeedFwdNetBatch, sampleTrajs, randomNetwork, makeIntervalsInfo, makeQcActivs, approxEllipsoid,
    findEllipsoid, findCircle, findReach2Dpoly, write_scale_csv, runScale, ellipsoidQuery,
)
from . import _lib  # noqa: F401
from . import vnnlib  # noqa: F401
from .vnnlib import (  # noqa: F401
    read_vnnlib, hplaneS, loadVnnlibCnf, loadReluQueriesCnf, verifyAcasSpec, verifyPairs, isSolutionGood, shardPairs, reachForm, safetyFromReach,
)

__version__ = "0.2.0"
