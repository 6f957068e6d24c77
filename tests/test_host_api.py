"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol declared in
include/nnsdp.h, index arithmetic (makeCliques) through the ABI equals the oracle, argument
validation follows the header's error convention, and compute entry points fail loudly (no CPU
fallback) when no device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
import nnsdp_amd as na
from nnsdp_amd import _lib
from oracle import qc


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(helpers.ROOT, "include", "nnsdp.h")).read()
    declared = set(re.findall(r"\b(nnsdp_[a-z_0-9A-Z]+)\s*\(", hdr))
    assert len(declared) >= 18
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)
    m = re.search(r"#define NNSDP_VERSION (\d+)", hdr)
    assert lib.nnsdp_version() == int(m.group(1))


def test_struct_layout_matches_header_field_count():
    hdr = open(os.path.join(helpers.ROOT, "include", "nnsdp.h")).read()
    for struct, cls in (("nnsdp_problem", _lib.Problem), ("nnsdp_options", _lib.Options), ("nnsdp_result", _lib.Result)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        nfields = len([ln for ln in body.split(";") if ln.strip()])
        assert nfields == len(cls._fields_), (struct, nfields, len(cls._fields_))


@pytest.mark.parametrize("xdims,beta", [([2] + [10] * 5 + [2], 0), ([2] + [10] * 5 + [2], 3), ([3, 7, 9, 8, 6, 7, 2], 2),
                                         ([2] + [40] * 20 + [2], 7), ([5] + [50] * 6 + [5], 1)])
def test_make_cliques_matches_oracle(xdims, beta):
    from oracle.nnet_io import random_net
    net = random_net(xdims, seed=1)
    for mode, omode in ((na.SingleDecomp, "single"), (na.DoubleDecomp, "double"), (na.DenseCone, "dense"), (na.PathDecomp, "path")):
        assert na.makeCliques(xdims, beta, mode) == qc.clique_index_sets(net, beta, omode)


def test_problem_dims_and_defaults():
    lib = _lib.load()
    d = helpers.load_problem("W10-D5", 3)
    cp = na.methods._CProblem(helpers.product_query(d))
    z, ac, n2, ng = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.nnsdp_problem_dims(C.byref(cp.p), C.byref(z), C.byref(ac), C.byref(n2), C.byref(ng)) == 0
    assert (z.value, ac.value, n2.value, ng.value) == (53, 50, 294, 347)      # SURVEY.md section 8 table (beta=3: 347)
    o = _lib.Options()
    lib.nnsdp_default_options(C.byref(o))
    assert o.decomp_mode == 1 and 0 < o.alpha < 2 and o.eps_rel == 1e-6
    assert lib.nnsdp_status_string(0) == b"OPTIMAL" and lib.nnsdp_status_string(3) == b"SLOW_PROGRESS"


def test_invalid_arguments_return_negative_codes():
    lib = _lib.load()
    assert lib.nnsdp_problem_dims(None, None, None, None, None) < 0
    assert b"null" in lib.nnsdp_last_error()
    xd = np.asarray([2, 3], dtype=np.int32)
    n, t = C.c_int32(), C.c_int32()
    assert lib.nnsdp_make_cliques(1, xd.ctypes.data_as(_lib.c_int32_p), 0, 1, C.byref(n), C.byref(t), None, None) < 0   # K < 2
    xd = np.asarray([2, 3, 3, 2], dtype=np.int32)
    assert lib.nnsdp_make_cliques(3, xd.ctypes.data_as(_lib.c_int32_p), -1, 1, C.byref(n), C.byref(t), None, None) < 0  # beta < 0
    assert lib.nnsdp_make_cliques(3, xd.ctypes.data_as(_lib.c_int32_p), 0, 9, C.byref(n), C.byref(t), None, None) < 0   # mode
    assert [len(c) for c in na.makeCliques([5] + [50] * 6 + [5], 0, na.PathDecomp)] == [56] + [101] * 5   # 2W+1, not 3W+1
    with pytest.raises(_lib.NnsdpError):
        na.project_psd_batched([np.zeros((4097, 4097))])        # n > 4096 (checked before any device use)
    with pytest.raises(ValueError):
        na.project_psd_batched([np.zeros((3, 4))])
    # mirror-side validation, like the reference's @assert in the QC constructors
    with pytest.raises(AssertionError):
        na.QcActivBounded(acymin=[1.0], acymax=[0.0])            # activ_bounded.jl:8
    with pytest.raises(AssertionError):
        na.QcActivSector(acxdim=2, beta=-1, smin=[0, 0], smax=[1, 1])   # activ_sector.jl:11
    with pytest.raises(AssertionError):
        na.FeedFwdNet(xdims=[2, 3], Ms=[np.zeros((3, 3))])      # MyNeuralNetwork.jl:18


@pytest.mark.skipif(_has_gpu(), reason="GPU present: the no-device failure path cannot be observed")
def test_compute_entry_points_fail_loudly_without_gpu():
    d = helpers.load_problem("W10-D5", 0)
    q = helpers.product_query(d)
    for fn in (lambda: na.runQuery(q, na.AdmmSdpOptions(max_iters=10)),
               lambda: na.makeZ(q, np.zeros(203)),
               lambda: na.project_psd_batched([np.eye(3)])):
        with pytest.raises(_lib.NnsdpError) as ei:
            fn()
        assert "no CPU fallback" in str(ei.value) and ei.value.code > 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(helpers.ROOT, "nn-sdp_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "oracle/" not in src or f.endswith(".md"), f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """the boundary from the other side: include/nnsdp.h compiles as C99 (-pedantic) and the host-only entry points run from
    a C program linked against the library (what a cgo / ccall / JNI binding does)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib = _lib.LIB_PATH
    if not os.path.exists(lib):
        pytest.fail("libnnsdp_hip.so is not built")
    exe = str(tmp_path / "abi_check")
    src = os.path.join(helpers.ROOT, "tests", "c_abi", "abi_check.c")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(helpers.ROOT, "include"), src, "-o", exe, lib,
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout, r.stderr)
