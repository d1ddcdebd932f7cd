"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/ilqr_hip.h declares;
host-only entry points (dims, defaults, URDF reader) work without a GPU.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests.helpers import ROOT, panda_segs, urdf_text


def _lib():
    from ilqr_planner_amd import capi

    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return capi


def test_every_declared_symbol_is_exported():
    capi = _lib()
    hdr = open(os.path.join(ROOT, "include", "ilqr_hip.h")).read()
    declared = set(re.findall(r"\b(ilqr_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    L = C.CDLL(capi.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"declared in ilqr_hip.h but not exported: {missing}"
    assert declared == set(capi.EXPORTS), (declared ^ set(capi.EXPORTS))


def test_dims_and_defaults():
    capi = _lib()
    L = capi.load()
    d = capi.ProblemDesc()
    L.ilqr_desc_defaults(C.byref(d))
    assert (d.reg, d.alpha_floor, d.stop_tol) == (1e-6, 1e-3, 1e-3)
    dims = capi.Dims()
    for kind, nd, exp in ((0, 1, (7, 7, 7, 6)), (0, 2, (14, 7, 14, 12)), (1, 1, (8, 8, 8, 7)), (1, 2, (15, 8, 15, 13))):
        d.kind, d.nb_deriv, d.dof = kind, nd, 7
        assert L.ilqr_dims_of(C.byref(d), C.byref(dims)) == 0
        assert (dims.n_x, dims.n_u, dims.n_f, dims.n_Q) == exp  # SURVEY.md 8 dimension key
    d.kind = 7
    assert L.ilqr_dims_of(C.byref(d), C.byref(dims)) != 0


def test_urdf_reader_matches_independent_reader():
    capi = _lib()
    a = capi.chain_from_urdf(urdf_text(), "panda_link0", "panda_tip", [0.1, -0.2, 0.3], [0.01, 0.02, 0.03])
    b = panda_segs([0.1, -0.2, 0.3], [0.01, 0.02, 0.03])
    assert a["dof"] == b["dof"] == 7 and a["seg_joint"] == b["seg_joint"]
    np.testing.assert_allclose(a["seg_xyz"], b["seg_xyz"], atol=0)
    np.testing.assert_allclose(a["seg_R"], b["seg_R"], atol=1e-16)
    mov = [i for i, j in enumerate(a["seg_joint"]) if j >= 0]
    np.testing.assert_allclose(np.array(a["seg_axis"])[mov], np.array(b["seg_axis"])[mov], atol=0)
    np.testing.assert_allclose(a["lower"], b["lower"])
    np.testing.assert_allclose(a["upper"], b["upper"])


def test_urdf_reader_errors():
    capi = _lib()
    with pytest.raises(RuntimeError, match=r"\[KDLRobot\] Unable to build kinematic chain from nope to panda_tip"):
        capi.chain_from_urdf(urdf_text(), "nope", "panda_tip")
    with pytest.raises(RuntimeError, match="parse error"):
        capi.chain_from_urdf("<robot><joint name='a'></robot>", "a", "b")
    with pytest.raises(RuntimeError, match="expected <robot>"):
        capi.chain_from_urdf("<?xml version='1.0'?><!-- c --><thing/>", "a", "b")


def test_context_creation_fails_loudly_without_gpu():
    capi = _lib()
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.Context(0)
