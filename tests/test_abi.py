"""The C-ABI library on a machine without a GPU: it loads, exports every symbol
include/pmg_amd.h declares, and its host-only entry points (tables, TQLI, error
reporting) agree with the oracle / the reference fixture.  No compute calls."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from oracle import pmg_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _declared():
    src = open(os.path.join(ROOT, "include", "pmg_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(pmg_[a-z0-9_]+)\s*\(", src))
    return sorted(names - {"pmg_exchange_fn", "pmg_allreduce_fn"})


def test_every_declared_symbol_is_exported_and_bound(built):
    import pmg_dolfinx_amd as pm

    L = C.CDLL(pm._lib.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(L, name), f"{name} declared in pmg_amd.h but not exported"
    assert sorted(pm._lib.exported_symbols()) == declared  # the Python binding covers the whole ABI
    assert pm._lib.lib().pmg_version() >= 100


def test_host_tables_match_oracle(built):
    from pmg_dolfinx_amd import _tables as t

    for n in range(2, 10):
        x, w = t.gll_points_weights(n)
        xo, wo = po.gll_points_weights(n)
        assert np.abs(x - xo).max() < 1e-15 and np.abs(w - wo).max() < 1e-15
        D, Do = t.lagrange_derivative_table(n), po.lagrange_deriv_matrix(xo)
        assert np.abs(D - Do).max() < 1e-13 * np.abs(Do).max()
    for pc, pf in ((1, 2), (2, 4), (1, 3), (3, 6), (4, 8)):
        assert np.abs(t.interpolation_table(pc, pf) - po.interpolation_matrix_1d(pc, pf)).max() < 1e-14


def test_tqli_golden_and_errors(built):
    import pmg_dolfinx_amd as pm
    from pmg_dolfinx_amd import _tables as t

    cases = json.load(open(os.path.join(HERE, "golden", "tqli_golden.json")))["cases"]
    for case in cases:
        d = t.tqli(case["d"], case["e"])
        assert np.allclose(d, case["tqli_d_out"], rtol=1e-13, atol=0)
    L = pm._lib.lib()
    x = np.zeros(1)
    rc = L.pmg_gll_table(1, x.ctypes.data_as(pm._lib.c_dp), x.ctypes.data_as(pm._lib.c_dp))
    assert rc == -1 and b"n" in L.pmg_last_error()


def test_missing_library_fails_loudly(built, monkeypatch):
    import pmg_dolfinx_amd as pm

    monkeypatch.setattr(pm._lib, "_lib", None)
    monkeypatch.setattr(pm._lib, "LIB_PATH", "/nonexistent/libpmg_amd.so")
    try:
        pm._lib.lib()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")


def test_cpp_adapter_drivers_build_and_fail_loudly_without_gpu(built):
    """include/pmg_amd.hpp compiles (build() does that for the two drivers of
    examples/); without a GPU a driver stops with the library's error, it has no
    other path to fall back to."""
    import subprocess

    import torch

    root = os.path.dirname(HERE)
    for name in ("mat_free_main", "pmg_main", "selftest_main"):
        exe = os.path.join(root, "pmg-dolfinx_amd", "bin", name)
        assert os.path.exists(exe)
        if name != "selftest_main":
            assert subprocess.run([exe, "--help"], capture_output=True, timeout=60).returncode == 0
    if not torch.cuda.is_available():
        r = subprocess.run([os.path.join(root, "pmg-dolfinx_amd", "bin", "mat_free_main"), "--n", "2"],
                           capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "error:" in r.stderr


def test_halo_window_plan_is_host_arithmetic(built):
    """pmg_layout_window_describe (no GPU involved): every neighbour's segment starts on a 256-byte boundary, the
    ghost -> owner region follows the owner -> ghost region, two slots; more neighbours than the flag block serves are
    refused with a message."""
    import ctypes as C

    import numpy as np

    from pmg_dolfinx_amd import _lib

    send = np.array([37, 1001, 64, 0], dtype=np.int32)
    recv = np.array([5, 33, 0, 700], dtype=np.int32)
    doubles, fwd, rev = C.c_int64(), np.zeros(4, np.int64), np.zeros(4, np.int64)
    _lib.call("pmg_layout_window_describe", 4, send.ctypes.data_as(_lib.c_ip), recv.ctypes.data_as(_lib.c_ip),
              C.byref(doubles), fwd.ctypes.data_as(_lib.c_lp), rev.ctypes.data_as(_lib.c_lp))
    pad = lambda n: (n + 31) // 32 * 32  # noqa: E731 -- 32 doubles = 256 bytes
    rlen = sum(pad(int(c)) for c in recv)
    slen = sum(pad(int(c)) for c in send)
    assert fwd.tolist() == [0, 32, 96, 96] and all(o % 32 == 0 for o in fwd)
    assert rev.tolist() == [rlen, rlen + 64, rlen + 64 + 1024, rlen + 64 + 1024 + 64]
    assert doubles.value == 2 * (rlen + slen)
    many = np.ones(65, dtype=np.int32)
    big = np.zeros(65, np.int64)
    with pytest.raises(RuntimeError, match="at most 64 neighbours"):
        _lib.call("pmg_layout_window_describe", 65, many.ctypes.data_as(_lib.c_ip), many.ctypes.data_as(_lib.c_ip),
                  C.byref(doubles), big.ctypes.data_as(_lib.c_lp), big.ctypes.data_as(_lib.c_lp))


def test_node_order_tables_are_host_arithmetic(built):
    """pmg_node_permutation and the *_ordered tables (no GPU involved): the endpoints-first order is basix's
    (vertex 0, vertex 1, interior left to right; the tables the reference takes from basix,
    src/laplacian.hpp:302-317, src/interpolate.hpp:118, are these rows and columns of the ascending ones)."""
    import pmg_dolfinx_amd as pm
    from pmg_dolfinx_amd import _lib

    ip, dp = _lib.c_ip, _lib.c_dp
    for P in range(1, 9):
        n = P + 1
        perm = pm.node_permutation("basix", P)
        assert perm.tolist() == ([0, n - 1] + list(range(1, n - 1)))[:n]
        assert pm.node_permutation("ascending", P).tolist() == list(range(n))
        assert np.array_equal(perm, pm.basix_node_permutation(P))
        xo, wo = po.gll_points_weights(n)
        x, w, D = np.zeros(n), np.zeros(n), np.zeros((n, n))
        _lib.call("pmg_gll_table_ordered", n, 1, None, x.ctypes.data_as(dp), w.ctypes.data_as(dp))
        assert x[0] == 0.0 and x[min(1, n - 1)] == 1.0 and np.all(np.diff(x[2:]) > 0)
        assert np.abs(x - xo[perm]).max() < 1e-15 and np.abs(w - wo[perm]).max() < 1e-15
        _lib.call("pmg_lagrange_derivative_table_ordered", n, 1, None, D.ctypes.data_as(dp))
        Do = po.lagrange_deriv_matrix(xo)
        assert np.abs(D - Do[np.ix_(perm, perm)]).max() < 1e-13 * np.abs(Do).max()
        # ... which is the derivative table of the Lagrange basis on the PERMUTED nodes, from first principles
        assert np.abs(D - po.lagrange_deriv_matrix(xo[perm])).max() < 1e-11 * np.abs(Do).max()
    for pc, pf in ((1, 2), (2, 4), (3, 6), (4, 8)):
        M = np.zeros((pf + 1, pc + 1))
        _lib.call("pmg_interpolation_table_ordered", pc, pf, 1, None, None, M.ctypes.data_as(dp))
        Mo = po.interpolation_matrix_1d(pc, pf)
        assert np.abs(M - Mo[np.ix_(pm.basix_node_permutation(pf), pm.basix_node_permutation(pc))]).max() < 1e-14
    custom = np.array([2, 0, 1], dtype=np.int32)
    out = np.zeros(3, dtype=np.int32)
    _lib.call("pmg_node_permutation", 2, 2, custom.ctypes.data_as(ip), out.ctypes.data_as(ip))
    assert out.tolist() == [2, 0, 1]
    bad = np.array([2, 2, 1], dtype=np.int32)
    with pytest.raises(RuntimeError, match="not a permutation"):
        _lib.call("pmg_node_permutation", 2, 2, bad.ctypes.data_as(ip), out.ctypes.data_as(ip))
    with pytest.raises(RuntimeError, match="unknown order"):
        _lib.call("pmg_node_permutation", 9, 2, None, out.ctypes.data_as(ip))
