"""1-D GLL tables from the library's host-side entry points (no GPU needed)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _dp(a):
    return a.ctypes.data_as(_lib.c_dp)


def gll_points_weights(n: int):
    x = np.empty(n)
    w = np.empty(n)
    _lib.call("pmg_gll_table", int(n), _dp(x), _dp(w))
    return x, w


def gll_points(n: int):
    return gll_points_weights(n)[0]


def lagrange_derivative_table(n: int):
    D = np.empty((n, n))
    _lib.call("pmg_lagrange_derivative_table", int(n), _dp(D))
    return D


def interpolation_table(p_coarse: int, p_fine: int):
    M = np.empty((p_fine + 1, p_coarse + 1))
    _lib.call("pmg_interpolation_table", int(p_coarse), int(p_fine), _dp(M))
    return M


def tqli(d, e):
    d = np.ascontiguousarray(d, dtype=np.float64).copy()
    e = np.ascontiguousarray(e, dtype=np.float64).copy()
    _lib.call("pmg_tqli", _dp(d), _dp(e), len(d))
    return d
