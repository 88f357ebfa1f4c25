"""ctypes front-end of the C/OpenMP oracle (``oracle/pmg_oracle.c``).

TEST INFRASTRUCTURE ONLY (see the header of ``pmg_oracle.py``)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import pmg_oracle as po

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_int8)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "pmg_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_level_create.restype = C.c_void_p
        L.orc_level_create.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _dp, _dp, _bp, _dp]
        L.orc_interp_create.restype = C.c_void_p
        L.orc_interp_create.argtypes = [C.c_void_p, C.c_void_p, _dp]
        for f in ("orc_level_destroy", "orc_interp_destroy"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = None
        L.orc_level_apply.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_level_diagonal.argtypes = [C.c_void_p, _dp]
        L.orc_level_set_diag_inverse.argtypes = [C.c_void_p, _dp]
        L.orc_geometry.argtypes = [C.c_int, C.c_int, _dp, _ip, _dp, _dp, _dp]
        L.orc_cheb_solve.argtypes = [C.c_void_p, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int]
        L.orc_prolong.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_restrict.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_tqli.argtypes = [_dp, _dp, C.c_int]
        L.orc_tqli.restype = C.c_int
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads.restype = None
        L.orc_vcycle.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _dp, C.c_int] + [
            C.POINTER(_dp)
        ] * 5 + [_dp, _dp]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n: int):
    lib().orc_set_num_threads(int(n))


def cpu_share() -> int:
    """Host cores this process may really use: the affinity mask capped by the cgroup's CPU quota (a container
    on a 128-thread host with a 16-CPU quota is throttled, not faster, with 128 OpenMP threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def tqli(d, e):
    d = np.ascontiguousarray(d, dtype=np.float64).copy()
    e = np.ascontiguousarray(e, dtype=np.float64).copy()
    rc = lib().orc_tqli(_d(d), _d(e), len(d))
    return rc, d


class CLevel:
    """One p-level: geometry, operator, diagonal (C restatement)."""

    def __init__(self, P, kappa, dofmap, xgeom, geom_dofmap, bc_marker):
        L = lib()
        self.P = P
        self.dofmap = np.ascontiguousarray(dofmap, dtype=np.int32)
        self.ncells = self.dofmap.shape[0]
        self.bc = np.ascontiguousarray(bc_marker, dtype=np.int8)
        self.ndofs = self.bc.shape[0]
        self.kappa = np.ascontiguousarray(np.broadcast_to(np.asarray(kappa, dtype=np.float64), (self.ncells,)))
        nodes, _ = po.gll_points_weights(P + 1)
        self.D = np.ascontiguousarray(po.lagrange_deriv_matrix(nodes))
        dphi, w3 = po.geometry_tables(P)
        nq = (P + 1) ** 3
        self.G = np.empty((self.ncells, nq, 6))
        xg = np.ascontiguousarray(xgeom, dtype=np.float64)
        gd = np.ascontiguousarray(geom_dofmap, dtype=np.int32)
        L.orc_geometry(self.ncells, nq, _d(xg), gd.ctypes.data_as(_ip), _d(dphi), _d(w3), _d(self.G))
        self.h = L.orc_level_create(
            P, self.ncells, self.ndofs, self.dofmap.ctypes.data_as(_ip), _d(self.G), _d(self.kappa),
            self.bc.ctypes.data_as(_bp), _d(self.D))
        if not self.h:
            raise RuntimeError("orc_level_create failed")
        d = self.diagonal()
        self.dinv = 1.0 / d
        L.orc_level_set_diag_inverse(self.h, _d(self.dinv))

    def apply(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.ndofs)
        lib().orc_level_apply(self.h, _d(x), _d(y))
        return y

    def diagonal(self):
        d = np.empty(self.ndofs)
        lib().orc_level_diagonal(self.h, _d(d))
        return d

    def diag_inverse(self):
        return self.dinv

    def cheb_solve(self, lmax, k, x, b, need_r=True, x_zero=False):
        r = np.empty(self.ndofs)
        z = np.empty(self.ndofs)
        q = np.empty(self.ndofs)
        b = np.ascontiguousarray(b, dtype=np.float64)
        lib().orc_cheb_solve(self.h, float(lmax), int(k), _d(x), _d(b), _d(r), _d(z), _d(q), int(need_r), int(x_zero))
        return x, r

    def __del__(self):
        try:
            lib().orc_level_destroy(self.h)
        except Exception:
            pass


class CInterp:
    def __init__(self, lc: CLevel, lf: CLevel):
        self.lc, self.lf = lc, lf
        self.M1 = np.ascontiguousarray(po.interpolation_matrix_1d(lc.P, lf.P))
        self.h = lib().orc_interp_create(lc.h, lf.h, _d(self.M1))

    def interpolate(self, coarse):
        fine = np.zeros(self.lf.ndofs)
        lib().orc_prolong(self.h, _d(np.ascontiguousarray(coarse)), _d(fine))
        return fine

    def reverse_interpolate(self, fine):
        coarse = np.empty(self.lc.ndofs)
        lib().orc_restrict(self.h, _d(np.ascontiguousarray(fine)), _d(coarse))
        return coarse

    def __del__(self):
        try:
            lib().orc_interp_destroy(self.h)
        except Exception:
            pass


class CMultigrid:
    """Lean V-cycle in C (same algorithm the GPU product runs)."""

    def __init__(self, levels, interps, lmax, cheb_k):
        self.levels, self.interps = levels, interps
        self.lmax = np.ascontiguousarray(lmax, dtype=np.float64)
        self.k = int(cheb_k)
        n = len(levels)
        self._w = [[np.zeros(l.ndofs) for l in levels] for _ in range(5)]
        self._lv = (C.c_void_p * n)(*[l.h for l in levels])
        self._ip = (C.c_void_p * max(1, n - 1))(*[i.h for i in interps])
        self._wp = [(_dp * n)(*[_d(a) for a in ws]) for ws in self._w]

    def apply(self, rhs, y):
        """y is the initial guess and is overwritten with the result."""
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        lib().orc_vcycle(len(self.levels), self._lv, self._ip, _d(self.lmax), self.k, *self._wp, _d(rhs), _d(y))
        return y
