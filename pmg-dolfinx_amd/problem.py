"""The driver-side wiring of ``examples/pmg/main.cpp:solve`` (``:41-380``) for a
structured box: per level function space data, operator, inverse diagonal, load
vector, eigenvalue estimate, Chebyshev smoother; the interpolators and the
V-cycle.  Generalised from the example's two levels {1, 3} to any ascending list
of degrees (BASELINE config 2 is {1, 2, 4})."""
from __future__ import annotations

import numpy as np

from .cg import CGSolver
from .chebyshev import Chebyshev
from .interpolate import Interpolator
from .laplacian import MatFreeLaplacian
from .mesh import BoxPartition, basix_node_permutation, default_proc_dims, dofmap_in_node_order
from .pmg import MultigridPreconditioner
from .vector import Layout, Vector


def make_layout(lv, group=None, device="cuda", comm=None) -> Layout:
    return Layout(lv.size_local, lv.num_ghosts, lv.neighbors, lv.send_counts, lv.recv_counts, lv.send_indices,
                  lv.recv_indices, group=group, device=device, comm=comm)


class PoissonHierarchy:
    def __init__(self, n, orders=(1, 2, 4), kappa=2.0, cheb_its=3, proc_dims=None, rank=0, size=1, group=None,
                 warp=None, eig_cg_its=20, eig_cg_rtol=1e-6, freq=(2, 3, 4), device="cuda", comm=None,
                 node_order="ascending", level_hook=None):
        import torch

        self.orders = tuple(int(p) for p in orders)
        if list(self.orders) != sorted(set(self.orders)):
            raise ValueError("orders must be strictly ascending (coarse -> fine)")
        dims = tuple(proc_dims) if proc_dims is not None else default_proc_dims(size)
        self.part = BoxPartition(n, dims, rank, warp=warp)
        part = self.part
        self.levels, self.layouts, self.operators, self.smoothers, self.eig_ranges = [], [], [], [], []
        self.rhs = []
        dev = torch.device(device)
        # geometry is shared by all levels (examples/pmg/main.cpp:243-254)
        self.xgeom = torch.from_numpy(part.xgeom).to(dev)
        self.geom_dofmap = torch.from_numpy(part.geom_dofmap).to(dev)
        self.kappa = torch.full((part.ncells,), float(kappa), dtype=torch.float64, device=dev)  # :190-193
        # node_order = "basix": the dofmaps are handed over as dolfinx would hold them (endpoints first per direction)
        self.node_order = node_order
        for P in self.orders:
            lv = part.level(P)
            if level_hook is not None:  # bench.py --corrupt-halo: a deliberately wrong halo plan for the gate's own test
                level_hook(lv)
            layout = make_layout(lv, group, device, comm)
            dofmap = lv.dofmap
            if node_order in ("basix", "endpoints_first"):
                dofmap = dofmap_in_node_order(lv.dofmap, basix_node_permutation(P))
            elif node_order != "ascending":
                raise ValueError("PoissonHierarchy: node_order is 'ascending' or 'basix'")
            op = MatFreeLaplacian(P, self.kappa, dofmap, self.xgeom, self.geom_dofmap, lv.lcells, lv.bcells,
                                  lv.bc_marker, layout, node_order=node_order)  # :270-272
            op.compute_diag_inverse()  # replaces :274-279
            self.levels.append(lv)
            self.layouts.append(layout)
            self.operators.append(op)
            # load vector, :289-300 with f of examples/pmg/poisson.py:6-8,30
            c = part.dof_coordinates(P)
            kx, ky, kz = freq
            fvals = ((kx * kx + ky * ky + kz * kz) * np.pi**2 * np.sin(kx * np.pi * c[:, 0])
                     * np.sin(ky * np.pi * c[:, 1]) * np.sin(kz * np.pi * c[:, 2]))
            f = Vector(layout)
            f.data.copy_(torch.from_numpy(fvals))
            b = Vector(layout)
            op.assemble_rhs(f, b)
            self.rhs.append(b)
            del f
        # smoothers, :306-330
        for i, P in enumerate(self.orders):
            layout, op = self.layouts[i], self.operators[i]
            cg = CGSolver(layout)
            cg.set_max_iterations(eig_cg_its)
            cg.set_tolerance(eig_cg_rtol)
            cg.store_coefficients(True)
            x = Vector(layout)
            y = Vector(layout)
            x.set(0.0)
            y.set(1.0)
            cg.solve(op, x, y)
            eig = np.sort(cg.compute_eigenvalues())
            rng = (0.1 * eig[-1], 1.1 * eig[-1])  # :327
            sm = Chebyshev(layout, rng)
            sm.set_max_iterations(cheb_its)
            self.smoothers.append(sm)
            self.eig_ranges.append(rng)
            del cg, x, y
        # interpolators, :336-341
        self.interpolators = []
        for i in range(len(self.orders) - 1):
            lc, lf = self.levels[i], self.levels[i + 1]
            self.interpolators.append(
                Interpolator(self.orders[i], self.orders[i + 1], self.operators[i].dofmap,
                             self.operators[i + 1].dofmap, lf.lcells, lf.bcells, self.layouts[i],
                             self.layouts[i + 1], fine_operator=self.operators[i + 1], node_order=node_order))
        # V-cycle, :348-355
        self.mg = MultigridPreconditioner(self.layouts, self.levels[0].bc_marker)
        self.mg.set_solvers(self.smoothers)
        self.mg.set_operators(self.operators)
        self.mg.set_interpolators(self.interpolators)

    @property
    def fine_ndofs_owned(self):
        return self.levels[-1].size_local

    def new_vector(self, level=-1):
        return Vector(self.layouts[level])
