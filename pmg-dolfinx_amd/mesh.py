"""Structured box mesh, brick partition with one ghost-cell layer, tensor-product
GLL dof numbering -- the host-side set-up the reference obtains from dolfinx.

Stands in for ``mesh::create_box`` + ``ghost_layer_mesh`` (``src/mesh.hpp:16-98``),
``fem::create_functionspace`` (``examples/pmg/main.cpp:83-87``),
``compute_boundary_cells`` (``src/mesh.hpp:105-143``) and the
``IndexMap``/``Scatterer`` index lists ``acc::Vector`` is built from
(``src/vector.hpp:83-96``).  Pure numpy, no GPU, no communication: every rank
derives its bricks, its neighbours' bricks and both sides of every halo list from
the geometry alone.

Partition rule: the n_x x n_y x n_z cell grid is cut into
p_x x p_y x p_z bricks, rank = (r_x p_y + r_y) p_z + r_z.  A rank owns the dofs of
its brick except those on an interface with a lower brick ("lower rank owns the
interface"), and additionally holds every cell that shares a vertex with its
brick (the ghost layer), so every owned row of the operator is complete locally
and only the forward (owner -> ghost) halo is needed (``src/mesh.hpp:11-12``).

Local numbering: owned dofs first, lexicographic (x slowest) in the owned dof box;
then ghosts grouped by owner rank, lexicographic inside each owner's box -- the
receive buffer is therefore already in ghost order.  Cells: owned cells first
(lexicographic), then ghost cells.  Cell-local dof index t = a*nd^2 + b*nd + c,
x slowest (``src/laplacian.hpp:173``).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def split_bounds(n: int, parts: int):
    """Balanced contiguous split of range(n) into ``parts`` pieces."""
    return [(n * i) // parts for i in range(parts + 1)]


def rank_to_coords(rank: int, dims):
    px, py, pz = dims
    return (rank // (py * pz), (rank // pz) % py, rank % pz)


def coords_to_rank(c, dims):
    return (c[0] * dims[1] + c[1]) * dims[2] + c[2]


def default_proc_dims(size: int):
    """2x2x2-style factorisation: spread factors over z, y, x in turn
    (1 -> 1x1x1, 2 -> 1x1x2, 4 -> 1x2x2, 8 -> 2x2x2)."""
    dims = [1, 1, 1]
    ax = 2
    n = size
    f = 2
    while n > 1:
        while n % f:
            f += 1
        dims[ax] *= f
        n //= f
        ax = (ax - 1) % 3
    return tuple(dims)


def _interval_intersect(a, b):
    lo, hi = max(a[0], b[0]), min(a[1], b[1])
    return (lo, hi) if lo <= hi else None


def _box_points(box):
    """Lexicographic (x slowest) integer points of a closed box [(lo,hi)]*3."""
    rng = [np.arange(lo, hi + 1) for lo, hi in box]
    X, Y, Z = np.meshgrid(*rng, indexing="ij")
    return X.ravel(), Y.ravel(), Z.ravel()


def basix_node_permutation(P: int) -> np.ndarray:
    """perm1d[j] = ascending position of 1-D node j of a basix interval element: vertex 0, vertex 1, then the
    interior nodes from left to right (the order of dolfinx's tensor-product dofmaps and of basix's GLL points)."""
    nd = P + 1
    return np.array([0, nd - 1] + list(range(1, nd - 1)), dtype=np.int32)[:nd]


def cell_permutation(perm1d) -> np.ndarray:
    """perm3[t_caller] = t_ascending for t = ja*nd^2 + jb*nd + jc."""
    p = np.asarray(perm1d, dtype=np.int64)
    nd = p.size
    return ((p[:, None, None] * nd + p[None, :, None]) * nd + p[None, None, :]).ravel().astype(np.int32)


def dofmap_in_node_order(dofmap: np.ndarray, perm1d) -> np.ndarray:
    """The ascending dofmap [ncells, nd^3] as a caller with cell-local node order ``perm1d`` would hold it
    (what dolfinx hands the reference for perm1d = basix_node_permutation(P))."""
    return np.ascontiguousarray(dofmap[:, cell_permutation(perm1d)])


@dataclass
class LevelData:
    """Everything one p-level needs on one rank (host arrays)."""

    P: int
    size_local: int
    num_ghosts: int
    dofmap: np.ndarray  # [ncells_local, (P+1)^3] int32, local dof indices
    bc_marker: np.ndarray  # [size_local + num_ghosts] int8
    local_to_global: np.ndarray  # [size_local + num_ghosts] int64
    ghost_owners: np.ndarray  # [num_ghosts] int32
    neighbors: list  # ranks exchanged with, ascending
    send_counts: list  # per neighbour
    recv_counts: list
    send_indices: np.ndarray  # owned local indices, grouped by neighbour (== Scatterer::local_indices)
    recv_indices: np.ndarray  # ghost positions (relative to size_local), grouped by neighbour
    lcells: np.ndarray  # cells touching no ghost dof
    bcells: np.ndarray  # cells touching a ghost dof + all ghost cells
    dof_coords: np.ndarray = field(default=None, repr=False)  # [ndofs_local, 3] physical

    @property
    def ndofs(self):
        return self.size_local + self.num_ghosts


class BoxPartition:
    """One rank's share of an n_x x n_y x n_z hex mesh of the box [lo, hi]."""

    def __init__(self, n, proc_dims=(1, 1, 1), rank=0, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0), warp=None):
        if np.isscalar(n):
            n = (int(n),) * 3
        self.n = tuple(int(v) for v in n)
        self.dims = tuple(int(v) for v in proc_dims)
        self.size = self.dims[0] * self.dims[1] * self.dims[2]
        if not (0 <= rank < self.size):
            raise ValueError("rank out of range")
        for a in range(3):
            if self.dims[a] > self.n[a]:
                raise ValueError("more bricks than cells along an axis")
        self.rank = int(rank)
        self.coords = rank_to_coords(rank, self.dims)
        self.lo, self.hi, self.warp = lo, hi, warp
        self._bounds = [split_bounds(self.n[a], self.dims[a]) for a in range(3)]
        self.own = self._own_cells(self.coords)
        self.ext = self._ext_cells(self.coords)

        # local cells: owned (lexicographic) then ghosts (lexicographic over the rest of ext)
        ex = [np.arange(l, h) for l, h in self.ext]
        CX, CY, CZ = np.meshgrid(*ex, indexing="ij")
        cx, cy, cz = CX.ravel(), CY.ravel(), CZ.ravel()
        owned = np.ones(cx.shape, dtype=bool)
        for c, (l, h) in zip((cx, cy, cz), self.own):
            owned &= (c >= l) & (c < h)
        order = np.concatenate([np.nonzero(owned)[0], np.nonzero(~owned)[0]])
        self.cell_coords = np.stack([cx[order], cy[order], cz[order]], axis=1)
        self.ncells_owned = int(owned.sum())
        self.ncells = int(order.size)

        # geometry: vertices of the extended brick, lexicographic
        vshape = [h - l + 1 for l, h in self.ext]
        g = [np.linspace(lo[a], hi[a], self.n[a] + 1)[self.ext[a][0]: self.ext[a][1] + 1] for a in range(3)]
        X, Y, Z = np.meshgrid(*g, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        if warp is not None:
            x = warp(x)
        self.xgeom = np.ascontiguousarray(x, dtype=np.float64)
        lc = self.cell_coords - np.array([self.ext[a][0] for a in range(3)])
        gd = np.empty((self.ncells, 8), dtype=np.int32)
        for i in range(2):
            for j in range(2):
                for k in range(2):
                    gd[:, i * 4 + j * 2 + k] = ((lc[:, 0] + i) * vshape[1] + (lc[:, 1] + j)) * vshape[2] + lc[:, 2] + k
        self.geom_dofmap = gd
        self._levels = {}

    # ---- brick geometry (cells) ----
    def _own_cells(self, coords):
        return [(self._bounds[a][coords[a]], self._bounds[a][coords[a] + 1]) for a in range(3)]

    def _ext_cells(self, coords):
        own = self._own_cells(coords)
        return [(max(own[a][0] - 1, 0), min(own[a][1] + 1, self.n[a])) for a in range(3)]

    # ---- dof boxes at degree P (closed intervals of global dof coordinates) ----
    def _owned_dof_box(self, coords, P):
        own = self._own_cells(coords)
        return [(own[a][0] * P + (1 if coords[a] > 0 else 0), own[a][1] * P) for a in range(3)]

    def _local_dof_box(self, coords, P):
        ext = self._ext_cells(coords)
        return [(ext[a][0] * P, ext[a][1] * P) for a in range(3)]

    def global_dof_shape(self, P):
        return tuple(self.n[a] * P + 1 for a in range(3))

    def global_ndofs(self, P):
        s = self.global_dof_shape(P)
        return s[0] * s[1] * s[2]

    def level(self, P: int) -> LevelData:
        if P in self._levels:
            return self._levels[P]
        nd = P + 1
        G = self.global_dof_shape(P)
        Bme = self._local_dof_box(self.coords, P)
        Ome = self._owned_dof_box(self.coords, P)
        bshape = [h - l + 1 for l, h in Bme]
        blo = np.array([l for l, _ in Bme])
        lid = np.full(bshape, -1, dtype=np.int64)

        def box_slices(box):
            return tuple(slice(box[a][0] - blo[a], box[a][1] - blo[a] + 1) for a in range(3))

        oshape = [h - l + 1 for l, h in Ome]
        size_local = int(np.prod(oshape))
        lid[box_slices(Ome)] = np.arange(size_local).reshape(oshape)

        neighbors, send_counts, recv_counts, send_lists, ghost_owner = [], [], [], [], []
        nghost = 0
        for q in range(self.size):
            if q == self.rank:
                continue
            qc = rank_to_coords(q, self.dims)
            Oq = self._owned_dof_box(qc, P)
            Bq = self._local_dof_box(qc, P)
            rbox = [_interval_intersect(Oq[a], Bme[a]) for a in range(3)]
            sbox = [_interval_intersect(Ome[a], Bq[a]) for a in range(3)]
            nrecv = 0 if any(b is None for b in rbox) else int(np.prod([h - l + 1 for l, h in rbox]))
            nsend = 0 if any(b is None for b in sbox) else int(np.prod([h - l + 1 for l, h in sbox]))
            if nrecv == 0 and nsend == 0:
                continue
            neighbors.append(q)
            recv_counts.append(nrecv)
            send_counts.append(nsend)
            if nrecv:
                rs = [h - l + 1 for l, h in rbox]
                lid[box_slices(rbox)] = size_local + nghost + np.arange(nrecv).reshape(rs)
                ghost_owner.append(np.full(nrecv, q, dtype=np.int32))
                nghost += nrecv
            if nsend:
                send_lists.append(lid[box_slices(sbox)].ravel().copy())
        if (lid < 0).any():
            raise RuntimeError("ghost layer reaches a dof with no owner among the ranks (brick too thin?)")
        num_ghosts = nghost

        # cell dofmap through the box lookup
        a = np.arange(nd)
        cc = self.cell_coords
        ix = cc[:, 0:1] * P - blo[0] + a[None, :]
        iy = cc[:, 1:2] * P - blo[1] + a[None, :]
        iz = cc[:, 2:3] * P - blo[2] + a[None, :]
        dm = lid[ix[:, :, None, None], iy[:, None, :, None], iz[:, None, None, :]]
        dofmap = np.ascontiguousarray(dm.reshape(self.ncells, nd**3), dtype=np.int32)

        # local -> global, boundary marker, coordinates
        bx, by, bz = _box_points(Bme)
        l2g = np.empty(size_local + num_ghosts, dtype=np.int64)
        flat = lid.ravel()
        l2g[flat] = (bx * G[1] + by) * G[2] + bz
        onb = (bx == 0) | (bx == G[0] - 1) | (by == 0) | (by == G[1] - 1) | (bz == 0) | (bz == G[2] - 1)
        bc = np.zeros(size_local + num_ghosts, dtype=np.int8)
        bc[flat] = onb.astype(np.int8)

        ghost_cell = np.arange(self.ncells) >= self.ncells_owned
        touches_ghost = (dofmap >= size_local).any(axis=1)
        mark = ghost_cell | touches_ghost  # src/mesh.hpp:119-128
        lcells = np.nonzero(~mark)[0].astype(np.int32)
        bcells = np.nonzero(mark)[0].astype(np.int32)

        lv = LevelData(
            P=P,
            size_local=size_local,
            num_ghosts=num_ghosts,
            dofmap=dofmap,
            bc_marker=bc,
            local_to_global=l2g,
            ghost_owners=np.concatenate(ghost_owner) if ghost_owner else np.zeros(0, dtype=np.int32),
            neighbors=neighbors,
            send_counts=send_counts,
            recv_counts=recv_counts,
            send_indices=(np.concatenate(send_lists) if send_lists else np.zeros(0, dtype=np.int64)).astype(np.int32),
            recv_indices=np.arange(num_ghosts, dtype=np.int32),
            lcells=lcells,
            bcells=bcells,
        )
        self._levels[P] = lv
        return lv

    def dof_coordinates(self, P: int) -> np.ndarray:
        """Physical coordinates of the local dofs [ndofs_local, 3]: trilinear image
        of the reference GLL nodes of each cell."""
        from ._tables import gll_points

        lv = self.level(P)
        if lv.dof_coords is not None:
            return lv.dof_coords
        nd = P + 1
        xi = gll_points(nd)
        phi = np.stack([1.0 - xi, xi], axis=1)
        N = np.einsum("ai,bj,cl->abcijl", phi, phi, phi).reshape(nd**3, 8)
        out = np.zeros((lv.ndofs, 3))
        step = max(1, 4_000_000 // (nd**3))
        for s in range(0, self.ncells, step):
            xc = self.xgeom[self.geom_dofmap[s: s + step]]
            pts = np.einsum("tk,ckd->ctd", N, xc)
            out[lv.dofmap[s: s + step].ravel()] = pts.reshape(-1, 3)
        lv.dof_coords = out
        return out
