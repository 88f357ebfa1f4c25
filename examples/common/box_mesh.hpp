// Structured hex mesh of a box and its GLL Lagrange function spaces, single rank:
// what the reference's drivers get from dolfinx (create_box, create_functionspace,
// locate_dofs_topological; examples/pmg/main.cpp:425-470, :196-240).  Cells, dofs
// and vertices are numbered lexicographically (x slowest); the dofs of a cell are
// ordered t = a*nd^2 + b*nd + c like the kernel's thread index (src/laplacian.hpp:173).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace examples
{
struct BoxMesh
{
  int n;                                  // cells per direction
  std::vector<double> xgeom;              // [npoints][3]
  std::vector<std::int32_t> geom_dofmap;  // [ncells][8], k = i*4 + j*2 + l
  std::int32_t ncells() const { return n * n * n; }
  std::int32_t npoints() const { return (n + 1) * (n + 1) * (n + 1); }

  explicit BoxMesh(int n_) : n(n_)
  {
    const int nv = n + 1;
    xgeom.resize((std::size_t)3 * nv * nv * nv);
    for (int i = 0; i < nv; ++i)
      for (int j = 0; j < nv; ++j)
        for (int k = 0; k < nv; ++k)
        {
          const std::size_t v = ((std::size_t)i * nv + j) * nv + k;
          xgeom[3 * v + 0] = (double)i / n;
          xgeom[3 * v + 1] = (double)j / n;
          xgeom[3 * v + 2] = (double)k / n;
        }
    geom_dofmap.resize((std::size_t)8 * n * n * n);
    for (int cx = 0; cx < n; ++cx)
      for (int cy = 0; cy < n; ++cy)
        for (int cz = 0; cz < n; ++cz)
        {
          const std::size_t c = ((std::size_t)cx * n + cy) * n + cz;
          for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
              for (int l = 0; l < 2; ++l)
                geom_dofmap[8 * c + i * 4 + j * 2 + l] = ((cx + i) * nv + (cy + j)) * nv + cz + l;
        }
  }
};

/// Degree-P space on the mesh: dofmap, Dirichlet marker on the whole boundary
/// (examples/mat_free/main.cpp:163-165,236-240), dof coordinates.
struct FunctionSpace
{
  int degree;
  std::int32_t ndofs;
  std::vector<std::int32_t> dofmap;   // [ncells][(P+1)^3]
  std::vector<std::int8_t> bc_marker; // [ndofs]
  std::vector<double> x;              // [ndofs][3]

  /// gll: the P+1 GLL points on [0,1] (pmg_gll_table).
  FunctionSpace(const BoxMesh& mesh, int P, const std::vector<double>& gll) : degree(P)
  {
    const int n = mesh.n, nd = P + 1, m = n * P + 1;
    ndofs = m * m * m;
    dofmap.resize((std::size_t)mesh.ncells() * nd * nd * nd);
    for (int cx = 0; cx < n; ++cx)
      for (int cy = 0; cy < n; ++cy)
        for (int cz = 0; cz < n; ++cz)
        {
          const std::size_t c = ((std::size_t)cx * n + cy) * n + cz;
          std::int32_t* d = dofmap.data() + c * nd * nd * nd;
          for (int a = 0; a < nd; ++a)
            for (int b = 0; b < nd; ++b)
              for (int e = 0; e < nd; ++e)
                d[(a * nd + b) * nd + e] = ((cx * P + a) * m + (cy * P + b)) * m + cz * P + e;
        }
    bc_marker.assign(ndofs, 0);
    x.resize((std::size_t)3 * ndofs);
    std::vector<double> line(m);
    for (int c = 0; c < n; ++c)
      for (int a = 0; a < nd; ++a)
        line[c * P + a] = (c + gll[a]) / n;
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j)
        for (int k = 0; k < m; ++k)
        {
          const std::size_t d = ((std::size_t)i * m + j) * m + k;
          x[3 * d + 0] = line[i];
          x[3 * d + 1] = line[j];
          x[3 * d + 2] = line[k];
          if (i == 0 || j == 0 || k == 0 || i == m - 1 || j == m - 1 || k == m - 1)
            bc_marker[d] = 1;
        }
  }
};

/// Cells per direction so that the degree-P space has about `ndofs` dofs
/// (the drivers' --ndofs, examples/pmg/main.cpp:425-440).
inline int cells_for_ndofs(std::size_t ndofs, int P)
{
  const double m = std::cbrt((double)ndofs);
  const int n = (int)std::lround((m - 1.0) / P);
  return n < 1 ? 1 : n;
}
} // namespace examples
