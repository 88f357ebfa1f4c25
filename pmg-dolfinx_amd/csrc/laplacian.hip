// Matrix-free GLL-collocated Laplacian for gfx950: geometry tensor, the
// sum-factorised stiffness kernel, matrix-free diagonal, collocated load vector.
// Replaces src/laplacian.hpp (geometry_computation :22-113, stiffness_operator
// :143-278, MatFreeLaplacian :284-526) of the reference.
//
// HBM layout (all owned by the handle):
//   G       [ncells][3][N] double2 : (G00,G01) (G02,G11) (G12,G22) per quadrature
//           point, so that one wave reads 1 KiB contiguous per load instruction
//           (the reference stores [cell][q][6] AoS, stride 48 B per lane).
//   mdofmap [ncells][N] int32     : dof index with the Dirichlet flag folded into
//           the sign bit -- one load instead of a dofmap load plus a dependent
//           1-byte bc_marker gather per dof (src/laplacian.hpp:182-189).
//   D       [nd][nd] double       : 1-D derivative table, staged in LDS per block.
//
// Kernel shape: one thread per (cell, dof); CPB cells per workgroup so that every
// degree fills >= 4 wavefronts of 64 (the reference launches (P+1)^3 threads per
// block: 8 threads at P=1).  Element dofs and the three flux components live in
// LDS; the 1-D contractions read them conflict-free (stride-nd^2/nd/1 reads
// broadcast within a wave).  Scatter-add uses the hardware FP64 atomic
// (global_atomic_add_f64; library is built with -munsafe-fp-atomics, the
// reference is not and gets CAS loops, examples/pmg/CMakeLists.txt:64).
//
// Roofline: HBM-bound, AI 0.85 (P=1) .. 2.05 (P=8) flop/B; algorithmic bytes per
// cell 48N + 4N + 8 + 17U.
#include "common.hpp"

#include <cmath>

using namespace pmg;

struct pmg_laplacian_s
{
  pmg_layout layout = nullptr;
  int P = 0, nd = 0, N = 0;
  int32_t ncells = 0, npoints = 0;
  // caller-owned
  const double* kappa = nullptr;
  const int32_t* dofmap = nullptr;
  const double* xgeom = nullptr;
  const int32_t* geom_dofmap = nullptr;
  const int8_t* bc = nullptr;
  // owned
  double2* G = nullptr;
  int32_t* mdofmap = nullptr;
  double* D = nullptr;         // [nd*nd]
  double* dphi_geom = nullptr; // [3][N][8]
  double* gweights = nullptr;  // [N]
  int32_t* lcells = nullptr;   // nullptr = identity list 0..n_l-1
  int32_t* bcells = nullptr;
  int32_t n_l = 0, n_b = 0;
  double* diag_inv = nullptr; // [size_local + num_ghosts]
  bool have_diag = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  long long launches = 0; // stiffness-kernel launches since creation
};

namespace
{
template <int P>
struct Shape
{
  static constexpr int ND = P + 1;
  static constexpr int N = ND * ND * ND;
  // cells per workgroup: fill >= 4 waves
  static constexpr int CPB = P == 1 ? 32 : P == 2 ? 9 : P == 3 ? 4 : P == 4 ? 2 : 1;
  static constexpr int THREADS = ((CPB * N + 63) / 64) * 64;
};

// ---- geometry: J, adj(J), det at one quadrature point (src/laplacian.hpp:72-97) ----
__device__ inline void jacobian(const double* __restrict__ xgeom,
                                const int32_t* __restrict__ gdofs, const double* __restrict__ dphi,
                                int nq, int q, double K[3][3], double& detJ)
{
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int k = 0; k < 8; ++k)
  {
    const double* xk = xgeom + 3 * (size_t)gdofs[k];
    double x0 = xk[0], x1 = xk[1], x2 = xk[2];
    double d0 = dphi[(0 * nq + q) * 8 + k], d1 = dphi[(1 * nq + q) * 8 + k],
           d2 = dphi[(2 * nq + q) * 8 + k];
    J[0][0] += x0 * d0;
    J[0][1] += x0 * d1;
    J[0][2] += x0 * d2;
    J[1][0] += x1 * d0;
    J[1][1] += x1 * d1;
    J[1][2] += x1 * d2;
    J[2][0] += x2 * d0;
    J[2][1] += x2 * d1;
    J[2][2] += x2 * d2;
  }
  K[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  K[0][1] = -J[0][1] * J[2][2] + J[0][2] * J[2][1];
  K[0][2] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
  K[1][0] = -J[1][0] * J[2][2] + J[1][2] * J[2][0];
  K[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
  K[1][2] = -J[0][0] * J[1][2] + J[0][2] * J[1][0];
  K[2][0] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  K[2][1] = -J[0][0] * J[2][1] + J[0][1] * J[2][0];
  K[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  // full cofactor expansion along the first row (the reference's :97 drops to
  // the diagonal-J special case)
  detJ = J[0][0] * K[0][0] + J[0][1] * K[1][0] + J[0][2] * K[2][0];
}

// G for every (cell, q), paired layout
__global__ void geometry_kernel(int ncells, int nq, const double* __restrict__ xgeom,
                                const int32_t* __restrict__ geom_dofmap,
                                const double* __restrict__ dphi, const double* __restrict__ w,
                                double2* __restrict__ G)
{
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)ncells * nq)
    return;
  int c = (int)(gid / nq), q = (int)(gid - (long long)c * nq);
  double K[3][3], detJ;
  jacobian(xgeom, geom_dofmap + (size_t)c * 8, dphi, nq, q, K, detJ);
  double s = w[q] / detJ;
  double g0 = (K[0][0] * K[0][0] + K[0][1] * K[0][1] + K[0][2] * K[0][2]) * s;
  double g1 = (K[1][0] * K[0][0] + K[1][1] * K[0][1] + K[1][2] * K[0][2]) * s;
  double g2 = (K[2][0] * K[0][0] + K[2][1] * K[0][1] + K[2][2] * K[0][2]) * s;
  double g3 = (K[1][0] * K[1][0] + K[1][1] * K[1][1] + K[1][2] * K[1][2]) * s;
  double g4 = (K[2][0] * K[1][0] + K[2][1] * K[1][1] + K[2][2] * K[1][2]) * s;
  double g5 = (K[2][0] * K[2][0] + K[2][1] * K[2][1] + K[2][2] * K[2][2]) * s;
  double2* Gc = G + (size_t)c * 3 * nq;
  Gc[q] = make_double2(g0, g1);
  Gc[nq + q] = make_double2(g2, g3);
  Gc[2 * nq + q] = make_double2(g4, g5);
}

// paired layout -> the reference's [cell][q][6]
__global__ void geometry_export_kernel(int ncells, int nq, const double2* __restrict__ G,
                                       double* __restrict__ out)
{
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)ncells * nq)
    return;
  int c = (int)(gid / nq), q = (int)(gid - (long long)c * nq);
  const double2* Gc = G + (size_t)c * 3 * nq;
  double2 a = Gc[q], b = Gc[nq + q], d = Gc[2 * nq + q];
  double* o = out + (size_t)gid * 6;
  o[0] = a.x;
  o[1] = a.y;
  o[2] = b.x;
  o[3] = b.y;
  o[4] = d.x;
  o[5] = d.y;
}

__global__ void mask_dofmap_kernel(long long n, const int32_t* __restrict__ dofmap,
                                   const int8_t* __restrict__ bc, int32_t* __restrict__ out)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
  {
    int32_t d = dofmap[i];
    out[i] = bc[d] ? (d | (int32_t)0x80000000) : d;
  }
}

// ---- the hot kernel: y += kappa * B^T G B x over a list of cells ----
template <int P>
__global__ void __launch_bounds__(Shape<P>::THREADS)
    stiffness_kernel(const double* __restrict__ x, double* __restrict__ y,
                     const double2* __restrict__ G, const int32_t* __restrict__ mdofmap,
                     const double* __restrict__ kappa, const int32_t* __restrict__ cells,
                     int ncells_list, const double* __restrict__ Dg)
{
  constexpr int ND = Shape<P>::ND, N = Shape<P>::N, CPB = Shape<P>::CPB, NSQ = ND * ND;
  __shared__ double sD[ND * ND];
  __shared__ double su[CPB * N];
  __shared__ double sf0[CPB * N];
  __shared__ double sf1[CPB * N];
  __shared__ double sf2[CPB * N];

  const int t = threadIdx.x;
  const int lc = t / N;
  const int tl = t - lc * N;
  const int ci = blockIdx.x * CPB + lc;
  const bool active = (lc < CPB) && (ci < ncells_list);

  if (t < ND * ND)
    sD[t] = Dg[t];

  int dof = 0;
  bool is_bc = false;
  double xo = 0.0, kap = 0.0;
  double2 g01 = make_double2(0, 0), g23 = g01, g45 = g01;
  if (active)
  {
    const int cell = cells ? cells[ci] : ci;
    const int32_t m = mdofmap[(size_t)cell * N + tl];
    const double2* Gc = G + (size_t)cell * 3 * N;
    g01 = Gc[tl];
    g23 = Gc[N + tl];
    g45 = Gc[2 * N + tl];
    kap = kappa[cell];
    dof = m & 0x7fffffff;
    is_bc = m < 0;
    xo = x[dof];
    su[lc * N + tl] = is_bc ? 0.0 : xo; // src/laplacian.hpp:186-189
  }
  __syncthreads();

  const int a = tl / NSQ;
  const int b = (tl - a * NSQ) / ND;
  const int c = tl - a * NSQ - b * ND;
  double val = 0.0;
  if (active)
  {
    const double* u = su + lc * N;
    double vx = 0.0, vy = 0.0, vz = 0.0;
#pragma unroll
    for (int i = 0; i < ND; ++i)
    {
      vx += sD[a * ND + i] * u[i * NSQ + b * ND + c]; // :195-199
      vy += sD[b * ND + i] * u[a * NSQ + i * ND + c]; // :206-210
      vz += sD[c * ND + i] * u[a * NSQ + b * ND + i]; // :214-218
    }
    sf0[lc * N + tl] = kap * (g01.x * vx + g01.y * vy + g23.x * vz); // :233
    sf1[lc * N + tl] = kap * (g01.y * vx + g23.y * vy + g45.x * vz); // :234
    sf2[lc * N + tl] = kap * (g23.x * vx + g45.x * vy + g45.y * vz); // :235
  }
  __syncthreads();
  if (active)
  {
    const double* f0 = sf0 + lc * N;
    const double* f1 = sf1 + lc * N;
    const double* f2 = sf2 + lc * N;
    double wx = 0.0, wy = 0.0, wz = 0.0;
#pragma unroll
    for (int q = 0; q < ND; ++q)
    {
      wx += sD[q * ND + a] * f0[q * NSQ + b * ND + c]; // :246-251
      wy += sD[q * ND + b] * f1[a * NSQ + q * ND + c]; // :255-259
      wz += sD[q * ND + c] * f2[a * NSQ + b * ND + q]; // :263-267
    }
    val = wx + wy + wz; // :270
    if (is_bc)
      y[dof] = xo; // :273-274 (every sharing cell stores the same value)
    else
      atomicAdd(&y[dof], val); // :277, global_atomic_add_f64
  }
}

// ---- matrix-free diagonal (replaces the CSR detour of examples/pmg/main.cpp:274-279) ----
__global__ void diagonal_kernel(int ncells, int nd, const double2* __restrict__ G,
                                const int32_t* __restrict__ mdofmap,
                                const double* __restrict__ kappa, const double* __restrict__ D,
                                double* __restrict__ diag)
{
  const int N = nd * nd * nd, nsq = nd * nd;
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)ncells * N)
    return;
  int cell = (int)(gid / N), t = (int)(gid - (long long)cell * N);
  int a = t / nsq, b = (t - a * nsq) / nd, c = t - a * nsq - b * nd;
  const double2* Gc = G + (size_t)cell * 3 * N;
  double s = 0.0;
  for (int q = 0; q < nd; ++q)
  {
    double da = D[q * nd + a], db = D[q * nd + b], dc = D[q * nd + c];
    s += da * da * Gc[q * nsq + b * nd + c].x;         // G00 at (q,b,c)
    s += db * db * Gc[N + a * nsq + q * nd + c].y;     // G11 at (a,q,c)
    s += dc * dc * Gc[2 * N + a * nsq + b * nd + q].y; // G22 at (a,b,q)
  }
  double daa = D[a * nd + a], dbb = D[b * nd + b], dcc = D[c * nd + c];
  s += 2.0 * (Gc[t].y * daa * dbb + Gc[N + t].x * daa * dcc + Gc[2 * N + t].x * dbb * dcc);
  int32_t m = mdofmap[(size_t)cell * N + t];
  if (m >= 0)
    atomicAdd(&diag[m], kappa[cell] * s);
}

__global__ void diag_invert_kernel(int n, const int8_t* __restrict__ bc, double* __restrict__ d)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
  {
    double v = d[i];
    d[i] = bc[i] ? 1.0 : (v != 0.0 ? 1.0 / v : 0.0);
  }
}

// ---- GLL-collocated load vector ----
__global__ void rhs_kernel(int ncells, int nq, const double* __restrict__ xgeom,
                           const int32_t* __restrict__ geom_dofmap,
                           const double* __restrict__ dphi, const double* __restrict__ w,
                           const int32_t* __restrict__ mdofmap, const double* __restrict__ kappa,
                           const double* __restrict__ f, double* __restrict__ b)
{
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)ncells * nq)
    return;
  int c = (int)(gid / nq), q = (int)(gid - (long long)c * nq);
  int32_t m = mdofmap[gid];
  if (m < 0)
    return; // set_bc: b[bc] = 0 (b is zeroed first)
  double K[3][3], detJ;
  jacobian(xgeom, geom_dofmap + (size_t)c * 8, dphi, nq, q, K, detJ);
  atomicAdd(&b[m], kappa[c] * w[q] * detJ * f[m]);
}

template <int P>
int launch_stiffness(pmg_laplacian op, const double* x, double* y, const int32_t* cells, int n,
                     hipStream_t s)
{
  if (n <= 0)
    return PMG_OK;
  constexpr int CPB = Shape<P>::CPB;
  int grid = (n + CPB - 1) / CPB;
  stiffness_kernel<P><<<grid, Shape<P>::THREADS, 0, s>>>(x, y, op->G, op->mdofmap, op->kappa,
                                                         cells, n, op->D);
  op->launches++;
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

int dispatch_stiffness(pmg_laplacian op, const double* x, double* y, const int32_t* cells, int n,
                       hipStream_t s)
{
  switch (op->P)
  {
  case 1:
    return launch_stiffness<1>(op, x, y, cells, n, s);
  case 2:
    return launch_stiffness<2>(op, x, y, cells, n, s);
  case 3:
    return launch_stiffness<3>(op, x, y, cells, n, s);
  case 4:
    return launch_stiffness<4>(op, x, y, cells, n, s);
  case 5:
    return launch_stiffness<5>(op, x, y, cells, n, s);
  case 6:
    return launch_stiffness<6>(op, x, y, cells, n, s);
  case 7:
    return launch_stiffness<7>(op, x, y, cells, n, s);
  case 8:
    return launch_stiffness<8>(op, x, y, cells, n, s);
  default:
    return fail(PMG_ERR_INVALID, "Unsupported degree"); // src/laplacian.hpp:346,479
  }
}

template <typename T>
int upload(T** dst, const T* src, size_t n, hipStream_t s)
{
  PMG_HIP(hipMalloc(dst, sizeof(T) * (n ? n : 1)));
  if (n)
    PMG_HIP(hipMemcpyAsync(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice, s));
  return PMG_OK;
}
} // namespace

namespace pmg
{
// used by solvers.hip
int laplacian_apply(pmg_laplacian op, double* in, double* out, hipStream_t s);
const double* laplacian_diag_inv(pmg_laplacian op) { return op->diag_inv; }
pmg_layout laplacian_layout(pmg_laplacian op) { return op->layout; }
long long laplacian_launches(pmg_laplacian op) { return op->launches; }

// operator()(in, out), src/laplacian.hpp:462-482 + impl_operator :373-460
int laplacian_apply(pmg_laplacian op, double* in, double* out, hipStream_t s)
{
  pmg_layout l = op->layout;
  PMG_HIP(hipMemsetAsync(out, 0, sizeof(double) * l->total(), s)); // :466
  PMG_TRY(pmg_scatter_fwd_begin(l, in, (pmg_stream)s));           // :378
  PMG_TRY(dispatch_stiffness(op, in, out, op->lcells, op->n_l, s)); // :380-413
  PMG_TRY(pmg_scatter_fwd_end(l, in, (pmg_stream)s));             // :425
  PMG_TRY(dispatch_stiffness(op, in, out, op->bcells, op->n_b, s)); // :429-455
  return PMG_OK;
}
} // namespace pmg

extern "C" int pmg_laplacian_create_with_tables(
    pmg_laplacian* out, pmg_layout layout, int degree, int32_t ncells, const double* kappa,
    const int32_t* dofmap, const double* xgeom, int32_t npoints, const int32_t* geom_dofmap,
    const double* dphi_geometry, const double* G_weights, const int32_t* lcells, int32_t n_lcells,
    const int32_t* bcells, int32_t n_bcells, const int8_t* bc_marker, pmg_stream stream)
{
  PMG_REQUIRE(out && layout, "pmg_laplacian_create: NULL handle");
  if (degree < 1 || degree > PMG_MAX_DEGREE)
    return fail(PMG_ERR_INVALID, "Unsupported degree"); // src/laplacian.hpp:346
  PMG_REQUIRE(ncells >= 0 && n_lcells >= 0 && n_bcells >= 0 && n_lcells + n_bcells <= ncells,
              "pmg_laplacian_create: cell lists (%d + %d) exceed ncells (%d)", n_lcells, n_bcells,
              ncells);
  PMG_REQUIRE(ncells == 0 || (kappa && dofmap && xgeom && geom_dofmap && bc_marker),
              "pmg_laplacian_create: NULL array");
  PMG_REQUIRE((n_lcells == 0 || lcells) && (n_bcells == 0 || bcells),
              "pmg_laplacian_create: NULL cell list");
  for (int i = 0; i < n_lcells; ++i)
    PMG_REQUIRE(lcells[i] >= 0 && lcells[i] < ncells, "pmg_laplacian_create: lcells[%d] = %d out of range", i, lcells[i]);
  for (int i = 0; i < n_bcells; ++i)
    PMG_REQUIRE(bcells[i] >= 0 && bcells[i] < ncells, "pmg_laplacian_create: bcells[%d] = %d out of range", i, bcells[i]);

  hipStream_t s = S(stream);
  auto* op = new pmg_laplacian_s;
  op->layout = layout;
  op->P = degree;
  op->nd = degree + 1;
  op->N = op->nd * op->nd * op->nd;
  op->ncells = ncells;
  op->npoints = npoints;
  op->kappa = kappa;
  op->dofmap = dofmap;
  op->xgeom = xgeom;
  op->geom_dofmap = geom_dofmap;
  op->bc = bc_marker;
  const int nd = op->nd, N = op->N;

  // 1-D tables (basix's job in the reference, src/laplacian.hpp:302-317)
  std::vector<double> pts(nd), wts(nd), D(nd * nd);
  gll_table(nd, pts.data(), wts.data());
  lagrange_derivative_table(nd, pts.data(), D.data());
  PMG_TRY(upload(&op->D, D.data(), D.size(), s));

  // trilinear coordinate-element derivatives at the GLL points, [3][N][8], and
  // the 3-D weights (examples/pmg/main.cpp:216-238)
  if (dphi_geometry && G_weights)
  {
    PMG_HIP(hipMalloc(&op->dphi_geom, sizeof(double) * 24 * N));
    PMG_HIP(hipMalloc(&op->gweights, sizeof(double) * N));
    PMG_HIP(hipMemcpyAsync(op->dphi_geom, dphi_geometry, sizeof(double) * 24 * N,
                           hipMemcpyDeviceToDevice, s));
    PMG_HIP(hipMemcpyAsync(op->gweights, G_weights, sizeof(double) * N, hipMemcpyDeviceToDevice, s));
  }
  else
  {
    std::vector<double> dphi(24 * (size_t)N), w3(N);
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nd; ++b)
        for (int c = 0; c < nd; ++c)
        {
          int q = (a * nd + b) * nd + c;
          w3[q] = wts[a] * wts[b] * wts[c];
          double ph[3][2] = {{1.0 - pts[a], pts[a]}, {1.0 - pts[b], pts[b]}, {1.0 - pts[c], pts[c]}};
          const double dp[2] = {-1.0, 1.0};
          for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
              for (int k = 0; k < 2; ++k)
              {
                int v = i * 4 + j * 2 + k;
                dphi[((size_t)0 * N + q) * 8 + v] = dp[i] * ph[1][j] * ph[2][k];
                dphi[((size_t)1 * N + q) * 8 + v] = ph[0][i] * dp[j] * ph[2][k];
                dphi[((size_t)2 * N + q) * 8 + v] = ph[0][i] * ph[1][j] * dp[k];
              }
        }
    PMG_TRY(upload(&op->dphi_geom, dphi.data(), dphi.size(), s));
    PMG_TRY(upload(&op->gweights, w3.data(), w3.size(), s));
    PMG_HIP(hipStreamSynchronize(s)); // host vectors go out of scope
  }

  // cell lists; an identity list (single rank: every cell is "local") needs no indirection
  bool identity = (n_lcells == ncells);
  for (int i = 0; identity && i < n_lcells; ++i)
    identity = (lcells[i] == i);
  op->n_l = n_lcells;
  op->n_b = n_bcells;
  if (!identity && n_lcells > 0)
    PMG_TRY(upload(&op->lcells, lcells, n_lcells, s));
  if (n_bcells > 0)
    PMG_TRY(upload(&op->bcells, bcells, n_bcells, s));

  const long long nq_total = (long long)ncells * N;
  PMG_HIP(hipMalloc(&op->G, sizeof(double2) * 3 * (nq_total ? nq_total : 1)));
  PMG_HIP(hipMalloc(&op->mdofmap, sizeof(int32_t) * (nq_total ? nq_total : 1)));
  PMG_HIP(hipMalloc(&op->diag_inv, sizeof(double) * (layout->total() ? layout->total() : 1)));
  PMG_HIP(hipEventCreate(&op->ev0));
  PMG_HIP(hipEventCreate(&op->ev1));
  if (nq_total > 0)
  {
    int blocks = (int)((nq_total + 255) / 256);
    geometry_kernel<<<blocks, 256, 0, s>>>(ncells, N, xgeom, geom_dofmap, op->dphi_geom,
                                          op->gweights, op->G);
    mask_dofmap_kernel<<<blocks > 4096 ? 4096 : blocks, 256, 0, s>>>(nq_total, dofmap, bc_marker,
                                                                   op->mdofmap);
    PMG_HIP(hipGetLastError());
  }
  PMG_HIP(hipStreamSynchronize(s)); // host cell lists may be freed by the caller
  *out = op;
  return PMG_OK;
}

extern "C" int pmg_laplacian_create(pmg_laplacian* out, pmg_layout layout, int degree,
                                    int32_t ncells, const double* kappa, const int32_t* dofmap,
                                    const double* xgeom, int32_t npoints,
                                    const int32_t* geom_dofmap, const int32_t* lcells,
                                    int32_t n_lcells, const int32_t* bcells, int32_t n_bcells,
                                    const int8_t* bc_marker, pmg_stream stream)
{
  return pmg_laplacian_create_with_tables(out, layout, degree, ncells, kappa, dofmap, xgeom,
                                          npoints, geom_dofmap, nullptr, nullptr, lcells, n_lcells,
                                          bcells, n_bcells, bc_marker, stream);
}

extern "C" int pmg_laplacian_destroy(pmg_laplacian op)
{
  if (!op)
    return PMG_OK;
  (void)hipFree(op->G);
  (void)hipFree(op->mdofmap);
  (void)hipFree(op->D);
  (void)hipFree(op->dphi_geom);
  (void)hipFree(op->gweights);
  (void)hipFree(op->lcells);
  (void)hipFree(op->bcells);
  (void)hipFree(op->diag_inv);
  if (op->ev0)
    (void)hipEventDestroy(op->ev0);
  if (op->ev1)
    (void)hipEventDestroy(op->ev1);
  delete op;
  return PMG_OK;
}

extern "C" int pmg_laplacian_degree(pmg_laplacian op) { return op ? op->P : -1; }

extern "C" int pmg_laplacian_apply(pmg_laplacian op, double* in, double* out, pmg_stream stream)
{
  PMG_REQUIRE(op && in && out, "pmg_laplacian_apply: NULL argument");
  PMG_REQUIRE(in != out, "pmg_laplacian_apply: in and out alias");
  return laplacian_apply(op, in, out, S(stream));
}

extern "C" int pmg_laplacian_get_diag_inverse(pmg_laplacian op, double* diag_inv,
                                              pmg_stream stream)
{
  PMG_REQUIRE(op && diag_inv, "pmg_laplacian_get_diag_inverse: NULL argument");
  PMG_REQUIRE(op->have_diag, "pmg_laplacian_get_diag_inverse: diagonal not set");
  PMG_HIP(hipMemcpyAsync(diag_inv, op->diag_inv, sizeof(double) * op->layout->total(),
                         hipMemcpyDeviceToDevice, S(stream)));
  return PMG_OK;
}

extern "C" int pmg_laplacian_set_diag_inverse(pmg_laplacian op, const double* diag_inv,
                                              pmg_stream stream)
{
  PMG_REQUIRE(op && diag_inv, "pmg_laplacian_set_diag_inverse: NULL argument");
  PMG_HIP(hipMemcpyAsync(op->diag_inv, diag_inv, sizeof(double) * op->layout->total(),
                         hipMemcpyDeviceToDevice, S(stream)));
  op->have_diag = true;
  return PMG_OK;
}

extern "C" int pmg_laplacian_compute_diag_inverse(pmg_laplacian op, pmg_stream stream)
{
  PMG_REQUIRE(op, "pmg_laplacian_compute_diag_inverse: NULL argument");
  hipStream_t s = S(stream);
  const int total = op->layout->total();
  PMG_HIP(hipMemsetAsync(op->diag_inv, 0, sizeof(double) * total, s));
  const long long n = (long long)op->ncells * op->N;
  if (n > 0)
    diagonal_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(op->ncells, op->nd, op->G, op->mdofmap,
                                                          op->kappa, op->D, op->diag_inv);
  if (total > 0)
    diag_invert_kernel<<<(total + 255) / 256, 256, 0, s>>>(total, op->bc, op->diag_inv);
  PMG_HIP(hipGetLastError());
  op->have_diag = true;
  return PMG_OK;
}

extern "C" int pmg_laplacian_get_geometry(pmg_laplacian op, double* G_out, pmg_stream stream)
{
  PMG_REQUIRE(op && G_out, "pmg_laplacian_get_geometry: NULL argument");
  const long long n = (long long)op->ncells * op->N;
  if (n > 0)
    geometry_export_kernel<<<(int)((n + 255) / 256), 256, 0, S(stream)>>>(op->ncells, op->N, op->G,
                                                                         G_out);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_laplacian_assemble_rhs(pmg_laplacian op, const double* f, double* b,
                                          pmg_stream stream)
{
  PMG_REQUIRE(op && f && b, "pmg_laplacian_assemble_rhs: NULL argument");
  hipStream_t s = S(stream);
  PMG_HIP(hipMemsetAsync(b, 0, sizeof(double) * op->layout->total(), s));
  const long long n = (long long)op->ncells * op->N;
  if (n > 0)
    rhs_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(op->ncells, op->N, op->xgeom,
                                                     op->geom_dofmap, op->dphi_geom, op->gweights,
                                                     op->mdofmap, op->kappa, f, b);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_laplacian_time_kernel(pmg_laplacian op, const double* in, double* out, int reps,
                                         double* ms_per_launch, pmg_stream stream)
{
  PMG_REQUIRE(op && in && out && ms_per_launch && reps > 0, "pmg_laplacian_time_kernel: bad argument");
  hipStream_t s = S(stream);
  // every local cell in one launch each, exactly the kernel the apply issues
  PMG_HIP(hipEventRecord(op->ev0, s));
  for (int r = 0; r < reps; ++r)
  {
    PMG_TRY(dispatch_stiffness(op, in, out, op->lcells, op->n_l, s));
    PMG_TRY(dispatch_stiffness(op, in, out, op->bcells, op->n_b, s));
  }
  PMG_HIP(hipEventRecord(op->ev1, s));
  PMG_HIP(hipEventSynchronize(op->ev1));
  float ms = 0.f;
  PMG_HIP(hipEventElapsedTime(&ms, op->ev0, op->ev1));
  *ms_per_launch = (double)ms / reps;
  return PMG_OK;
}
