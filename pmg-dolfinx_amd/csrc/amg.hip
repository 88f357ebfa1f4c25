// Algebraic multigrid for the coarsest (degree 1) level of the p-multigrid hierarchy.
//
// The reference solves its coarsest level with PETSc's KSPCG (at most 60 iterations) preconditioned
// by hypre BoomerAMG on an assembled aijhipsparse matrix (src/amg.hpp:33-47, wired in at
// src/pmg.hpp:106-107 through examples/pmg/main.cpp:331-335).  hypre and PETSc are third-party
// arithmetic that is not part of this repository; this file is the library's own solver for that
// slot, written for the same inputs the degree-1 operator already has:
//
//   set-up (host, once): the degree-1 stiffness matrix is assembled from the operator's geometry
//     tensor (trilinear hexahedra with the 2-point GLL rule: a 7..27-point stencil); smoothed
//     aggregation (Vanek, Mandel, Brezina 1996) builds the hierarchy: strength graph
//     |a_ij| >= theta sqrt(a_ii a_jj), greedy aggregation, piecewise-constant tentative
//     prolongator, one damped-Jacobi smoothing step P = (I - 4/(3 rho) D^-1 A) T, Galerkin
//     product P^T A P; the last level is inverted densely.  Dirichlet rows (identity rows of the
//     operator) stay out of the coarse spaces.
//   cycle (device): V(k, k) with the same 4th-kind Chebyshev / Jacobi smoother as the p-levels
//     (cheb_iterate), lean form (the pre-smoother leaves the residual, the post-smoother skips its
//     last product), CSR products one sub-wavefront per row.  Nothing synchronises the host.
//   solve: either a fixed number of cycles from a zero initial guess (a fixed linear operator: the
//     p-multigrid cycle stays a valid CG preconditioner and can be captured in a graph), or the
//     reference's shape -- CG on the matrix-free operator preconditioned by one cycle, to a
//     relative tolerance, at most max_iter iterations.
//
// Several ranks, two forms.  Replicated (pmg_amg_create_replicated, what the harness uses): the owned
// rows are gathered once, with global column indices, into the global degree-1 matrix on every rank
// (coarse-grid agglomeration); a solve is ONE all-reduce of the zero-padded right-hand side followed by
// the single-rank solve of the whole coarse problem on every rank -- identical arithmetic everywhere, no
// communication inside, the same iteration counts as on one rank.  The degree-1 level is 1/64 of the
// p = 4 dofs, so the redundant work stays small next to a fine-level smooth.  Rank-local
// (pmg_amg_create on a distributed operator): the hierarchy of the rank's owned block as a block
// preconditioner of a distributed Krylov solve (ghost couplings dropped from the preconditioner only):
// no gather, but 9 -> 31 -> 38 -> 40 CG iterations on 1 -> 2 -> 4 -> 8 ranks (tools/amg_rank_scaling.py).
//
// Parity: unpinned by the reference (third-party arithmetic, no fixture).  The tests compare the
// device cycle with a numpy restatement of the same hierarchy (oracle/amg_oracle.py) and check the
// solver's defining properties (symmetry, contraction, h-independent iteration counts).
#include "common.hpp"
#include "patches.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <numeric>
#include <random>
#include <thread>

using namespace pmg;

namespace pmg
{
int laplacian_apply(pmg_laplacian op, double* in, double* out, hipStream_t s);
const double* laplacian_diag_inv(pmg_laplacian op);
pmg_layout laplacian_layout(pmg_laplacian op);
PatchView laplacian_patches(pmg_laplacian op);
struct LaplacianInputs
{
  int degree;
  int32_t ncells;
  const int32_t* dofmap; // device
  const int8_t* bc;      // device
  const double* kappa;   // device
};
LaplacianInputs laplacian_inputs(pmg_laplacian op);
} // namespace pmg

namespace
{
struct StageTimer
{
  bool on = std::getenv("PMG_AMG_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void lap(const char* what)
  {
    if (!on)
      return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[pmg_amg set-up] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

struct HostCsr
{
  int n = 0, m = 0; // rows, columns
  std::vector<int> rp, ci;
  std::vector<double> v;
  long long nnz() const { return (long long)ci.size(); }
};

struct DevCsr
{
  int n = 0, m = 0, tpr = 4; // threads per row of the product kernels
  int* rp = nullptr;
  int* ci = nullptr;
  double* v = nullptr;
};

template <typename T>
int to_device(T** dst, const std::vector<T>& src)
{
  PMG_HIP(hipMalloc(dst, sizeof(T) * std::max<size_t>(src.size(), 1)));
  if (!src.empty())
    PMG_HIP(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
  return PMG_OK;
}

int upload_csr(DevCsr& d, const HostCsr& h)
{
  d.n = h.n;
  d.m = h.m;
  const double avg = h.n ? (double)h.nnz() / h.n : 1.0;
  d.tpr = avg > 48 ? 32 : avg > 24 ? 16 : avg > 10 ? 8 : avg > 4 ? 4 : 2;
  PMG_TRY(to_device(&d.rp, h.rp));
  PMG_TRY(to_device(&d.ci, h.ci));
  PMG_TRY(to_device(&d.v, h.v));
  return PMG_OK;
}

void free_csr(DevCsr& d)
{
  (void)hipFree(d.rp);
  (void)hipFree(d.ci);
  (void)hipFree(d.v);
  d = DevCsr();
}

// ---- device products: TPR lanes of a wavefront share a row ------------------------------------
// MODE 0: y = A x    1: y = b - A x    2: y += A x
template <int TPR, int MODE>
__global__ void csr_product_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                   const double* __restrict__ v, const double* __restrict__ x,
                                   const double* __restrict__ b, double* __restrict__ y)
{
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = gid / TPR, sub = gid % TPR;
  double acc = 0.0;
  if (row < n)
  {
    const int e = rp[row + 1];
    for (int k = rp[row] + sub; k < e; k += TPR)
      acc += v[k] * x[ci[k]];
  }
#pragma unroll
  for (int off = TPR / 2; off > 0; off >>= 1)
    acc += __shfl_down(acc, off, TPR);
  if (row < n && sub == 0)
  {
    if (MODE == 0)
      y[row] = acc;
    else if (MODE == 1)
      y[row] = b[row] - acc;
    else
      y[row] += acc;
  }
}

template <int MODE>
int csr_product(const DevCsr& A, const double* x, const double* b, double* y, hipStream_t s)
{
  if (A.n == 0)
    return PMG_OK;
  const long long threads = (long long)A.n * A.tpr;
  const int blocks = (int)((threads + 255) / 256);
  switch (A.tpr)
  {
  case 2:
    csr_product_kernel<2, MODE><<<blocks, 256, 0, s>>>(A.n, A.rp, A.ci, A.v, x, b, y);
    break;
  case 4:
    csr_product_kernel<4, MODE><<<blocks, 256, 0, s>>>(A.n, A.rp, A.ci, A.v, x, b, y);
    break;
  case 8:
    csr_product_kernel<8, MODE><<<blocks, 256, 0, s>>>(A.n, A.rp, A.ci, A.v, x, b, y);
    break;
  case 16:
    csr_product_kernel<16, MODE><<<blocks, 256, 0, s>>>(A.n, A.rp, A.ci, A.v, x, b, y);
    break;
  default:
    csr_product_kernel<32, MODE><<<blocks, 256, 0, s>>>(A.n, A.rp, A.ci, A.v, x, b, y);
    break;
  }
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

// y = Ainv b, dense n x n, one wavefront per row
__global__ void dense_apply_kernel(int n, const double* __restrict__ Ainv, const double* __restrict__ b,
                                   double* __restrict__ y)
{
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= n)
    return;
  double acc = 0.0;
  for (int k = lane; k < n; k += 64)
    acc += Ainv[(size_t)row * n + k] * b[k];
  for (int off = 32; off > 0; off >>= 1)
    acc += __shfl_down(acc, off, 64);
  if (lane == 0)
    y[row] = acc;
}

// ---- host threads for the set-up ------------------------------------------------------------
// The set-up runs on the host.  Its row-wise loops are cut into fixed blocks of rows that a few threads pull from a
// counter; every block's result depends on the block alone and partial sums are combined in block order, so the
// hierarchy is bit-identical whatever the number of threads -- which the replicated form relies on: every rank must
// build the same hierarchy.
int host_threads()
{
  static int cached = 0;
  if (cached == 0)
  {
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1)
      n = 1;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) // the container's CPU quota, not the host's core count
    {
      char q[32] = {0};
      long long per = 0;
      if (std::fscanf(f, "%31s %lld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0)
        n = std::min<long long>(n, std::max<long long>(1, (std::atoll(q) + per / 2) / per));
      std::fclose(f);
    }
    n = std::min(n, 16);
    if (const char* e = std::getenv("PMG_HOST_THREADS"))
      n = std::max(1, std::atoi(e));
    cached = n;
  }
  return cached;
}
constexpr int ROW_BLOCK = 4096;
// f(block, first row, last row (exclusive), thread) for every block of [0, n)
template <typename F>
void for_row_blocks(int n, F f)
{
  const int nb = (n + ROW_BLOCK - 1) / ROW_BLOCK;
  const int T = std::min(host_threads(), std::max(nb, 1));
  if (T <= 1)
  {
    for (int b = 0; b < nb; ++b)
      f(b, b * ROW_BLOCK, std::min(n, (b + 1) * ROW_BLOCK), 0);
    return;
  }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      for (int b = next.fetch_add(1); b < nb; b = next.fetch_add(1))
        f(b, b * ROW_BLOCK, std::min(n, (b + 1) * ROW_BLOCK), t);
    });
  for (auto& x : th)
    x.join();
}
// rows produced block by block -> one CSR matrix (row lengths in C.rp[i + 1] on entry)
void assemble_blocks(HostCsr& C, const std::vector<std::vector<int>>& bci, const std::vector<std::vector<double>>& bv)
{
  for (int i = 0; i < C.n; ++i)
    C.rp[i + 1] += C.rp[i];
  C.ci.resize(C.rp[C.n]);
  C.v.resize(C.rp[C.n]);
  for_row_blocks(C.n, [&](int b, int r0, int, int) {
    std::copy(bci[b].begin(), bci[b].end(), C.ci.begin() + C.rp[r0]);
    std::copy(bv[b].begin(), bv[b].end(), C.v.begin() + C.rp[r0]);
  });
}

// ---- host sparse algebra -------------------------------------------------------------------
HostCsr transpose(const HostCsr& A)
{
  HostCsr T;
  T.n = A.m;
  T.m = A.n;
  T.rp.assign(T.n + 1, 0);
  for (int c : A.ci)
    T.rp[c + 1]++;
  for (int i = 0; i < T.n; ++i)
    T.rp[i + 1] += T.rp[i];
  T.ci.resize(A.ci.size());
  T.v.resize(A.v.size());
  std::vector<int> pos(T.rp.begin(), T.rp.end() - 1);
  for (int i = 0; i < A.n; ++i)
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
    {
      const int p = pos[A.ci[k]]++;
      T.ci[p] = i;
      T.v[p] = A.v[k];
    }
  return T;
}

// C = A B (Gustavson), rows sorted by column
HostCsr spgemm(const HostCsr& A, const HostCsr& B)
{
  HostCsr C;
  C.n = A.n;
  C.m = B.m;
  C.rp.assign(C.n + 1, 0);
  const int nb = (A.n + ROW_BLOCK - 1) / ROW_BLOCK, T = host_threads();
  std::vector<std::vector<int>> bci(nb), marker(T);
  std::vector<std::vector<double>> bv(nb), acc(T);
  for_row_blocks(A.n, [&](int blk, int r0, int r1, int t) {
    if (marker[t].empty())
    {
      marker[t].assign(B.m, -1);
      acc[t].assign(B.m, 0.0);
    }
    std::vector<int>& mk = marker[t];
    std::vector<double>& ac = acc[t];
    std::vector<int> cols;
    for (int i = r0; i < r1; ++i)
    {
      cols.clear();
      for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
      {
        const int j = A.ci[k];
        const double a = A.v[k];
        for (int l = B.rp[j]; l < B.rp[j + 1]; ++l)
        {
          const int c = B.ci[l];
          if (mk[c] != i)
          {
            mk[c] = i;
            ac[c] = 0.0;
            cols.push_back(c);
          }
          ac[c] += a * B.v[l];
        }
      }
      std::sort(cols.begin(), cols.end());
      for (int c : cols)
      {
        bci[blk].push_back(c);
        bv[blk].push_back(ac[c]);
      }
      C.rp[i + 1] = (int)cols.size();
    }
  });
  assemble_blocks(C, bci, bv);
  return C;
}

std::vector<double> diagonal(const HostCsr& A)
{
  std::vector<double> d(A.n, 0.0);
  for (int i = 0; i < A.n; ++i)
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
      if (A.ci[k] == i)
        d[i] = A.v[k];
  return d;
}

// largest eigenvalue of D^-1 A by the power method (A SPD: D^-1 A has real positive eigenvalues);
// fixed seed and iteration count, so the hierarchy is reproducible
double lambda_max_jacobi(const HostCsr& A, const std::vector<double>& d, int its)
{
  const int n = A.n;
  if (n == 0)
    return 1.0;
  std::mt19937_64 gen(12345);
  std::uniform_real_distribution<double> U(0.5, 1.0);
  std::vector<double> x(n), y(n);
  for (double& v : x)
    v = U(gen);
  double lam = 1.0;
  const int nb = (n + ROW_BLOCK - 1) / ROW_BLOCK;
  std::vector<double> py(nb), px(nb);
  for (int it = 0; it < its; ++it)
  {
    for_row_blocks(n, [&](int blk, int r0, int r1, int) {
      double sy = 0, sx = 0;
      for (int i = r0; i < r1; ++i)
      {
        double s = 0;
        for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
          s += A.v[k] * x[A.ci[k]];
        y[i] = s / d[i];
        sy += y[i] * y[i];
        sx += x[i] * x[i];
      }
      py[blk] = sy;
      px[blk] = sx;
    });
    double nrm = 0, xn = 0;
    for (int blk = 0; blk < nb; ++blk) // block order: the same sums whatever the number of threads
    {
      nrm += py[blk];
      xn += px[blk];
    }
    nrm = std::sqrt(nrm);
    if (!(nrm > 0))
      return 1.0;
    lam = nrm / std::sqrt(xn);
    for_row_blocks(n, [&](int, int r0, int r1, int) {
      for (int i = r0; i < r1; ++i)
        x[i] = y[i] / nrm;
    });
  }
  return lam;
}

// Greedy aggregation on the strength graph.  agg[i] = aggregate of node i, -1 = the node takes no
// part in the coarse space (no strong connection: Dirichlet rows, isolated nodes).
int aggregate(const HostCsr& A, const std::vector<double>& d, double theta, std::vector<int>& agg)
{
  const int n = A.n;
  std::vector<int> srp(n + 1, 0), sci;
  std::vector<double> sw;
  for (int i = 0; i < n; ++i)
  {
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
    {
      const int j = A.ci[k];
      if (j != i && std::fabs(A.v[k]) >= theta * std::sqrt(std::fabs(d[i] * d[j])) && A.v[k] != 0.0)
      {
        sci.push_back(j);
        sw.push_back(std::fabs(A.v[k]));
      }
    }
    srp[i + 1] = (int)sci.size();
  }
  agg.assign(n, -2); // -2 = undecided
  int na = 0;
  for (int i = 0; i < n; ++i)
    if (srp[i] == srp[i + 1])
      agg[i] = -1;
  // pass 1: a node whose whole strong neighbourhood is free becomes a root
  for (int i = 0; i < n; ++i)
  {
    if (agg[i] != -2)
      continue;
    bool free_nb = true;
    for (int k = srp[i]; k < srp[i + 1] && free_nb; ++k)
      free_nb = agg[sci[k]] == -2;
    if (!free_nb)
      continue;
    agg[i] = na;
    for (int k = srp[i]; k < srp[i + 1]; ++k)
      agg[sci[k]] = na;
    ++na;
  }
  // pass 2: leftovers join the aggregate (of pass 1) they are most strongly tied to
  std::vector<int> agg1(agg);
  for (int i = 0; i < n; ++i)
  {
    if (agg[i] != -2)
      continue;
    double best = 0;
    int to = -2;
    for (int k = srp[i]; k < srp[i + 1]; ++k)
      if (agg1[sci[k]] >= 0 && sw[k] > best)
      {
        best = sw[k];
        to = agg1[sci[k]];
      }
    if (to >= 0)
      agg[i] = to;
  }
  // pass 3: what is still free forms aggregates of its own
  for (int i = 0; i < n; ++i)
  {
    if (agg[i] != -2)
      continue;
    agg[i] = na;
    for (int k = srp[i]; k < srp[i + 1]; ++k)
      if (agg[sci[k]] == -2)
        agg[sci[k]] = na;
    ++na;
  }
  return na;
}

// P = (I - omega D^-1 A) T, T the normalised piecewise-constant tentative prolongator
HostCsr smoothed_prolongator(const HostCsr& A, const std::vector<double>& d, const std::vector<int>& agg, int na,
                             double omega)
{
  const int n = A.n;
  std::vector<int> size(na, 0);
  for (int i = 0; i < n; ++i)
    if (agg[i] >= 0)
      size[agg[i]]++;
  std::vector<double> t(n, 0.0); // T[i, agg[i]]
  for (int i = 0; i < n; ++i)
    if (agg[i] >= 0)
      t[i] = 1.0 / std::sqrt((double)size[agg[i]]);
  HostCsr P;
  P.n = n;
  P.m = na;
  P.rp.assign(n + 1, 0);
  const int nb = (n + ROW_BLOCK - 1) / ROW_BLOCK, T = host_threads();
  std::vector<std::vector<int>> bci(nb), markers(T);
  std::vector<std::vector<double>> bv(nb), accs(T);
  for_row_blocks(n, [&](int blk, int r0, int r1, int th) {
    if (markers[th].empty())
    {
      markers[th].assign(std::max(na, 1), -1);
      accs[th].assign(std::max(na, 1), 0.0);
    }
    std::vector<int>& marker = markers[th];
    std::vector<double>& acc = accs[th];
    std::vector<int> cols;
    for (int i = r0; i < r1; ++i)
    {
      cols.clear();
      auto add = [&](int c, double val) {
        if (marker[c] != i)
        {
          marker[c] = i;
          acc[c] = 0.0;
          cols.push_back(c);
        }
        acc[c] += val;
      };
      if (agg[i] >= 0)
      {
        add(agg[i], t[i]);
        for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
        {
          const int j = A.ci[k];
          if (agg[j] >= 0)
            add(agg[j], -omega * A.v[k] / d[i] * t[j]);
        }
      }
      std::sort(cols.begin(), cols.end());
      int len = 0;
      for (int c : cols)
        if (acc[c] != 0.0)
        {
          bci[blk].push_back(c);
          bv[blk].push_back(acc[c]);
          ++len;
        }
      P.rp[i + 1] = len;
    }
  });
  assemble_blocks(P, bci, bv);
  return P;
}

// dense inverse of an SPD matrix by Cholesky (n <= a few thousand); returns false if not SPD
bool dense_spd_inverse(const HostCsr& A, std::vector<double>& inv)
{
  const int n = A.n;
  std::vector<double> L((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
      L[(size_t)i * n + A.ci[k]] = A.v[k];
  for (int j = 0; j < n; ++j) // in-place lower Cholesky
  {
    double s = L[(size_t)j * n + j];
    for (int k = 0; k < j; ++k)
      s -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(s > 0.0))
      return false;
    const double ljj = std::sqrt(s);
    L[(size_t)j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i)
    {
      double t = L[(size_t)i * n + j];
      for (int k = 0; k < j; ++k)
        t -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = t / ljj;
    }
  }
  inv.assign((size_t)n * n, 0.0);
  std::vector<double> col(n);
  for (int c = 0; c < n; ++c) // solve L L^T x = e_c
  {
    for (int i = 0; i < n; ++i)
    {
      double t = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k)
        t -= L[(size_t)i * n + k] * col[k];
      col[i] = t / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i)
    {
      double t = col[i];
      for (int k = i + 1; k < n; ++k)
        t -= L[(size_t)k * n + i] * col[k];
      col[i] = t / L[(size_t)i * n + i];
    }
    for (int i = 0; i < n; ++i)
      inv[(size_t)i * n + c] = col[i];
  }
  return true;
}
} // namespace

struct AmgLevel
{
  DevCsr A, P, R;
  double* dinv = nullptr;
  double lmax = 1.0;
  int n = 0;
  // work vectors of the level (x and b of level 0 are the caller's)
  double *x = nullptr, *b = nullptr, *r = nullptr, *z = nullptr, *q = nullptr;
  long long nnz = 0;
};

struct pmg_amg_s
{
  pmg_laplacian op = nullptr; // the degree-1 operator (Krylov mode applies it)
  pmg_layout layout = nullptr;
  std::vector<AmgLevel> levels;
  double* dense_inv = nullptr; // coarsest level, n x n
  int n_coarsest = 0;
  int smoother_its = 2;
  int cycles = 0;       // > 0: stationary mode, that many cycles per solve
  int max_iter = 60;    // Krylov mode (src/amg.hpp:39-40)
  double rtol = 1e-5;   // KSP's default relative tolerance
  pmg_cg cg = nullptr;  // work vectors of the Krylov mode
  double* xc = nullptr; // stationary mode with several cycles: correction
  int last_iterations = 0;
  // Replicated form (several ranks): the hierarchy is built on the GATHERED global matrix, every rank
  // solves the whole coarse problem (one all-reduce of the zero-padded right-hand side per solve, no
  // communication inside the solve); gid = global index of the owned dofs
  bool replicated = false;
  int32_t n_global = 0;
  int32_t* gid = nullptr;       // [size_local] device
  double* d0_dinv = nullptr;    // [size_local + ghosts] this rank's rows of the level-0 inverse diagonal (dist0)
  double *gb = nullptr, *gx = nullptr; // [n_global] device
  double* h_stage = nullptr;    // pinned, [n_global]: the callback route of the all-reduce
  pmg_layout glayout = nullptr; // the replicated problem seen as one rank's (Krylov work vectors)
  // Distributed fine level (replicated form, default): level 0 -- three quarters of a cycle's work -- stays on the
  // partitioned degree-1 operator (matrix-free application with its halo exchange, the layout's own smoother data);
  // only the levels below it are replicated.  A cycle then moves ONE all-reduce of a level-1 vector (1/9 of the
  // level-0 size) instead of a level-0 one, and every rank smooths its own share of level 0 instead of all of it.
  bool dist0 = false;
  bool distributed_setup = false; // level 0 was never gathered: the fully replicated solve is not available
  DevCsr P0l, R0l; // the level 0 -> 1 transfer restricted to the owned dofs: [size_local x n1], [n1 x size_local]
  double *d0_r = nullptr, *d0_z = nullptr, *d0_q = nullptr, *d0_b = nullptr, *d0_xc = nullptr; // on the layout
  pmg_cg cg0 = nullptr; // Krylov mode on the layout (distributed dot products)
  // host copy of the hierarchy for pmg_amg_export (tests)
  std::vector<HostCsr> hA, hP;
  std::vector<double> hlmax;
};

namespace
{
int alloc_d(double** p, size_t n)
{
  PMG_HIP(hipMalloc(p, sizeof(double) * std::max<size_t>(n, 1)));
  // hipMemset runs on the NULL stream and returns before it has run; the caller's stream is usually a non-blocking
  // one (torch's), which the null stream does not order: a kernel issued there right behind this call could be
  // overtaken by the fill (seen once as a smoother with a zero diagonal).  Set-up code: wait for it.
  PMG_HIP(hipMemset(*p, 0, sizeof(double) * std::max<size_t>(n, 1)));
  PMG_HIP(hipStreamSynchronize(nullptr));
  return PMG_OK;
}

// One V(k, k) cycle on level l: x = cycle(b), from a zero initial guess.
int amg_cycle(pmg_amg amg, int l, double* x, const double* b, hipStream_t s)
{
  const int L = (int)amg->levels.size();
  AmgLevel& lv = amg->levels[l];
  if (l == L - 1)
  {
    if (amg->dense_inv)
    {
      if (lv.n > 0)
        dense_apply_kernel<<<(lv.n * 64 + 255) / 256, 256, 0, s>>>(lv.n, amg->dense_inv, b, x);
      PMG_HIP(hipGetLastError());
      return PMG_OK;
    }
    // a coarsest level too large to invert: smooth it harder
    const ChebWork w{lv.r, lv.z, lv.q};
    return cheb_iterate(
        w, [&lv, s](double* in, double* out) { return csr_product<0>(lv.A, in, nullptr, out, s); }, lv.dinv, lv.n,
        lv.lmax, 4 * amg->smoother_its, x, b, ResidualNone, true, s);
  }
  const ChebWork w{lv.r, lv.z, lv.q};
  const ApplyFn A = [&lv, s](double* in, double* out) { return csr_product<0>(lv.A, in, nullptr, out, s); };
  AmgLevel& lc = amg->levels[l + 1];
  PMG_TRY(cheb_iterate(w, A, lv.dinv, lv.n, lv.lmax, amg->smoother_its, x, b, ResidualUpdated, true, s)); // leaves r = b - A x
  PMG_TRY(csr_product<0>(lv.R, lv.r, nullptr, lc.b, s));
  PMG_TRY(amg_cycle(amg, l + 1, lc.x, lc.b, s));
  PMG_TRY(csr_product<2>(lv.P, lc.x, nullptr, x, s)); // x += P x_c
  PMG_TRY(cheb_iterate(w, A, lv.dinv, lv.n, lv.lmax, amg->smoother_its, x, b, ResidualNone, false, s));
  return PMG_OK;
}
} // namespace

namespace
{
// scatter the owned entries into a zero global vector / read them back
__global__ void to_global_kernel(int n, const int32_t* __restrict__ gid, const double* __restrict__ v,
                                 double* __restrict__ g)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    g[gid[i]] = v[i];
}
__global__ void from_global_kernel(int n, const int32_t* __restrict__ gid, const double* __restrict__ g,
                                   double* __restrict__ v)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    v[i] = g[gid[i]];
}

// sum a host array over the ranks of a layout (set-up only; chunks that fit the transports)
int host_allreduce_sum(pmg_layout l, double* v, size_t n, hipStream_t s)
{
  if (!l->multi_rank() || n == 0)
    return PMG_OK;
  const size_t chunk = (size_t)1 << 24;
  if (l->comm)
  {
    double* d = nullptr;
    PMG_HIP(hipMalloc(&d, sizeof(double) * std::min(n, chunk)));
    int rc = PMG_OK;
    for (size_t o = 0; o < n && rc == PMG_OK; o += chunk)
    {
      const size_t m = std::min(chunk, n - o);
      if (hipMemcpyAsync(d, v + o, sizeof(double) * m, hipMemcpyHostToDevice, s) != hipSuccess)
        rc = fail(PMG_ERR_HIP, "host_allreduce_sum: copy failed");
      if (rc == PMG_OK)
        rc = comm_allreduce(l, d, (int)m, false, s);
      if (rc == PMG_OK && hipMemcpyAsync(v + o, d, sizeof(double) * m, hipMemcpyDeviceToHost, s) != hipSuccess)
        rc = fail(PMG_ERR_HIP, "host_allreduce_sum: copy failed");
      if (rc == PMG_OK && hipStreamSynchronize(s) != hipSuccess)
        rc = fail(PMG_ERR_HIP, "host_allreduce_sum: synchronise failed");
    }
    (void)hipFree(d);
    return rc;
  }
  for (size_t o = 0; o < n; o += chunk)
    if (l->allreduce(l->user, v + o, (int)std::min(chunk, n - o)) != 0)
      return fail(PMG_ERR_INVALID, "allreduce callback failed");
  return PMG_OK;
}

int build_hierarchy(pmg_amg amg, HostCsr&& A0, size_t level0_len, double theta0 = 0.08);
} // namespace

namespace pmg
{
pmg_layout amg_layout(pmg_amg amg) { return amg->layout; }

// stationary cycles are stream-ordered and capturable; the Krylov mode synchronises the host (and so
// does the callback route of the replicated form's all-reduce)
long long amg_capture_state(pmg_amg amg)
{
  if (amg->cycles <= 0 || (amg->replicated && !amg->layout->comm && amg->layout->allreduce))
    return -1;
  return ((long long)amg->cycles << 16) ^ amg->smoother_its ^ (amg->replicated ? 1 << 30 : 0) ^ (amg->dist0 ? 1 << 29 : 0);
}
} // namespace pmg

namespace
{
// the solve on one rank's vectors of n entries: `A` is the operator of the Krylov mode
int solve_on(pmg_amg amg, double* x, const double* b, int n, size_t total, const ApplyFn& A, hipStream_t s)
{
  if (amg->cycles > 0)
  {
    PMG_TRY(amg_cycle(amg, 0, x, b, s));
    AmgLevel& l0 = amg->levels[0];
    for (int c = 1; c < amg->cycles; ++c) // x += cycle(b - A x)
    {
      PMG_TRY(csr_product<1>(l0.A, x, b, l0.b, s));
      PMG_TRY(amg_cycle(amg, 0, amg->xc, l0.b, s));
      launch_add(n, x, amg->xc, s);
    }
    amg->last_iterations = amg->cycles;
    return PMG_OK;
  }
  // the reference's shape: CG (<= max_iter, rtol) on the operator, preconditioned by one cycle
  PMG_HIP(hipMemsetAsync(x, 0, sizeof(double) * total, s)); // KSP-style zero initial guess
  const PrecondFn M = [amg, s](double* z, const double* r) { return amg_cycle(amg, 0, z, r, s); };
  PMG_TRY(pmg_cg_set_max_iterations(amg->cg, amg->max_iter));
  PMG_TRY(pmg_cg_set_tolerance(amg->cg, amg->rtol));
  return cg_iterate(amg->cg, A, nullptr, &M, false, x, b, &amg->last_iterations, s);
}
} // namespace

namespace
{
// sum a device vector over the ranks of the layout (communicator, or the callbacks through pinned memory)
int device_allreduce_sum(pmg_amg amg, double* v, int n, hipStream_t s)
{
  pmg_layout l = amg->layout;
  if (l->comm)
    return comm_allreduce(l, v, n, false, s);
  if (l->allreduce)
  {
    PMG_HIP(hipMemcpyAsync(amg->h_stage, v, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    PMG_HIP(hipStreamSynchronize(s));
    for (size_t o = 0; o < (size_t)n; o += (size_t)1 << 24)
      if (l->allreduce(l->user, amg->h_stage + o, (int)std::min<size_t>((size_t)1 << 24, n - o)) != 0)
        return fail(PMG_ERR_INVALID, "allreduce callback failed");
    PMG_HIP(hipMemcpyAsync(v, amg->h_stage, sizeof(double) * n, hipMemcpyHostToDevice, s));
  }
  return PMG_OK;
}

// One V(k, k) cycle with the fine level distributed: x = cycle(b) on the layout's vectors, zero initial guess.
int dist_cycle(pmg_amg amg, double* x, const double* b, hipStream_t s)
{
  pmg_layout l = amg->layout;
  pmg_laplacian op = amg->op;
  const int n = l->size_local;
  AmgLevel& l0 = amg->levels[0];
  AmgLevel& l1 = amg->levels[1];
  const ChebWork w{amg->d0_r, amg->d0_z, amg->d0_q};
  const ApplyFn A = [op, s](double* in, double* out) { return laplacian_apply(op, in, out, s); };
  // The inverse diagonal of the hierarchy's OWN level-0 matrix, restricted to this rank (d0_dinv): the bound l0.lmax
  // was computed for D^-1 A with that diagonal, and the operator's diag_inv may never have been computed or may have
  // been replaced by the caller (pmg_laplacian_set_diag_inverse) -- ADVICE r03.
  const double* dinv = amg->d0_dinv;
  PMG_TRY(cheb_iterate(w, A, dinv, n, l0.lmax, amg->smoother_its, x, b, ResidualUpdated, true, s)); // r = b - A x
  PMG_TRY(csr_product<0>(amg->R0l, amg->d0_r, nullptr, l1.b, s)); // this rank's share of R r
  PMG_TRY(device_allreduce_sum(amg, l1.b, l1.n, s));
  PMG_TRY(amg_cycle(amg, 1, l1.x, l1.b, s));                      // replicated from here down
  PMG_TRY(csr_product<2>(amg->P0l, l1.x, nullptr, x, s));         // x += P x_c, owned rows
  PMG_TRY(cheb_iterate(w, A, dinv, n, l0.lmax, amg->smoother_its, x, b, ResidualNone, false, s));
  return PMG_OK;
}

int dist_solve(pmg_amg amg, double* x, const double* b, hipStream_t s)
{
  pmg_layout l = amg->layout;
  pmg_laplacian op = amg->op;
  const int n = l->size_local;
  if (amg->cycles > 0)
  {
    PMG_TRY(dist_cycle(amg, x, b, s));
    for (int c = 1; c < amg->cycles; ++c) // x += cycle(b - A x)
    {
      PMG_TRY(laplacian_apply(op, x, amg->d0_q, s));
      launch_axpy(n, amg->d0_b, -1.0, amg->d0_q, b, s);
      PMG_TRY(dist_cycle(amg, amg->d0_xc, amg->d0_b, s));
      launch_add(n, x, amg->d0_xc, s);
    }
    amg->last_iterations = amg->cycles;
    return PMG_OK;
  }
  PMG_HIP(hipMemsetAsync(x, 0, sizeof(double) * l->total(), s)); // KSP-style zero initial guess
  const ApplyFn A = [op, s](double* in, double* out) { return laplacian_apply(op, in, out, s); };
  const PrecondFn M = [amg, s](double* z, const double* r) { return dist_cycle(amg, z, r, s); };
  PMG_TRY(pmg_cg_set_max_iterations(amg->cg0, amg->max_iter));
  PMG_TRY(pmg_cg_set_tolerance(amg->cg0, amg->rtol));
  return cg_iterate(amg->cg0, A, nullptr, &M, false, x, b, &amg->last_iterations, s);
}
} // namespace

namespace pmg
{
// x = (approximately) A^-1 b on the coarsest p-level, x zero on entry is not assumed
int amg_solve(pmg_amg amg, double* x, const double* b, hipStream_t s)
{
  Range range("pmg:amg_solve");
  pmg_layout l = amg->layout;
  const int n = l->size_local;
  if (amg->replicated && amg->dist0)
    return dist_solve(amg, x, b, s);
  if (amg->replicated)
  {
    // gather the right-hand side (zero-padded global vector, summed over the ranks), solve the WHOLE
    // coarse problem on every rank -- identical arithmetic everywhere, no communication inside -- and keep
    // the owned entries
    const int ng = amg->n_global;
    PMG_HIP(hipMemsetAsync(amg->gb, 0, sizeof(double) * ng, s));
    if (n > 0)
      to_global_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, amg->gid, b, amg->gb);
    PMG_HIP(hipGetLastError());
    if (l->comm)
      PMG_TRY(comm_allreduce(l, amg->gb, ng, false, s));
    else if (l->allreduce)
    {
      PMG_HIP(hipMemcpyAsync(amg->h_stage, amg->gb, sizeof(double) * ng, hipMemcpyDeviceToHost, s));
      PMG_HIP(hipStreamSynchronize(s));
      for (size_t o = 0; o < (size_t)ng; o += (size_t)1 << 24)
        if (l->allreduce(l->user, amg->h_stage + o, (int)std::min<size_t>((size_t)1 << 24, ng - o)) != 0)
          return fail(PMG_ERR_INVALID, "allreduce callback failed");
      PMG_HIP(hipMemcpyAsync(amg->gb, amg->h_stage, sizeof(double) * ng, hipMemcpyHostToDevice, s));
    }
    AmgLevel& l0 = amg->levels[0];
    const ApplyFn A = [&l0, s](double* in, double* out) { return csr_product<0>(l0.A, in, nullptr, out, s); };
    PMG_TRY(solve_on(amg, amg->gx, amg->gb, ng, (size_t)ng, A, s));
    if (n > 0)
      from_global_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, amg->gid, amg->gx, x);
    PMG_HIP(hipGetLastError());
    return PMG_OK;
  }
  PMG_REQUIRE(amg->cycles == 0 || !l->multi_rank(),
              "pmg_amg: stationary cycles of a rank-local hierarchy are a single-rank solver; on several ranks "
              "use the Krylov mode or the replicated hierarchy (pmg_amg_create_replicated)");
  pmg_laplacian op = amg->op;
  const ApplyFn A = [op, s](double* in, double* out) { return laplacian_apply(op, in, out, s); };
  return solve_on(amg, x, b, n, (size_t)l->total(), A, s);
}
} // namespace pmg


// `global_index` == nullptr: the hierarchy of this rank's own block.  Otherwise the replicated form.
static int amg_create_distributed_tail(pmg_amg amg, HostCsr&& A0, const int64_t* global_index, int64_t n_global,
                                       const std::vector<int8_t>& bc, hipStream_t s);

// distributed_setup: the first coarsening per rank, only the levels below it gathered (pmg_amg_create_distributed)
static int amg_create(pmg_amg* out, pmg_laplacian op, const int64_t* global_index, int64_t n_global,
                      pmg_stream stream, bool distributed_setup = false)
{
  PMG_REQUIRE(out && op, "pmg_amg_create: NULL argument");
  const LaplacianInputs in = laplacian_inputs(op);
  PMG_REQUIRE(in.degree == 1, "pmg_amg_create: the operator must have degree 1 (it has degree %d)", in.degree);
  hipStream_t s = S(stream);
  pmg_layout layout = laplacian_layout(op);
  const int n = layout->size_local, total = layout->total();
  const bool replicated = global_index != nullptr;
  if (replicated)
  {
    PMG_REQUIRE(n_global >= n && n_global < ((int64_t)1 << 31), "pmg_amg_create_replicated: bad global size");
    for (int i = 0; i < total; ++i)
      PMG_REQUIRE(global_index[i] >= 0 && global_index[i] < n_global,
                  "pmg_amg_create_replicated: global index %lld out of range", (long long)global_index[i]);
  }
  auto* amg = new pmg_amg_s;
  HandleGuard<pmg_amg> guard(amg, pmg_amg_destroy);
  amg->op = op;
  amg->layout = layout;
  amg->replicated = replicated;
  amg->n_global = replicated ? (int32_t)n_global : 0;

  StageTimer tmc;
  // ---- level 0: assemble the owned block of the degree-1 stiffness matrix ----
  std::vector<int32_t> dofmap((size_t)in.ncells * 8);
  std::vector<int8_t> bc(total);
  std::vector<double> kappa(in.ncells), G((size_t)in.ncells * 8 * 6);
  {
    double* dG = nullptr;
    PMG_HIP(hipMalloc(&dG, sizeof(double) * std::max<size_t>(G.size(), 1)));
    int rc = pmg_laplacian_get_geometry(op, dG, stream);
    if (rc == PMG_OK && !G.empty())
      rc = hipMemcpyAsync(G.data(), dG, sizeof(double) * G.size(), hipMemcpyDeviceToHost, s) == hipSuccess
               ? PMG_OK
               : fail(PMG_ERR_HIP, "pmg_amg_create: copy of the geometry tensor failed");
    if (rc == PMG_OK)
      rc = hipStreamSynchronize(s) == hipSuccess ? PMG_OK : fail(PMG_ERR_HIP, "pmg_amg_create: synchronise failed");
    (void)hipFree(dG);
    PMG_TRY(rc);
  }
  if (in.ncells > 0)
  {
    PMG_HIP(hipMemcpy(dofmap.data(), in.dofmap, sizeof(int32_t) * dofmap.size(), hipMemcpyDeviceToHost));
    PMG_HIP(hipMemcpy(kappa.data(), in.kappa, sizeof(double) * kappa.size(), hipMemcpyDeviceToHost));
  }
  if (total > 0)
    PMG_HIP(hipMemcpy(bc.data(), in.bc, sizeof(int8_t) * bc.size(), hipMemcpyDeviceToHost));

  // the cells the operator lists (both cell lists), from its patches
  const PatchView pv = laplacian_patches(op);
  std::vector<int32_t> cells;
  for (int32_t c : *pv.pcell_h)
    if (c >= 0)
      cells.push_back(c);

  HostCsr A0;
  {
    constexpr int W = 32; // row capacity of the fast path (a trilinear hex mesh has <= 27)
    std::vector<int> cnt(n, 0), cols((size_t)n * W, -1);
    std::vector<double> vals((size_t)n * W, 0.0);
    std::vector<std::vector<std::pair<int, double>>> spill(0);
    std::vector<int> spill_of(n, -1);
    auto add = [&](int row, int col, double v) {
      int* rc = cols.data() + (size_t)row * W;
      double* rv = vals.data() + (size_t)row * W;
      const int c = cnt[row];
      for (int k = 0; k < c && k < W; ++k)
        if (rc[k] == col)
        {
          rv[k] += v;
          return;
        }
      if (c < W)
      {
        rc[c] = col;
        rv[c] = v;
        cnt[row] = c + 1;
        return;
      }
      if (spill_of[row] < 0)
      {
        spill_of[row] = (int)spill.size();
        spill.emplace_back();
      }
      auto& sp = spill[spill_of[row]];
      for (auto& e : sp)
        if (e.first == col)
        {
          e.second += v;
          return;
        }
      sp.emplace_back(col, v);
    };
    for (int32_t c : cells)
    {
      const int32_t* dm = dofmap.data() + (size_t)c * 8;
      const double* Gc = G.data() + (size_t)c * 48;
      double Ke[8][8] = {};
      // collocated basis on the 2-point GLL rule: at quadrature point q only the basis functions of
      // q itself and of its three axis neighbours have a gradient; l_0' = -1, l_1' = +1
      for (int q = 0; q < 8; ++q)
      {
        const double* g = Gc + q * 6; // (G00, G01, G02, G11, G12, G22), src/laplacian.hpp:99-111
        const double Gm[3][3] = {{g[0], g[1], g[2]}, {g[1], g[3], g[4]}, {g[2], g[4], g[5]}};
        const int qa = (q >> 2) & 1, qb = (q >> 1) & 1, qc = q & 1;
        const int node[4] = {q, q ^ 4, q ^ 2, q ^ 1};
        const double sa = qa ? 1.0 : -1.0, sb = qb ? 1.0 : -1.0, sc = qc ? 1.0 : -1.0;
        const double grad[4][3] = {{sa, sb, sc}, {-sa, 0, 0}, {0, -sb, 0}, {0, 0, -sc}};
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j)
          {
            double v = 0;
            for (int d = 0; d < 3; ++d)
              for (int e = 0; e < 3; ++e)
                v += grad[i][d] * Gm[d][e] * grad[j][e];
            Ke[node[i]][node[j]] += kappa[c] * v;
          }
      }
      for (int i = 0; i < 8; ++i)
      {
        const int row = dm[i];
        if (row >= n || bc[row])
          continue; // ghost rows are another rank's; Dirichlet rows are identity rows
        for (int j = 0; j < 8; ++j)
        {
          const int col = dm[j];
          if (bc[col] || (!replicated && col >= n))
            continue; // Dirichlet columns: masked; ghost columns: dropped from a rank-local hierarchy
          add(row, col, Ke[i][j]);
        }
      }
    }
    A0.n = n;
    A0.m = replicated ? total : n;
    A0.rp.assign(n + 1, 0);
    std::vector<std::pair<int, double>> row;
    for (int i = 0; i < n; ++i)
    {
      row.clear();
      for (int k = 0; k < std::min(cnt[i], W); ++k)
        row.emplace_back(cols[(size_t)i * W + k], vals[(size_t)i * W + k]);
      if (spill_of[i] >= 0)
        row.insert(row.end(), spill[spill_of[i]].begin(), spill[spill_of[i]].end());
      if (row.empty())
        row.emplace_back(i, 1.0); // Dirichlet dof, or a dof of no listed cell: identity row
      std::sort(row.begin(), row.end());
      for (auto& e : row)
      {
        A0.ci.push_back(e.first);
        A0.v.push_back(e.second);
      }
      A0.rp[i + 1] = (int)A0.ci.size();
    }
  }

  tmc.lap("level-0 assembly");
  if (replicated && distributed_setup)
  {
    PMG_TRY(amg_create_distributed_tail(amg, std::move(A0), global_index, n_global, bc, s));
    tmc.lap("distributed first coarsening + hierarchy below it");
    PMG_HIP(hipStreamSynchronize(s));
    *out = guard.release();
    return PMG_OK;
  }
  size_t level0_len = (size_t)total;
  if (replicated)
  {
    // ---- gather: every rank contributes its owned rows with global column ids in fixed-width slots of
    // a zero global table; the sum over the ranks is the global matrix (column + 1, so that 0 = empty)
    int wloc = 0;
    for (int i = 0; i < n; ++i)
      wloc = std::max(wloc, A0.rp[i + 1] - A0.rp[i]);
    // the maximum over the ranks, from sums: one slot per width would be wasteful, use a unary code
    std::vector<double> wcode(256, 0.0);
    PMG_REQUIRE(wloc < 256, "pmg_amg_create_replicated: a row with %d entries", wloc);
    wcode[wloc] = 1.0;
    PMG_TRY(host_allreduce_sum(layout, wcode.data(), wcode.size(), s));
    int W = 1;
    for (int w = 0; w < 256; ++w)
      if (wcode[w] > 0.0)
        W = std::max(W, w);
    const size_t ng = (size_t)n_global;
    std::vector<double> gc(ng * W, 0.0), gv(ng * W, 0.0);
    for (int i = 0; i < n; ++i)
    {
      const size_t g = (size_t)global_index[i];
      int k = 0;
      for (int e = A0.rp[i]; e < A0.rp[i + 1]; ++e, ++k)
      {
        gc[g * W + k] = (double)(global_index[A0.ci[e]] + 1);
        gv[g * W + k] = A0.v[e];
      }
    }
    PMG_TRY(host_allreduce_sum(layout, gc.data(), gc.size(), s));
    PMG_TRY(host_allreduce_sum(layout, gv.data(), gv.size(), s));
    HostCsr Ag;
    Ag.n = Ag.m = (int)ng;
    Ag.rp.assign(ng + 1, 0);
    {
      const int nbk = ((int)ng + ROW_BLOCK - 1) / ROW_BLOCK;
      std::vector<std::vector<int>> bci(nbk);
      std::vector<std::vector<double>> bv(nbk);
      std::atomic<long long> unowned(-1);
      for_row_blocks((int)ng, [&](int blk, int r0, int r1, int) {
        std::vector<std::pair<int, double>> row;
        for (int g = r0; g < r1; ++g)
        {
          row.clear();
          for (int k = 0; k < W; ++k)
            if (gc[(size_t)g * W + k] > 0.5)
              row.emplace_back((int)(gc[(size_t)g * W + k] - 0.5), gv[(size_t)g * W + k]);
          if (row.empty())
            unowned.store(g);
          std::sort(row.begin(), row.end());
          for (auto& e : row)
          {
            bci[blk].push_back(e.first);
            bv[blk].push_back(e.second);
          }
          Ag.rp[g + 1] = (int)row.size();
        }
      });
      PMG_REQUIRE(unowned.load() < 0, "pmg_amg_create_replicated: global dof %lld is owned by no rank", unowned.load());
      assemble_blocks(Ag, bci, bv);
    }
    A0 = std::move(Ag);
    level0_len = ng;
    std::vector<int32_t> gid(n);
    for (int i = 0; i < n; ++i)
      gid[i] = (int32_t)global_index[i];
    PMG_TRY(to_device(&amg->gid, gid));
    PMG_TRY(alloc_d(&amg->gb, ng));
    PMG_TRY(alloc_d(&amg->gx, ng));
    PMG_HIP(hipHostMalloc(&amg->h_stage, sizeof(double) * std::max<size_t>(ng, 1), hipHostMallocDefault));
    PMG_TRY(pmg_layout_create(&amg->glayout, (int32_t)ng, 0, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                              nullptr, nullptr));
  }
  tmc.lap("gather (replicated form)");
  PMG_TRY(build_hierarchy(amg, std::move(A0), level0_len));
  PMG_TRY(pmg_cg_create(&amg->cg, replicated ? amg->glayout : layout));
  if (replicated && amg->levels.size() >= 2)
  {
    // the rows of P_0 of this rank's owned dofs, in the layout's numbering; R_0 restricted likewise
    const HostCsr& P0 = amg->hP[0];
    HostCsr Pl;
    Pl.n = n;
    Pl.m = P0.m;
    Pl.rp.assign(n + 1, 0);
    for (int i = 0; i < n; ++i)
    {
      const int g = (int)global_index[i];
      for (int e = P0.rp[g]; e < P0.rp[g + 1]; ++e)
      {
        Pl.ci.push_back(P0.ci[e]);
        Pl.v.push_back(P0.v[e]);
      }
      Pl.rp[i + 1] = (int)Pl.ci.size();
    }
    PMG_TRY(upload_csr(amg->P0l, Pl));
    PMG_TRY(upload_csr(amg->R0l, transpose(Pl)));
    PMG_TRY(alloc_d(&amg->d0_r, total));
    PMG_TRY(alloc_d(&amg->d0_z, total));
    PMG_TRY(alloc_d(&amg->d0_q, total));
    PMG_TRY(alloc_d(&amg->d0_b, total));
    PMG_TRY(alloc_d(&amg->d0_xc, total));
    PMG_TRY(alloc_d(&amg->d0_dinv, total));
    if (n > 0) // owned rows of the assembled level-0 diagonal (levels[0].dinv is indexed by global row)
      from_global_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, amg->gid, amg->levels[0].dinv, amg->d0_dinv);
    PMG_HIP(hipGetLastError());
    PMG_TRY(pmg_cg_create(&amg->cg0, layout));
    amg->dist0 = true;
  }
  PMG_HIP(hipStreamSynchronize(s));
  *out = guard.release();
  return PMG_OK;
}

extern "C" int pmg_amg_create(pmg_amg* out, pmg_laplacian op, pmg_stream stream)
{
  return amg_create(out, op, nullptr, 0, stream);
}

extern "C" int pmg_amg_create_replicated(pmg_amg* out, pmg_laplacian op, const int64_t* global_index,
                                         int64_t n_global, pmg_stream stream)
{
  PMG_REQUIRE(global_index, "pmg_amg_create_replicated: NULL global index");
  return amg_create(out, op, global_index, n_global, stream);
}

namespace
{
// theta0: strength threshold of the first coarsening done here (halved from level to level)
int build_hierarchy(pmg_amg amg, HostCsr&& A0, size_t level0_len, double theta0)
{
  StageTimer tm;
  // ---- the hierarchy ----
  const int max_levels = 12, coarsest_max = 800;
  std::vector<HostCsr> As, Ps;
  As.push_back(std::move(A0));
  std::vector<double> lmaxs;
  double theta = theta0;
  while (true)
  {
    const HostCsr& A = As.back();
    const std::vector<double> d = diagonal(A);
    for (int i = 0; i < A.n; ++i)
      PMG_REQUIRE(d[i] > 0.0, "pmg_amg_create: non-positive diagonal entry on level %d", (int)As.size() - 1);
    const double rho = 1.05 * lambda_max_jacobi(A, d, 20);
    tm.lap("power method");
    lmaxs.push_back(rho);
    if (A.n <= coarsest_max || (int)As.size() >= max_levels)
      break;
    std::vector<int> agg;
    const int na = aggregate(A, d, theta, agg);
    tm.lap("aggregation");
    if (na == 0 || na >= A.n)
      break; // nothing to coarsen
    HostCsr P = smoothed_prolongator(A, d, agg, na, 4.0 / (3.0 * rho));
    tm.lap("smoothed prolongator");
    HostCsr R = transpose(P);
    tm.lap("transpose");
    HostCsr AP = spgemm(A, P);
    tm.lap("A P");
    HostCsr Ac = spgemm(R, AP);
    tm.lap("R (A P)");
    Ps.push_back(std::move(P));
    As.push_back(std::move(Ac));
    theta *= 0.5;
  }

  const int L = (int)As.size();
  tm.lap("(levels done)");
  amg->levels.resize(L);
  for (int l = 0; l < L; ++l)
  {
    AmgLevel& lv = amg->levels[l];
    lv.n = As[l].n;
    lv.nnz = As[l].nnz();
    lv.lmax = lmaxs[l];
    PMG_TRY(upload_csr(lv.A, As[l]));
    std::vector<double> dinv = diagonal(As[l]);
    for (double& v : dinv)
      v = 1.0 / v;
    PMG_TRY(to_device(&lv.dinv, dinv));
    if (l + 1 < L)
    {
      PMG_TRY(upload_csr(lv.P, Ps[l]));
      PMG_TRY(upload_csr(lv.R, transpose(Ps[l])));
    }
    const size_t len = (l == 0) ? level0_len : (size_t)lv.n;
    PMG_TRY(alloc_d(&lv.r, len));
    PMG_TRY(alloc_d(&lv.z, len));
    PMG_TRY(alloc_d(&lv.q, len));
    PMG_TRY(alloc_d(&lv.b, len)); // level 0: the residual of the second and later stationary cycles
    if (l > 0)
      PMG_TRY(alloc_d(&lv.x, len));
  }
  PMG_TRY(alloc_d(&amg->xc, level0_len));
  // coarsest level: dense inverse when small enough
  {
    const HostCsr& Ac = As.back();
    std::vector<double> inv;
    if (Ac.n <= 4096 && dense_spd_inverse(Ac, inv))
    {
      amg->n_coarsest = Ac.n;
      PMG_TRY(to_device(&amg->dense_inv, inv));
    }
  }
  tm.lap("upload + dense inverse");
  amg->hA = std::move(As);
  amg->hP = std::move(Ps);
  amg->hlmax = lmaxs;
  return PMG_OK;
}
} // namespace

// ---- distributed set-up (round 4): the first coarsening per rank ------------------------------------------------
// pmg_amg_create_replicated gathers the GLOBAL degree-1 matrix on every rank and coarsens it there (2.1 M rows and
// ~2.5 s per rank at 8 x 64^3).  Here no rank ever holds the global level-0 matrix:
//   * every rank aggregates its OWNED dofs (aggregates do not cross rank boundaries) and numbers its aggregates
//     globally by the smallest global dof number among their members (one all-reduce of a marker vector; no rank
//     numbers needed);
//   * the tentative prolongator of the ghost dofs (aggregate number, weight) arrives through the layout's own forward
//     scatter, so the owned rows of P_0 = (I - omega D^-1 A_0) T are smoothed with the FULL rows of A_0 (a first
//     version smoothed with the rank's own block only: 13 instead of 9 iterations on 8 ranks);
//   * A_0 P_0 on the owned rows needs the rows of P_0 of the ghost dofs: ONE layer of overlap, moved as padded
//     (column, value) tables through the same forward scatter -- any exchange mechanism, callbacks included;
//   * every rank forms its SHARE of A_1 = P_0^T A_0 P_0 (the sum over its owned fine rows) and level 1 -- about 1/9 of
//     level 0 -- is the sum of the shares, gathered on every rank as (position, value) pairs in per-rank segments of
//     one all-reduced array; the hierarchy below it is built and solved replicated, exactly as the solve phase has done
//     since round 3 (level 0 smoothed on the partitioned operator, one all-reduce of a level-1 vector per cycle);
//   * the smoothing bound of level 0 comes from a power method on the partitioned matrix-free operator.
namespace
{
// max over the ranks of a small non-negative integer (< 4096), from sums: a unary code
int allreduce_max_small(pmg_layout l, int v, int* out, hipStream_t s)
{
  std::vector<double> code(4096, 0.0);
  PMG_REQUIRE(v >= 0 && v < 4096, "pmg_amg: a row with %d entries", v);
  code[v] = 1.0;
  PMG_TRY(host_allreduce_sum(l, code.data(), code.size(), s));
  int m = 0;
  for (int w = 0; w < 4096; ++w)
    if (code[w] > 0.0)
      m = w;
  *out = m;
  return PMG_OK;
}

// largest eigenvalue of D^-1 A of the PARTITIONED operator by the power method on the device (the operator's own halo
// exchange, the layout's reductions); the start vector depends on the global dof number only, so every partition of the
// same problem iterates on the same vector
int lambda_max_distributed(pmg_amg amg, const int64_t* global_index, const double* dinv_d, int its, double* lam,
                           hipStream_t s)
{
  pmg_layout l = amg->layout;
  const int n = l->size_local, total = l->total();
  std::vector<double> x0(total, 0.0);
  for (int i = 0; i < n; ++i)
  {
    uint64_t h = (uint64_t)global_index[i] * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    x0[i] = 0.5 + 0.5 * (double)(h >> 11) / 9007199254740992.0;
  }
  double *x = amg->d0_r, *q = amg->d0_q;
  PMG_HIP(hipMemcpyAsync(x, x0.data(), sizeof(double) * total, hipMemcpyHostToDevice, s));
  PMG_HIP(hipStreamSynchronize(s));
  double value = 1.0;
  for (int it = 0; it < its; ++it)
  {
    PMG_TRY(laplacian_apply(amg->op, x, q, s));
    launch_pointwise(n, q, q, dinv_d, s);
    double qq = 0.0, xx = 0.0;
    PMG_TRY(dot_host(l, q, q, &qq, s));
    PMG_TRY(dot_host(l, x, x, &xx, s));
    if (!(qq > 0.0) || !(xx > 0.0))
      break;
    value = std::sqrt(qq / xx);
    PMG_HIP(hipMemcpyAsync(x, q, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    PMG_TRY(pmg_vec_scale(l, x, 1.0 / std::sqrt(qq), (pmg_stream)s));
  }
  *lam = value;
  return PMG_OK;
}
} // namespace

// one scalar per local dof moved owner -> ghost through the layout's forward scatter (host arrays of `total` entries)
static int host_scatter_fwd(pmg_layout layout, std::vector<double>& h, double* dv, hipStream_t s)
{
  const int n = layout->size_local, total = layout->total(), ng = total - n;
  PMG_HIP(hipMemcpyAsync(dv, h.data(), sizeof(double) * total, hipMemcpyHostToDevice, s));
  PMG_TRY(pmg_scatter_fwd_begin(layout, dv, (pmg_stream)s));
  PMG_TRY(pmg_scatter_fwd_end(layout, dv, (pmg_stream)s));
  if (ng > 0)
    PMG_HIP(hipMemcpyAsync(h.data() + n, dv + n, sizeof(double) * ng, hipMemcpyDeviceToHost, s));
  PMG_HIP(hipStreamSynchronize(s));
  return PMG_OK;
}

static int amg_create_distributed_tail(pmg_amg amg, HostCsr&& A0, const int64_t* global_index, int64_t n_global,
                                       const std::vector<int8_t>& bc, hipStream_t s)
{
  pmg_layout layout = amg->layout;
  const int n = layout->size_local, total = layout->total(), ng = total - n;
  amg->distributed_setup = true;
  (void)bc;
  StageTimer tm;
  const std::vector<double> d = diagonal(A0); // the diagonal of the owned rows: the smoother's D
  for (int i = 0; i < n; ++i)
    PMG_REQUIRE(d[i] > 0.0, "pmg_amg_create_distributed: row %d has no positive diagonal", i);
  // what the solve needs on the layout, and the smoothing bound of level 0 first (the prolongator uses it)
  PMG_TRY(alloc_d(&amg->d0_r, total));
  PMG_TRY(alloc_d(&amg->d0_z, total));
  PMG_TRY(alloc_d(&amg->d0_q, total));
  PMG_TRY(alloc_d(&amg->d0_b, total));
  PMG_TRY(alloc_d(&amg->d0_xc, total));
  {
    std::vector<double> dinv(total, 1.0);
    for (int i = 0; i < n; ++i)
      dinv[i] = 1.0 / d[i];
    PMG_TRY(to_device(&amg->d0_dinv, dinv));
  }
  double lam0 = 1.0;
  PMG_TRY(lambda_max_distributed(amg, global_index, amg->d0_dinv, 20, &lam0, s));
  const double rho0 = 1.05 * lam0;
  tm.lap("level-0 bound (power method on the partitioned operator)");

  // (A) aggregation of the OWNED dofs on the rank's own block (aggregates do not cross rank boundaries)
  HostCsr Aown;
  Aown.n = Aown.m = n;
  Aown.rp.assign(n + 1, 0);
  for (int i = 0; i < n; ++i)
  {
    for (int e = A0.rp[i]; e < A0.rp[i + 1]; ++e)
      if (A0.ci[e] < n)
      {
        Aown.ci.push_back(A0.ci[e]);
        Aown.v.push_back(A0.v[e]);
      }
    Aown.rp[i + 1] = (int)Aown.ci.size();
  }
  std::vector<int> agg;
  const int na = aggregate(Aown, d, 0.08, agg);
  Aown = HostCsr();
  std::vector<int> size(std::max(na, 1), 0);
  for (int i = 0; i < n; ++i)
    if (agg[i] >= 0)
      size[agg[i]]++;
  tm.lap("aggregation of the owned dofs");

  // (B) global numbers of the aggregates: rank among all aggregates' smallest global member
  std::vector<int64_t> root(na, INT64_MAX);
  for (int i = 0; i < n; ++i)
    if (agg[i] >= 0)
      root[agg[i]] = std::min(root[agg[i]], global_index[i]);
  std::vector<double> marker((size_t)n_global, 0.0);
  for (int a = 0; a < na; ++a)
  {
    PMG_REQUIRE(root[a] != INT64_MAX, "pmg_amg_create_distributed: empty aggregate");
    marker[(size_t)root[a]] = 1.0;
  }
  PMG_TRY(host_allreduce_sum(layout, marker.data(), marker.size(), s));
  int n1 = 0;
  {
    std::vector<int32_t> cid((size_t)n_global, -1);
    for (int64_t g = 0; g < n_global; ++g)
      if (marker[(size_t)g] > 0.5)
      {
        PMG_REQUIRE(marker[(size_t)g] < 1.5, "pmg_amg_create_distributed: dof %lld is owned by two ranks", (long long)g);
        cid[(size_t)g] = n1++;
      }
    for (int a = 0; a < na; ++a)
      root[a] = cid[(size_t)root[a]]; // from here on: the aggregate's global number
  }
  PMG_REQUIRE(n1 > 0, "pmg_amg_create_distributed: nothing to coarsen");
  tm.lap("global aggregate numbers");

  // (C) the tentative prolongator on owned AND ghost dofs (aggregate number + 1 and weight 1 / sqrt(size): one scalar
  // exchange each), then the owned rows of P_0 = (I - 4 / (3 rho) D^-1 A_0) T with the FULL rows of A_0 -- ghost
  // couplings included, so the columns of a rank's rows may be neighbours' aggregates
  double* dv = nullptr;
  PMG_HIP(hipMalloc(&dv, sizeof(double) * std::max(total, 1)));
  std::vector<double> tcol(total, 0.0), tval(total, 0.0);
  for (int i = 0; i < n; ++i)
    if (agg[i] >= 0)
    {
      tcol[i] = (double)(root[agg[i]] + 1);
      tval[i] = 1.0 / std::sqrt((double)size[agg[i]]);
    }
  int rc = host_scatter_fwd(layout, tcol, dv, s);
  if (rc == PMG_OK)
    rc = host_scatter_fwd(layout, tval, dv, s);
  if (rc != PMG_OK)
  {
    (void)hipFree(dv);
    return rc;
  }
  const double omega = 4.0 / (3.0 * rho0);
  HostCsr Pl; // [n x n1], global coarse columns
  Pl.n = n;
  Pl.m = n1;
  Pl.rp.assign(n + 1, 0);
  {
    const int nb = (n + ROW_BLOCK - 1) / ROW_BLOCK, T = host_threads();
    std::vector<std::vector<int>> bci(nb), markers(T);
    std::vector<std::vector<double>> bv(nb), accs(T);
    for_row_blocks(n, [&](int blk, int r0, int r1, int th) {
      if (markers[th].empty())
      {
        markers[th].assign(n1, -1);
        accs[th].assign(n1, 0.0);
      }
      std::vector<int>& mk = markers[th];
      std::vector<double>& ac = accs[th];
      std::vector<int> cols;
      for (int i = r0; i < r1; ++i)
      {
        cols.clear();
        auto add = [&](int c, double val) {
          if (mk[c] != i)
          {
            mk[c] = i;
            ac[c] = 0.0;
            cols.push_back(c);
          }
          ac[c] += val;
        };
        if (tcol[i] > 0.5)
        {
          add((int)(tcol[i] - 0.5), tval[i]);
          for (int e = A0.rp[i]; e < A0.rp[i + 1]; ++e)
          {
            const int j = A0.ci[e];
            if (tcol[j] > 0.5)
              add((int)(tcol[j] - 0.5), -omega * A0.v[e] / d[i] * tval[j]);
          }
        }
        std::sort(cols.begin(), cols.end());
        int len = 0;
        for (int c : cols)
          if (ac[c] != 0.0)
          {
            bci[blk].push_back(c);
            bv[blk].push_back(ac[c]);
            ++len;
          }
        Pl.rp[i + 1] = len;
      }
    });
    assemble_blocks(Pl, bci, bv);
  }
  tm.lap("smoothed prolongator (owned rows)");

  // (D) rows of P_0 of the ghost dofs: padded (column + 1, value) tables through the layout's forward scatter
  int wloc = 0, Wp = 0;
  for (int i = 0; i < n; ++i)
    wloc = std::max(wloc, Pl.rp[i + 1] - Pl.rp[i]);
  rc = allreduce_max_small(layout, wloc, &Wp, s);
  HostCsr Pext; // [total x n1]: owned rows, then the ghosts' rows
  Pext.n = total;
  Pext.m = n1;
  Pext.rp.assign(total + 1, 0);
  {
    std::vector<double> gcols((size_t)ng * std::max(Wp, 1), 0.0), gvals((size_t)ng * std::max(Wp, 1), 0.0);
    std::vector<double> h(total, 0.0);
    for (int k = 0; k < Wp && rc == PMG_OK; ++k)
      for (int what = 0; what < 2 && rc == PMG_OK; ++what)
      {
        for (int i = 0; i < n; ++i)
        {
          const int e = Pl.rp[i] + k;
          h[i] = e < Pl.rp[i + 1] ? (what == 0 ? (double)(Pl.ci[e] + 1) : Pl.v[e]) : 0.0;
        }
        rc = host_scatter_fwd(layout, h, dv, s);
        for (int g = 0; g < ng && rc == PMG_OK; ++g)
          (what == 0 ? gcols : gvals)[(size_t)g * Wp + k] = h[n + g];
      }
    (void)hipFree(dv);
    PMG_TRY(rc);
    Pext.ci = Pl.ci;
    Pext.v = Pl.v;
    for (int i = 0; i < n; ++i)
      Pext.rp[i + 1] = Pl.rp[i + 1];
    for (int g = 0; g < ng; ++g)
    {
      for (int k = 0; k < Wp; ++k)
        if (gcols[(size_t)g * Wp + k] > 0.5)
        {
          const int c = (int)(gcols[(size_t)g * Wp + k] - 0.5);
          PMG_REQUIRE(c >= 0 && c < n1, "pmg_amg_create_distributed: a ghost's prolongator row names aggregate %d", c);
          Pext.ci.push_back(c);
          Pext.v.push_back(gvals[(size_t)g * Wp + k]);
        }
      Pext.rp[n + g + 1] = (int)Pext.ci.size();
    }
  }
  tm.lap("prolongator rows of the ghosts (one layer of overlap)");

  // (E) this rank's SHARE of A_1 = P_0^T A_0 P_0: sum over its owned fine rows i of P_i^T (A_0 P_0)_i -- entries in the
  // rows of its own aggregates and, along the interfaces, of its neighbours'
  HostCsr A1part = spgemm(transpose(Pl), spgemm(A0, Pext)); // [n1 x n1], non-empty only in the rows this rank touches
  tm.lap("Galerkin product of the owned rows");

  // (F) level 1 on every rank = the sum of the shares: (row * n1 + column, value) pairs in the rank's own segment of one
  // long array (segments ordered by the rank's smallest owned global dof number: no rank numbers needed), summed over
  // the ranks, then assembled row by row
  HostCsr A1;
  {
    const long long cnt = A1part.nnz();
    int64_t mykey = INT64_MAX;
    for (int i = 0; i < n; ++i)
      mykey = std::min(mykey, global_index[i]);
    std::fill(marker.begin(), marker.end(), 0.0);
    PMG_REQUIRE(n == 0 || mykey != INT64_MAX, "pmg_amg_create_distributed: no owned dof");
    if (n > 0)
      marker[(size_t)mykey] = (double)cnt + 0.25; // + 0.25: a rank with an empty share still marks its key
    PMG_TRY(host_allreduce_sum(layout, marker.data(), marker.size(), s));
    long long offset = 0, totalcnt = 0;
    for (int64_t g = 0; g < n_global; ++g)
      if (marker[(size_t)g] > 0.0)
      {
        const long long c = (long long)marker[(size_t)g];
        if (g < mykey)
          offset += c;
        totalcnt += c;
      }
    std::vector<double>().swap(marker);
    std::vector<double> keys((size_t)totalcnt, 0.0), vals((size_t)totalcnt, 0.0);
    {
      long long o = offset;
      for (int I = 0; I < n1; ++I)
        for (int e = A1part.rp[I]; e < A1part.rp[I + 1]; ++e, ++o)
        {
          keys[(size_t)o] = (double)((long long)I * n1 + A1part.ci[e]) + 1.0; // + 1: 0 = empty
          vals[(size_t)o] = A1part.v[e];
        }
    }
    A1part = HostCsr();
    PMG_TRY(host_allreduce_sum(layout, keys.data(), keys.size(), s));
    PMG_TRY(host_allreduce_sum(layout, vals.data(), vals.size(), s));
    // bucket by row, then sort each row by column and add duplicates
    std::vector<int> rstart(n1 + 1, 0);
    for (long long o = 0; o < totalcnt; ++o)
    {
      PMG_REQUIRE(keys[(size_t)o] > 0.5, "pmg_amg_create_distributed: two ranks share a segment of the gather");
      rstart[(int)(((long long)(keys[(size_t)o] - 0.5)) / n1) + 1]++;
    }
    for (int I = 0; I < n1; ++I)
      rstart[I + 1] += rstart[I];
    std::vector<int> ecol((size_t)totalcnt);
    std::vector<double> eval((size_t)totalcnt);
    {
      std::vector<int> pos(rstart.begin(), rstart.end() - 1);
      for (long long o = 0; o < totalcnt; ++o)
      {
        const long long k = (long long)(keys[(size_t)o] - 0.5);
        const int I = (int)(k / n1), J = (int)(k - (long long)I * n1);
        const int p = pos[I]++;
        ecol[(size_t)p] = J;
        eval[(size_t)p] = vals[(size_t)o];
      }
    }
    std::vector<double>().swap(keys);
    std::vector<double>().swap(vals);
    A1.n = A1.m = n1;
    A1.rp.assign(n1 + 1, 0);
    const int nbk = (n1 + ROW_BLOCK - 1) / ROW_BLOCK;
    std::vector<std::vector<int>> bci(nbk);
    std::vector<std::vector<double>> bv(nbk);
    for_row_blocks(n1, [&](int blk, int r0, int r1, int) {
      std::vector<std::pair<int, double>> row;
      for (int I = r0; I < r1; ++I)
      {
        row.clear();
        for (int p = rstart[I]; p < rstart[I + 1]; ++p)
          row.emplace_back(ecol[(size_t)p], eval[(size_t)p]);
        // (stable order of equal columns = order of the ranks' segments: the same sums on every rank)
        std::stable_sort(row.begin(), row.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        int len = 0;
        for (size_t q = 0; q < row.size();)
        {
          double v = 0.0;
          size_t r = q;
          for (; r < row.size() && row[r].first == row[q].first; ++r)
            v += row[r].second;
          bci[blk].push_back(row[q].first);
          bv[blk].push_back(v);
          ++len;
          q = r;
        }
        A1.rp[I + 1] = len;
      }
    });
    assemble_blocks(A1, bci, bv);
  }
  tm.lap("gather of level 1 (sum of the ranks' shares)");
  PMG_TRY(build_hierarchy(amg, std::move(A1), (size_t)n1, 0.04)); // level 1's threshold: half of level 0's 0.08
  // (build_hierarchy leaves the solution vector of ITS first level to the caller; here that level is level 1 of the
  // solve and needs one of its own)
  PMG_TRY(alloc_d(&amg->levels[0].x, (size_t)n1));
  // level 0 in front of it: lives on the layout, has no assembled matrix on the device
  AmgLevel l0;
  l0.n = n;
  l0.nnz = A0.nnz();
  l0.lmax = rho0;
  amg->levels.insert(amg->levels.begin(), l0);
  amg->hA.insert(amg->hA.begin(), HostCsr());
  amg->hP.insert(amg->hP.begin(), Pl);
  amg->hlmax.insert(amg->hlmax.begin(), rho0);

  PMG_TRY(upload_csr(amg->P0l, Pl));
  PMG_TRY(upload_csr(amg->R0l, transpose(Pl)));
  PMG_HIP(hipHostMalloc(&amg->h_stage, sizeof(double) * std::max<size_t>((size_t)std::max(n1, 1), 1), hipHostMallocDefault));
  PMG_TRY(pmg_layout_create(&amg->glayout, (int32_t)n1, 0, 0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                            nullptr));
  PMG_TRY(pmg_cg_create(&amg->cg, amg->glayout));
  PMG_TRY(pmg_cg_create(&amg->cg0, layout));
  amg->hA[0] = std::move(A0);
  amg->dist0 = true;
  tm.lap("hierarchy below level 1 + uploads");
  return PMG_OK;
}

extern "C" int pmg_amg_create_distributed(pmg_amg* out, pmg_laplacian op, const int64_t* global_index,
                                          int64_t n_global, pmg_stream stream)
{
  PMG_REQUIRE(global_index, "pmg_amg_create_distributed: NULL global index");
  return amg_create(out, op, global_index, n_global, stream, true);
}

extern "C" int pmg_amg_destroy(pmg_amg amg)
{
  if (!amg)
    return PMG_OK;
  for (AmgLevel& lv : amg->levels)
  {
    free_csr(lv.A);
    free_csr(lv.P);
    free_csr(lv.R);
    (void)hipFree(lv.dinv);
    (void)hipFree(lv.x);
    (void)hipFree(lv.b);
    (void)hipFree(lv.r);
    (void)hipFree(lv.z);
    (void)hipFree(lv.q);
  }
  (void)hipFree(amg->dense_inv);
  (void)hipFree(amg->xc);
  (void)hipFree(amg->gid);
  (void)hipFree(amg->d0_dinv);
  (void)hipFree(amg->gb);
  (void)hipFree(amg->gx);
  if (amg->h_stage)
    (void)hipHostFree(amg->h_stage);
  free_csr(amg->P0l);
  free_csr(amg->R0l);
  (void)hipFree(amg->d0_r);
  (void)hipFree(amg->d0_z);
  (void)hipFree(amg->d0_q);
  (void)hipFree(amg->d0_b);
  (void)hipFree(amg->d0_xc);
  pmg_cg_destroy(amg->cg0);
  pmg_cg_destroy(amg->cg);
  pmg_layout_destroy(amg->glayout);
  delete amg;
  return PMG_OK;
}

extern "C" int pmg_amg_set_smoother_iterations(pmg_amg amg, int k)
{
  PMG_REQUIRE(amg && k >= 1, "pmg_amg_set_smoother_iterations: bad argument");
  amg->smoother_its = k;
  return PMG_OK;
}

extern "C" int pmg_amg_set_cycles(pmg_amg amg, int cycles)
{
  PMG_REQUIRE(amg && cycles >= 0, "pmg_amg_set_cycles: bad argument");
  amg->cycles = cycles;
  return PMG_OK;
}

// Replicated form: 1 (default where the hierarchy has a second level) = level 0 stays on the partitioned operator,
// only the levels below are replicated; 0 = the whole hierarchy replicated (round 2's form).
extern "C" int pmg_amg_set_distributed_fine_level(pmg_amg amg, int enable)
{
  PMG_REQUIRE(amg, "pmg_amg_set_distributed_fine_level: NULL argument");
  PMG_REQUIRE(amg->replicated, "pmg_amg_set_distributed_fine_level: not a replicated hierarchy");
  PMG_REQUIRE(!enable || amg->cg0, "pmg_amg_set_distributed_fine_level: the hierarchy has one level only");
  PMG_REQUIRE(enable || !amg->distributed_setup,
              "pmg_amg_set_distributed_fine_level: this hierarchy was set up without gathering its level 0 "
              "(pmg_amg_create_distributed): it cannot be replicated in the solve");
  amg->dist0 = enable != 0;
  return PMG_OK;
}

extern "C" int pmg_amg_set_krylov(pmg_amg amg, int max_iter, double rtol)
{
  PMG_REQUIRE(amg && max_iter >= 0 && rtol >= 0.0, "pmg_amg_set_krylov: bad argument");
  amg->max_iter = max_iter;
  amg->rtol = rtol;
  amg->cycles = 0;
  return PMG_OK;
}

extern "C" int pmg_amg_solve(pmg_amg amg, double* x, const double* b, int* iterations, pmg_stream stream)
{
  PMG_REQUIRE(amg && x && b, "pmg_amg_solve: NULL argument");
  PMG_REQUIRE(x != b, "pmg_amg_solve: x and b alias");
  PMG_TRY(amg_solve(amg, x, b, S(stream)));
  if (iterations)
    *iterations = amg->last_iterations;
  return PMG_OK;
}

extern "C" int pmg_amg_cycle(pmg_amg amg, double* x, const double* b, pmg_stream stream)
{
  PMG_REQUIRE(amg && x && b && x != b, "pmg_amg_cycle: bad argument");
  PMG_REQUIRE(!amg->replicated, "pmg_amg_cycle: a replicated hierarchy works on gathered vectors; use pmg_amg_solve");
  return amg_cycle(amg, 0, x, b, S(stream));
}

extern "C" int pmg_amg_num_levels(pmg_amg amg) { return amg ? (int)amg->levels.size() : -1; }

extern "C" int pmg_amg_level_info(pmg_amg amg, int level, long long* rows, long long* nnz, double* lambda_max)
{
  PMG_REQUIRE(amg && level >= 0 && level < (int)amg->levels.size(), "pmg_amg_level_info: bad level");
  if (rows)
    *rows = amg->levels[level].n;
  if (nnz)
    *nnz = amg->levels[level].nnz;
  if (lambda_max)
    *lambda_max = amg->levels[level].lmax;
  return PMG_OK;
}

// which = 0: A_level, 1: P_level (level -> level + 1).  Call with NULL arrays for the sizes.
extern "C" int pmg_amg_export(pmg_amg amg, int level, int which, long long* rows, long long* cols, long long* nnz,
                              int32_t* rowptr, int32_t* colidx, double* values)
{
  PMG_REQUIRE(amg && level >= 0 && (which == 0 || which == 1), "pmg_amg_export: bad argument");
  const std::vector<HostCsr>& src = which == 0 ? amg->hA : amg->hP;
  PMG_REQUIRE(level < (int)src.size(), "pmg_amg_export: no such level");
  const HostCsr& M = src[level];
  if (rows)
    *rows = M.n;
  if (cols)
    *cols = M.m;
  if (nnz)
    *nnz = M.nnz();
  if (rowptr)
    std::copy(M.rp.begin(), M.rp.end(), rowptr);
  if (colidx)
    std::copy(M.ci.begin(), M.ci.end(), colidx);
  if (values)
    std::copy(M.v.begin(), M.v.end(), values);
  return PMG_OK;
}
