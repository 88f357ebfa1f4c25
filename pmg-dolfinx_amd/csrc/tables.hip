// Host-side 1-D tables: what the reference obtains from basix at construction
// time (src/laplacian.hpp:302-317, src/interpolate.hpp:118), restated from the
// published definitions (GLL rule, Lagrange basis on the GLL nodes).  Also the
// error plumbing and the TQLI eigenvalue routine (src/cg.hpp:15-84).
#include "common.hpp"

#include <cmath>
#include <dlfcn.h>

namespace pmg
{
thread_local std::string g_last_error;

// roctx, bound at run time (an optional dependency: without libroctx64 the ranges are no-ops)
namespace
{
struct Roctx
{
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx()
  {
    void* h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_NOLOAD);
    const char* names[] = {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so.4"};
    for (int i = 0; !h && i < 3; ++i)
      h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h)
      return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop)
      push = nullptr, pop = nullptr;
  }
};
Roctx& roctx()
{
  static Roctx r;
  return r;
}
} // namespace

void range_push(const char* name)
{
  if (roctx().push)
    roctx().push(name);
}
void range_pop()
{
  if (roctx().pop)
    roctx().pop();
}

int fail(int code, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// Legendre P_m(x) and P_{m-1}(x) by the three-term recurrence
static void legendre_pair(int m, long double x, long double& pm, long double& pm1)
{
  long double p0 = 1.0L, p1 = x;
  if (m == 0)
  {
    pm = 1.0L;
    pm1 = 0.0L;
    return;
  }
  for (int k = 2; k <= m; ++k)
  {
    long double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    p0 = p1;
    p1 = p2;
  }
  pm = p1;
  pm1 = p0;
}

// n-point Gauss-Lobatto-Legendre rule on [0,1]: nodes -1, +1 and the roots of
// P'_{n-1}; weights 2 / (n (n-1) P_{n-1}(x)^2).
void gll_table(int n, double* x, double* w)
{
  const int m = n - 1;
  const long double pi = 3.141592653589793238462643383279502884L;
  std::vector<long double> xi(n);
  for (int i = 0; i < n; ++i)
  {
    long double t = -cosl(pi * i / m); // Chebyshev-Gauss-Lobatto start
    if (i != 0 && i != m)
    {
      for (int it = 0; it < 100; ++it)
      {
        long double pm, pm1;
        legendre_pair(m, t, pm, pm1);
        // (1-t^2) P'_m = m (P_{m-1} - t P_m);  d/dt[(1-t^2) P'_m] = -m (m+1) P_m
        long double f = m * (pm1 - t * pm);
        long double df = -(long double)m * (m + 1) * pm;
        long double dt = f / df;
        t -= dt;
        if (fabsl(dt) < 1e-19L)
          break;
      }
    }
    xi[i] = t;
  }
  xi[0] = -1.0L;
  xi[m] = 1.0L;
  for (int i = 0; i < n; ++i)
  {
    long double s = 0.5L * (xi[i] - xi[m - i]); // exact symmetry
    long double pm, pm1;
    legendre_pair(m, s, pm, pm1);
    x[i] = (double)(0.5L * (s + 1.0L));
    w[i] = (double)(1.0L / ((long double)n * m * pm * pm));
  }
}

// D[q*n + i] = l_i'(x_q), barycentric form
void lagrange_derivative_table(int n, const double* x, double* D)
{
  std::vector<long double> bw(n, 1.0L);
  for (int j = 0; j < n; ++j)
    for (int k = 0; k < n; ++k)
      if (k != j)
        bw[j] /= ((long double)x[j] - (long double)x[k]);
  for (int q = 0; q < n; ++q)
  {
    long double diag = 0.0L;
    for (int i = 0; i < n; ++i)
    {
      if (i == q)
        continue;
      long double v = (bw[i] / bw[q]) / ((long double)x[q] - (long double)x[i]);
      D[q * n + i] = (double)v;
      diag -= v;
    }
    D[q * n + q] = (double)diag;
  }
}

// M[j*nc + k] = l^c_k(xf_j); coinciding nodes give exact 0/1
void lagrange_eval_table(int nc, const double* xc, int nf, const double* xf, double* M)
{
  for (int j = 0; j < nf; ++j)
    for (int k = 0; k < nc; ++k)
    {
      long double v = 1.0L;
      for (int m = 0; m < nc; ++m)
        if (m != k)
          v *= ((long double)xf[j] - (long double)xc[m]) / ((long double)xc[k] - (long double)xc[m]);
      // the reference drops |v| <= 1e-12 (src/interpolate.hpp:119-135)
      M[j * nc + k] = fabsl(v) <= 1e-12L ? 0.0 : (double)v;
    }
}

// ---- cell-local node order (pmg_amd.h) ----
int node_permutation(int node_order, int degree, const int32_t* custom, std::vector<int32_t>& perm1d)
{
  PMG_REQUIRE(degree >= 1 && degree <= 63, "node order: bad degree %d", degree);
  const int nd = degree + 1;
  perm1d.resize(nd);
  switch (node_order)
  {
  case PMG_NODES_ASCENDING:
    for (int j = 0; j < nd; ++j)
      perm1d[j] = j;
    break;
  case PMG_NODES_ENDPOINTS_FIRST: // basix interval element: vertex 0, vertex 1, interior left to right
    perm1d[0] = 0;
    perm1d[1] = nd - 1;
    for (int j = 2; j < nd; ++j)
      perm1d[j] = j - 1;
    break;
  case PMG_NODES_CUSTOM:
  {
    PMG_REQUIRE(custom, "node order: PMG_NODES_CUSTOM needs a permutation");
    std::vector<char> seen(nd, 0);
    for (int j = 0; j < nd; ++j)
    {
      PMG_REQUIRE(custom[j] >= 0 && custom[j] < nd && !seen[custom[j]],
                  "node order: perm1d is not a permutation of 0..%d (entry %d = %d)", nd - 1, j, custom[j]);
      seen[custom[j]] = 1;
      perm1d[j] = custom[j];
    }
    break;
  }
  default:
    return fail(PMG_ERR_INVALID, "node order: unknown order %d", node_order);
  }
  return PMG_OK;
}

std::vector<int32_t> cell_permutation(int nd, const std::vector<int32_t>& perm1d)
{
  std::vector<int32_t> p3((size_t)nd * nd * nd);
  for (int a = 0; a < nd; ++a)
    for (int b = 0; b < nd; ++b)
      for (int c = 0; c < nd; ++c)
        p3[(a * nd + b) * nd + c] = (perm1d[a] * nd + perm1d[b]) * nd + perm1d[c];
  return p3;
}

namespace
{
template <typename T>
__global__ void permute_rows_kernel(long long total, int n, int width, const int32_t* __restrict__ perm,
                                    const T* __restrict__ in, T* __restrict__ out)
{
  const long long rw = (long long)n * width;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x)
  {
    const long long row = i / rw;
    const int r = (int)(i - row * rw), t = r / width, k = r - t * width;
    out[row * rw + (long long)perm[t] * width + k] = in[i];
  }
}
template <typename T>
int permute_rows(long long nrows, int n, int width, const int32_t* perm_d, const T* in, T* out, hipStream_t s)
{
  const long long total = nrows * n * width;
  if (total <= 0)
    return PMG_OK;
  const long long blocks = (total + 255) / 256;
  permute_rows_kernel<T><<<(unsigned)(blocks > 16384 ? 16384 : blocks), 256, 0, s>>>(total, n, width, perm_d, in, out);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}
} // namespace
int permute_rows_i32(long long nrows, int n, const int32_t* perm_d, const int32_t* in, int32_t* out, hipStream_t s)
{
  return permute_rows<int32_t>(nrows, n, 1, perm_d, in, out, s);
}
int permute_rows_f64(long long nrows, int n, int width, const int32_t* perm_d, const double* in, double* out,
                     hipStream_t s)
{
  return permute_rows<double>(nrows, n, width, perm_d, in, out, s);
}
} // namespace pmg

using namespace pmg;

extern "C" const char* pmg_last_error(void) { return g_last_error.c_str(); }
extern "C" int pmg_version(void) { return 100; }

extern "C" int pmg_gll_table(int n, double* points, double* weights)
{
  PMG_REQUIRE(n >= 2 && n <= 64 && points && weights, "pmg_gll_table: need 2 <= n <= 64");
  gll_table(n, points, weights);
  return PMG_OK;
}

extern "C" int pmg_lagrange_derivative_table(int n, double* D)
{
  PMG_REQUIRE(n >= 2 && n <= 64 && D, "pmg_lagrange_derivative_table: need 2 <= n <= 64");
  std::vector<double> x(n), w(n);
  gll_table(n, x.data(), w.data());
  lagrange_derivative_table(n, x.data(), D);
  return PMG_OK;
}

extern "C" int pmg_interpolation_table(int p_coarse, int p_fine, double* M)
{
  PMG_REQUIRE(p_coarse >= 1 && p_fine >= 1 && p_coarse <= 63 && p_fine <= 63 && M,
              "pmg_interpolation_table: bad degrees");
  std::vector<double> xc(p_coarse + 1), wc(p_coarse + 1), xf(p_fine + 1), wf(p_fine + 1);
  gll_table(p_coarse + 1, xc.data(), wc.data());
  gll_table(p_fine + 1, xf.data(), wf.data());
  lagrange_eval_table(p_coarse + 1, xc.data(), p_fine + 1, xf.data(), M);
  return PMG_OK;
}

extern "C" int pmg_node_permutation(int node_order, int degree, const int32_t* custom_perm1d, int32_t* perm1d)
{
  PMG_REQUIRE(perm1d, "pmg_node_permutation: NULL output");
  std::vector<int32_t> p;
  PMG_TRY(node_permutation(node_order, degree, custom_perm1d, p));
  for (size_t j = 0; j < p.size(); ++j)
    perm1d[j] = p[j];
  return PMG_OK;
}

// The tables as the caller's library (basix) would return them: entry (caller row q, caller column i) =
// ascending entry (perm[q], perm[i]).
extern "C" int pmg_gll_table_ordered(int n, int node_order, const int32_t* custom_perm1d, double* points,
                                     double* weights)
{
  PMG_REQUIRE(n >= 2 && n <= 64 && points && weights, "pmg_gll_table_ordered: need 2 <= n <= 64");
  std::vector<int32_t> p;
  PMG_TRY(node_permutation(node_order, n - 1, custom_perm1d, p));
  std::vector<double> x(n), w(n);
  gll_table(n, x.data(), w.data());
  for (int j = 0; j < n; ++j)
  {
    points[j] = x[p[j]];
    weights[j] = w[p[j]];
  }
  return PMG_OK;
}

extern "C" int pmg_lagrange_derivative_table_ordered(int n, int node_order, const int32_t* custom_perm1d, double* D)
{
  PMG_REQUIRE(n >= 2 && n <= 64 && D, "pmg_lagrange_derivative_table_ordered: need 2 <= n <= 64");
  std::vector<int32_t> p;
  PMG_TRY(node_permutation(node_order, n - 1, custom_perm1d, p));
  std::vector<double> x(n), w(n), Da((size_t)n * n);
  gll_table(n, x.data(), w.data());
  lagrange_derivative_table(n, x.data(), Da.data());
  for (int q = 0; q < n; ++q)
    for (int i = 0; i < n; ++i)
      D[q * n + i] = Da[(size_t)p[q] * n + p[i]];
  return PMG_OK;
}

extern "C" int pmg_interpolation_table_ordered(int p_coarse, int p_fine, int node_order, const int32_t* custom_coarse,
                                               const int32_t* custom_fine, double* M)
{
  PMG_REQUIRE(p_coarse >= 1 && p_fine >= 1 && p_coarse <= 63 && p_fine <= 63 && M,
              "pmg_interpolation_table_ordered: bad degrees");
  std::vector<int32_t> pc, pf;
  PMG_TRY(node_permutation(node_order, p_coarse, custom_coarse, pc));
  PMG_TRY(node_permutation(node_order, p_fine, custom_fine, pf));
  const int nc = p_coarse + 1, nf = p_fine + 1;
  std::vector<double> Ma((size_t)nf * nc);
  PMG_TRY(pmg_interpolation_table(p_coarse, p_fine, Ma.data()));
  for (int j = 0; j < nf; ++j)
    for (int k = 0; k < nc; ++k)
      M[j * nc + k] = Ma[(size_t)pf[j] * nc + pc[k]];
  return PMG_OK;
}

// QL-implicit, src/cg.hpp:15-84
extern "C" int pmg_tqli(double* d, double* e, int n)
{
  PMG_REQUIRE(d && e && n >= 1, "pmg_tqli: bad arguments");
  for (int l = 0; l < n; l++)
  {
    int iter = 0;
    for (;;)
    {
      int m;
      for (m = l; m < n - 1; m++)
      {
        double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) + dd == dd)
          break;
      }
      if (m == l)
        break;
      if (iter++ == 30)
        return fail(PMG_ERR_NUMERIC, "pmg_tqli: no convergence after 30 sweeps");
      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = std::sqrt(g * g + 1.0);
      g = d[m] - d[l] + e[l] / (g >= 0 ? g + r : g - r);
      double p = 0.0, s = 1.0, c = 1.0;
      bool early = false;
      for (int i = m - 1; i >= l; i--)
      {
        double f = s * e[i];
        double b = c * e[i];
        r = std::sqrt(f * f + g * g);
        e[i + 1] = r;
        if (r == 0.0)
        {
          d[i + 1] -= p;
          e[m] = 0.0;
          early = true;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + 2.0 * c * b;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - b;
      }
      if (early)
        continue;
      d[l] -= p;
      e[l] = g;
      e[m] = 0.0;
    }
    e[l] = 0.0;
  }
  return PMG_OK;
}
