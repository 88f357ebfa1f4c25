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
