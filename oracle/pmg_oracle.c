/* CPU oracle, C/OpenMP restatement of the p-multigrid hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into or called from the product
 * (pmg-dolfinx_amd/).  Used by tests/ as a second, independent restatement
 * next to oracle/pmg_oracle.py, and by bench.py's cpu_baseline leg (kind
 * "port") to time the same V-cycle on the host cores.
 *
 * Citations are relative to the reference tree Wells-Group/pmg-dolfinx @
 * 2024_08_07.  The reference's own implementation of this path cannot be built
 * here (every header needs dolfinx/basix), so this is a port of
 * the algorithm.  Apart from TQLI (the reference's python_tests/tqli.py fixture,
 * tests/golden) the reference holds no golden vectors for this path -- parity
 * unpinned by the reference itself; it is pinned by the analytic KATs in tests/
 * and cross-checked against the numpy restatement.
 *
 * Layouts follow the reference: dofmap [ncells][N] int32 with local index
 * t = a*nd^2 + b*nd + c (src/laplacian.hpp:173); G [ncells][N][6]
 * (src/laplacian.hpp:99-111); D [nq][nd] row = quadrature point (:198).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXND 9

typedef struct
{
  int P, nd, N, ncells, ndofs;
  const int32_t* dofmap;
  const double* G;
  const double* kappa;
  const int8_t* bc;
  double D[ORC_MAXND * ORC_MAXND];
  double* dinv; /* owned */
  /* cell colouring for the scatter-add (owned): cells of one colour share no dof, so the threads of
   * a colour add without atomics.  This is how a CPU implementation gets a deterministic, contention-free
   * scatter; the reference's GPU kernel uses atomicAdd (src/laplacian.hpp:277), its CPU path assembles. */
  int ncolors;
  int* color_off;   /* [ncolors+1] */
  int* color_cells; /* [ncells], grouped by colour, ascending inside a colour */
} orc_level;

typedef struct
{
  orc_level *lc, *lf;
  int Nc, Nf;
  /* dense cell matrix M [Nf][Nc], |v|<=1e-12 dropped (src/interpolate.hpp:119-135) */
  double* M;
  double* inv_mult; /* 1/multiplicity of each fine dof (src/interpolate.hpp:172-178) */
} orc_interp;

int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
  if (n > 0)
    omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ---- geometry: src/laplacian.hpp:22-113 (detJ by full cofactor expansion) ---- */
void orc_geometry(int ncells, int nq, const double* xgeom, const int32_t* geom_dofmap,
                  const double* dphi /* [3][nq][8] */, const double* w, double* G)
{
#pragma omp parallel for schedule(static)
  for (int c = 0; c < ncells; ++c)
  {
    double xc[8][3];
    for (int k = 0; k < 8; ++k)
      for (int j = 0; j < 3; ++j)
        xc[k][j] = xgeom[3 * (size_t)geom_dofmap[(size_t)c * 8 + k] + j];
    for (int q = 0; q < nq; ++q)
    {
      double J[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
        {
          double s = 0;
          for (int k = 0; k < 8; ++k)
            s += xc[k][i] * dphi[((size_t)j * nq + q) * 8 + k];
          J[i][j] = s;
        }
      double K[3][3]
          = {{J[1][1] * J[2][2] - J[1][2] * J[2][1], -J[0][1] * J[2][2] + J[0][2] * J[2][1],
              J[0][1] * J[1][2] - J[0][2] * J[1][1]},
             {-J[1][0] * J[2][2] + J[1][2] * J[2][0], J[0][0] * J[2][2] - J[0][2] * J[2][0],
              -J[0][0] * J[1][2] + J[0][2] * J[1][0]},
             {J[1][0] * J[2][1] - J[1][1] * J[2][0], -J[0][0] * J[2][1] + J[0][1] * J[2][0],
              J[0][0] * J[1][1] - J[0][1] * J[1][0]}};
      double detJ = J[0][0] * K[0][0] + J[0][1] * K[1][0] + J[0][2] * K[2][0];
      double s = w[q] / detJ;
      double* g = G + ((size_t)c * nq + q) * 6;
      g[0] = (K[0][0] * K[0][0] + K[0][1] * K[0][1] + K[0][2] * K[0][2]) * s;
      g[1] = (K[1][0] * K[0][0] + K[1][1] * K[0][1] + K[1][2] * K[0][2]) * s;
      g[2] = (K[2][0] * K[0][0] + K[2][1] * K[0][1] + K[2][2] * K[0][2]) * s;
      g[3] = (K[1][0] * K[1][0] + K[1][1] * K[1][1] + K[1][2] * K[1][2]) * s;
      g[4] = (K[2][0] * K[1][0] + K[2][1] * K[1][1] + K[2][2] * K[1][2]) * s;
      g[5] = (K[2][0] * K[2][0] + K[2][1] * K[2][1] + K[2][2] * K[2][2]) * s;
    }
  }
}

/* ---- levels ---- */
orc_level* orc_level_create(int P, int ncells, int ndofs, const int32_t* dofmap, const double* G,
                            const double* kappa, const int8_t* bc, const double* D)
{
  if (P < 1 || P + 1 > ORC_MAXND)
    return NULL;
  orc_level* l = (orc_level*)calloc(1, sizeof(orc_level));
  l->P = P;
  l->nd = P + 1;
  l->N = l->nd * l->nd * l->nd;
  l->ncells = ncells;
  l->ndofs = ndofs;
  l->dofmap = dofmap;
  l->G = G;
  l->kappa = kappa;
  l->bc = bc;
  memcpy(l->D, D, sizeof(double) * l->nd * l->nd);
  l->dinv = (double*)malloc(sizeof(double) * ndofs);
  for (int i = 0; i < ndofs; ++i)
    l->dinv[i] = 1.0;
  /* greedy colouring: a cell takes the lowest colour none of its dofs has seen yet (64 colours at most;
   * a structured hex mesh needs 8) */
  {
    uint64_t* used = (uint64_t*)calloc(ndofs > 0 ? ndofs : 1, sizeof(uint64_t));
    int* col = (int*)malloc(sizeof(int) * (ncells > 0 ? ncells : 1));
    int nc = 0, ok = 1;
    for (int c = 0; c < ncells && ok; ++c)
    {
      const int32_t* dm = dofmap + (size_t)c * l->N;
      uint64_t m = 0;
      for (int t = 0; t < l->N; ++t)
        m |= used[dm[t]];
      int k = 0;
      while (k < 64 && ((m >> k) & 1))
        ++k;
      if (k == 64)
      {
        ok = 0;
        break;
      }
      col[c] = k;
      if (k + 1 > nc)
        nc = k + 1;
      for (int t = 0; t < l->N; ++t)
        used[dm[t]] |= (uint64_t)1 << k;
    }
    free(used);
    if (ok)
    {
      l->ncolors = nc;
      l->color_off = (int*)calloc(nc + 1, sizeof(int));
      l->color_cells = (int*)malloc(sizeof(int) * (ncells > 0 ? ncells : 1));
      for (int c = 0; c < ncells; ++c)
        l->color_off[col[c] + 1]++;
      for (int k = 0; k < nc; ++k)
        l->color_off[k + 1] += l->color_off[k];
      int* at = (int*)malloc(sizeof(int) * (nc > 0 ? nc : 1));
      for (int k = 0; k < nc; ++k)
        at[k] = l->color_off[k];
      for (int c = 0; c < ncells; ++c)
        l->color_cells[at[col[c]]++] = c;
      free(at);
    }
    else
      l->ncolors = 0; /* more than 64 colours: the scatter falls back to atomics */
    free(col);
  }
  return l;
}

void orc_level_destroy(orc_level* l)
{
  if (l)
  {
    free(l->dinv);
    free(l->color_off);
    free(l->color_cells);
    free(l);
  }
}

void orc_level_set_diag_inverse(orc_level* l, const double* dinv)
{
  memcpy(l->dinv, dinv, sizeof(double) * l->ndofs);
}

/* one cell of src/laplacian.hpp:143-278 */
static void cell_kernel(const orc_level* l, int c, const double* u, double* out)
{
  const int nd = l->nd, N = l->N;
  const double* D = l->D;
  const double* G = l->G + (size_t)c * N * 6;
  const double kap = l->kappa[c];
  double f0[ORC_MAXND * ORC_MAXND * ORC_MAXND], f1[ORC_MAXND * ORC_MAXND * ORC_MAXND],
      f2[ORC_MAXND * ORC_MAXND * ORC_MAXND];
  for (int a = 0; a < nd; ++a)
    for (int b = 0; b < nd; ++b)
      for (int cc = 0; cc < nd; ++cc)
      {
        double vx = 0, vy = 0, vz = 0;
        for (int i = 0; i < nd; ++i)
        {
          vx += D[a * nd + i] * u[(i * nd + b) * nd + cc];  /* :195-199 */
          vy += D[b * nd + i] * u[(a * nd + i) * nd + cc];  /* :206-210 */
          vz += D[cc * nd + i] * u[(a * nd + b) * nd + i];  /* :214-218 */
        }
        int t = (a * nd + b) * nd + cc;
        const double* g = G + (size_t)t * 6;
        f0[t] = kap * (g[0] * vx + g[1] * vy + g[2] * vz); /* :233 */
        f1[t] = kap * (g[1] * vx + g[3] * vy + g[4] * vz); /* :234 */
        f2[t] = kap * (g[2] * vx + g[4] * vy + g[5] * vz); /* :235 */
      }
  for (int a = 0; a < nd; ++a)
    for (int b = 0; b < nd; ++b)
      for (int cc = 0; cc < nd; ++cc)
      {
        double vx = 0, vy = 0, vz = 0;
        for (int q = 0; q < nd; ++q)
        {
          vx += D[q * nd + a] * f0[(q * nd + b) * nd + cc];  /* :246-251 */
          vy += D[q * nd + b] * f1[(a * nd + q) * nd + cc];  /* :255-259 */
          vz += D[q * nd + cc] * f2[(a * nd + b) * nd + q]; /* :263-267 */
        }
        out[(a * nd + b) * nd + cc] = vx + vy + vz; /* :270 */
      }
}

/* y = A x, src/laplacian.hpp:462-482 + :143-278 (y zeroed, BC cols masked, BC rows y=x) */
void orc_level_apply(const orc_level* l, const double* x, double* y)
{
  const int N = l->N;
  memset(y, 0, sizeof(double) * l->ndofs);
#pragma omp parallel
  {
    double u[ORC_MAXND * ORC_MAXND * ORC_MAXND], out[ORC_MAXND * ORC_MAXND * ORC_MAXND];
    if (l->ncolors > 0)
    {
      /* colour by colour (implicit barrier between colours): plain adds */
      for (int k = 0; k < l->ncolors; ++k)
      {
#pragma omp for schedule(static)
        for (int i = l->color_off[k]; i < l->color_off[k + 1]; ++i)
        {
          const int c = l->color_cells[i];
          const int32_t* dm = l->dofmap + (size_t)c * N;
          for (int t = 0; t < N; ++t)
            u[t] = l->bc[dm[t]] ? 0.0 : x[dm[t]]; /* :186-189 */
          cell_kernel(l, c, u, out);
          for (int t = 0; t < N; ++t)
            if (!l->bc[dm[t]])
              y[dm[t]] += out[t]; /* :277 */
        }
      }
    }
    else
    {
#pragma omp for schedule(static)
      for (int c = 0; c < l->ncells; ++c)
      {
        const int32_t* dm = l->dofmap + (size_t)c * N;
        for (int t = 0; t < N; ++t)
          u[t] = l->bc[dm[t]] ? 0.0 : x[dm[t]]; /* :186-189 */
        cell_kernel(l, c, u, out);
        for (int t = 0; t < N; ++t)
        {
          if (!l->bc[dm[t]])
          {
#pragma omp atomic
            y[dm[t]] += out[t]; /* :277 */
          }
        }
      }
    }
  }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < l->ndofs; ++i)
    if (l->bc[i])
      y[i] = x[i]; /* :273-274 */
}

/* matrix-free diagonal of the BC-treated operator (BC rows = 1); this is what
 * the reference reads off an assembled CSR (src/csr.hpp:100-110) */
void orc_level_diagonal(const orc_level* l, double* diag)
{
  const int nd = l->nd, N = l->N;
  const double* D = l->D;
  memset(diag, 0, sizeof(double) * l->ndofs);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < l->ncells; ++c)
  {
    const int32_t* dm = l->dofmap + (size_t)c * N;
    const double* G = l->G + (size_t)c * N * 6;
    const double kap = l->kappa[c];
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nd; ++b)
        for (int cc = 0; cc < nd; ++cc)
        {
          double s = 0;
          for (int q = 0; q < nd; ++q)
          {
            s += D[q * nd + a] * D[q * nd + a] * G[(size_t)((q * nd + b) * nd + cc) * 6 + 0];
            s += D[q * nd + b] * D[q * nd + b] * G[(size_t)((a * nd + q) * nd + cc) * 6 + 3];
            s += D[q * nd + cc] * D[q * nd + cc] * G[(size_t)((a * nd + b) * nd + q) * 6 + 5];
          }
          int t = (a * nd + b) * nd + cc;
          const double* g = G + (size_t)t * 6;
          s += 2.0 * (g[1] * D[a * nd + a] * D[b * nd + b] + g[2] * D[a * nd + a] * D[cc * nd + cc]
                      + g[4] * D[b * nd + b] * D[cc * nd + cc]);
#pragma omp atomic
          diag[dm[t]] += kap * s;
        }
  }
  for (int i = 0; i < l->ndofs; ++i)
    if (l->bc[i])
      diag[i] = 1.0;
}

/* ---- 4th-kind Chebyshev, src/chebyshev.hpp:46-91.
 * need_r = 1 keeps r = b - A x current on exit (the last apply of the reference
 * loop); need_r = 0 skips that last apply -- x is unchanged by it.
 * x_zero = 1 uses r = b for the initial residual (A 0 = 0 exactly). ---- */
void orc_cheb_solve(const orc_level* l, double lmax, int k, double* x, const double* b, double* r,
                    double* z, double* q, int need_r, int x_zero)
{
  const int n = l->ndofs;
  const double* dinv = l->dinv;
  if (x_zero)
  {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
    {
      r[i] = b[i];
      z[i] = r[i] * dinv[i] * (4.0 / (3.0 * lmax));
    }
  }
  else
  {
    orc_level_apply(l, x, q); /* :56 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
    {
      r[i] = b[i] - q[i];                             /* :57 */
      z[i] = r[i] * dinv[i] * (4.0 / (3.0 * lmax)); /* :67-68 */
    }
  }
  for (int it = 1; it <= k; ++it)
  {
    if (it == k && !need_r)
    {
#pragma omp parallel for schedule(static)
      for (int i = 0; i < n; ++i)
        x[i] += z[i]; /* :73 */
      break;
    }
    orc_level_apply(l, z, q); /* :76 */
    const double c1 = (2.0 * it - 1.0) / (2.0 * it + 3.0);
    const double c2 = (8.0 * it + 4.0) / (2.0 * it + 3.0) / lmax;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
    {
      x[i] += z[i];                            /* :73 */
      r[i] -= q[i];                            /* :77 */
      z[i] = c1 * z[i] + c2 * dinv[i] * r[i]; /* :80-83 */
    }
  }
}

/* ---- transfers, src/interpolate.hpp ---- */
orc_interp* orc_interp_create(orc_level* lc, orc_level* lf, const double* M1 /* [ndf][ndc] */)
{
  orc_interp* it = (orc_interp*)calloc(1, sizeof(orc_interp));
  it->lc = lc;
  it->lf = lf;
  it->Nc = lc->N;
  it->Nf = lf->N;
  const int ndc = lc->nd, ndf = lf->nd;
  it->M = (double*)malloc(sizeof(double) * it->Nc * it->Nf);
  for (int a = 0; a < ndf; ++a)
    for (int b = 0; b < ndf; ++b)
      for (int c = 0; c < ndf; ++c)
        for (int i = 0; i < ndc; ++i)
          for (int j = 0; j < ndc; ++j)
            for (int k = 0; k < ndc; ++k)
            {
              double v = M1[a * ndc + i] * M1[b * ndc + j] * M1[c * ndc + k];
              if (fabs(v) <= 1e-12)
                v = 0.0;
              it->M[(size_t)((a * ndf + b) * ndf + c) * it->Nc + (i * ndc + j) * ndc + k] = v;
            }
  it->inv_mult = (double*)calloc(lf->ndofs, sizeof(double));
  for (size_t e = 0; e < (size_t)lf->ncells * lf->N; ++e)
    it->inv_mult[lf->dofmap[e]] += 1.0;
  for (int i = 0; i < lf->ndofs; ++i)
    it->inv_mult[i] = it->inv_mult[i] > 0 ? 1.0 / it->inv_mult[i] : 0.0;
  return it;
}

void orc_interp_destroy(orc_interp* it)
{
  if (it)
  {
    free(it->M);
    free(it->inv_mult);
    free(it);
  }
}

/* prolongation, src/interpolate.hpp:21-45 (plain store; every sharing cell writes the same value) */
void orc_prolong(const orc_interp* it, const double* coarse, double* fine)
{
  const int Nc = it->Nc, Nf = it->Nf;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < it->lf->ncells; ++c)
  {
    const int32_t* d1 = it->lc->dofmap + (size_t)c * Nc;
    const int32_t* d2 = it->lf->dofmap + (size_t)c * Nf;
    double uc[ORC_MAXND * ORC_MAXND * ORC_MAXND];
    for (int k = 0; k < Nc; ++k)
      uc[k] = coarse[d1[k]];
    for (int j = 0; j < Nf; ++j)
    {
      double v = 0;
      const double* row = it->M + (size_t)j * Nc;
      for (int k = 0; k < Nc; ++k)
        v += row[k] * uc[k];
      fine[d2[j]] = v;
    }
  }
}

/* restriction, src/interpolate.hpp:60-87 (output zeroed first, :270) */
void orc_restrict(const orc_interp* it, const double* fine, double* coarse)
{
  const int Nc = it->Nc, Nf = it->Nf;
  memset(coarse, 0, sizeof(double) * it->lc->ndofs);
  /* cells that share no fine dof share no coarse dof either (every shared vertex / edge / face carries
   * dofs of both degrees): the fine level's colouring serves the coarse scatter */
  const orc_level* lf = it->lf;
  const int colored = lf->ncolors > 0;
  const int nrounds = colored ? lf->ncolors : 1;
#pragma omp parallel
  for (int r = 0; r < nrounds; ++r)
  {
    const int i0 = colored ? lf->color_off[r] : 0, i1 = colored ? lf->color_off[r + 1] : lf->ncells;
#pragma omp for schedule(static)
    for (int i = i0; i < i1; ++i)
    {
      const int c = colored ? lf->color_cells[i] : i;
      const int32_t* d1 = it->lc->dofmap + (size_t)c * Nc;
      const int32_t* d2 = it->lf->dofmap + (size_t)c * Nf;
      double uf[ORC_MAXND * ORC_MAXND * ORC_MAXND];
      for (int k = 0; k < Nf; ++k)
        uf[k] = fine[d2[k]] * it->inv_mult[d2[k]];
      for (int j = 0; j < Nc; ++j)
      {
        double v = 0;
        for (int k = 0; k < Nf; ++k)
          v += it->M[(size_t)k * Nc + j] * uf[k];
        if (colored)
          coarse[d1[j]] += v;
        else
        {
#pragma omp atomic
          coarse[d1[j]] += v;
        }
      }
    }
  }
}

/* ---- lean V-cycle: src/pmg.hpp:56-155 minus the residual recomputations that
 * only feed log lines.  Levels coarse -> fine.  work: per level 5 vectors
 * (u, b, r, z, q) owned by the caller.  y is in/out (initial guess / result). ---- */
void orc_vcycle(int nlevels, orc_level** lv, orc_interp** ip, const double* lmax, int cheb_k,
                double** u, double** b, double** r, double** z, double** q, const double* rhs,
                double* y)
{
  const int L = nlevels;
  for (int i = 0; i < L - 1; ++i)
    memset(u[i], 0, sizeof(double) * lv[i]->ndofs);                /* :63-64 */
  memcpy(u[L - 1], y, sizeof(double) * lv[L - 1]->ndofs);          /* :65 */
  memcpy(b[L - 1], rhs, sizeof(double) * lv[L - 1]->ndofs);        /* :68 */
  for (int i = L - 1; i > 0; --i)
  {
    /* :83 pre-smooth, and r = b - A u of :86-87 comes out of the recurrence */
    orc_cheb_solve(lv[i], lmax[i], cheb_k, u[i], b[i], r[i], z[i], q[i], 1, i < L - 1);
    orc_restrict(ip[i - 1], r[i], b[i - 1]); /* :92 */
  }
  for (int j = 0; j < lv[0]->ndofs; ++j) /* :100-103 */
    if (lv[0]->bc[j])
      b[0][j] = 0.0;
  orc_cheb_solve(lv[0], lmax[0], cheb_k, u[0], b[0], r[0], z[0], q[0], 0, 1); /* :109 */
  for (int i = 0; i < L - 1; ++i)
  {
    orc_prolong(ip[i], u[i], q[i + 1]); /* :123 (q reused as du) */
    const int n = lv[i + 1]->ndofs;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < n; ++j)
      u[i + 1][j] += q[i + 1][j];                                                      /* :129 */
    orc_cheb_solve(lv[i + 1], lmax[i + 1], cheb_k, u[i + 1], b[i + 1], r[i + 1], z[i + 1], q[i + 1],
                   0, 0); /* :138 */
  }
  memcpy(y, u[L - 1], sizeof(double) * lv[L - 1]->ndofs); /* :154 */
}

/* ---- TQLI, src/cg.hpp:15-84 ---- */
int orc_tqli(double* d, double* e, int n)
{
  for (int l = 0; l < n; l++)
  {
    int iter = 0;
    for (;;)
    {
      int m;
      for (m = l; m < n - 1; m++)
      {
        double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) + dd == dd)
          break;
      }
      if (m == l)
        break;
      if (iter++ == 30)
        return -1;
      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = sqrt(g * g + 1.0);
      g = d[m] - d[l] + e[l] / (g >= 0 ? g + r : g - r);
      double p = 0.0, s = 1.0, c = 1.0;
      int i, early = 0;
      for (i = m - 1; i >= l; i--)
      {
        double f = s * e[i];
        double b = c * e[i];
        r = sqrt(f * f + g * g);
        e[i + 1] = r;
        if (r == 0.0)
        {
          d[i + 1] -= p;
          e[m] = 0.0;
          early = 1;
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
  return 0;
}
