// p-transfer between two tensor-product GLL spaces on the same cells.
// Replaces src/interpolate.hpp (interpolate_Q1Q2 :21-45, interpolate_Q2Q1 :60-87,
// Interpolator :93-329).  The reference runs one *thread* per cell over a CSR of
// the dense N_f x N_c cell matrix; here a workgroup takes several cells, one
// thread per fine dof, the cell's values staged in LDS and the cell matrix
// applied as the tensor product of the 1-D table M1 (nd_f x nd_c) it is built
// from (entries with |v| <= 1e-12 dropped, :119-135).
#include "common.hpp"

using namespace pmg;

struct pmg_interpolator_s
{
  pmg_layout lc = nullptr, lf = nullptr;
  int pc = 0, pf = 0, ndc = 0, ndf = 0, Nc = 0, Nf = 0;
  int32_t ncells = 0;
  const int32_t* dmc = nullptr; // caller-owned
  const int32_t* dmf = nullptr;
  double* M1 = nullptr;       // [ndf][ndc]
  double* inv_mult = nullptr; // [fine total], 1/multiplicity (src/interpolate.hpp:172-178)
  int32_t* lcells = nullptr;  // nullptr = identity
  int32_t* bcells = nullptr;
  int32_t n_l = 0, n_b = 0;
  int cpb = 1, threads = 64;
};

namespace
{
__global__ void count_kernel(long long n, const int32_t* __restrict__ dm, double* __restrict__ cnt)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    atomicAdd(&cnt[dm[i]], 1.0);
}

__global__ void invert_kernel(int n, double* __restrict__ v)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    v[i] = v[i] > 0.0 ? 1.0 / v[i] : 0.0;
}

// LDS: M1 [ndf*ndc] | per cell: coarse values [Nc]
__global__ void prolong_kernel(int ncells_list, const int32_t* __restrict__ cells, int cpb, int ndc,
                               int ndf, const int32_t* __restrict__ dmc,
                               const int32_t* __restrict__ dmf, const double* __restrict__ M1g,
                               const double* __restrict__ coarse, double* __restrict__ fine)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int Nc = ndc * ndc * ndc, Nf = ndf * ndf * ndf;
  double* sM = smem;
  double* sc = smem + ndf * ndc;
  const int t = threadIdx.x;
  const int lc = t / Nf, tl = t - lc * Nf;
  const int ci = blockIdx.x * cpb + lc;
  const bool active = lc < cpb && ci < ncells_list;
  if (t < ndf * ndc)
    sM[t] = M1g[t];
  int cell = 0;
  if (active)
  {
    cell = cells ? cells[ci] : ci;
    if (tl < Nc)
      sc[lc * Nc + tl] = coarse[dmc[(size_t)cell * Nc + tl]];
  }
  __syncthreads();
  if (!active)
    return;
  const int a = tl / (ndf * ndf), b = (tl / ndf) % ndf, c = tl % ndf;
  const double* uc = sc + lc * Nc;
  double v = 0.0;
  for (int i = 0; i < ndc; ++i)
  {
    double vi = 0.0;
    for (int j = 0; j < ndc; ++j)
    {
      double vj = 0.0;
      for (int k = 0; k < ndc; ++k)
        vj += sM[c * ndc + k] * uc[(i * ndc + j) * ndc + k];
      vi += sM[b * ndc + j] * vj;
    }
    v += sM[a * ndc + i] * vi;
  }
  fine[dmf[(size_t)cell * Nf + tl]] = v; // plain store, src/interpolate.hpp:42
}

// LDS: M1 [ndf*ndc] | per cell: weighted fine values [Nf]
__global__ void restrict_kernel(int ncells_list, const int32_t* __restrict__ cells, int cpb,
                                int ndc, int ndf, const int32_t* __restrict__ dmc,
                                const int32_t* __restrict__ dmf, const double* __restrict__ M1g,
                                const double* __restrict__ inv_mult,
                                const double* __restrict__ fine, double* __restrict__ coarse)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int Nc = ndc * ndc * ndc, Nf = ndf * ndf * ndf;
  double* sM = smem;
  double* sw = smem + ndf * ndc;
  const int t = threadIdx.x;
  const int lc = t / Nf, tl = t - lc * Nf;
  const int ci = blockIdx.x * cpb + lc;
  const bool active = lc < cpb && ci < ncells_list;
  if (t < ndf * ndc)
    sM[t] = M1g[t];
  int cell = 0;
  if (active)
  {
    cell = cells ? cells[ci] : ci;
    int d = dmf[(size_t)cell * Nf + tl];
    sw[lc * Nf + tl] = fine[d] * inv_mult[d]; // src/interpolate.hpp:81-82
  }
  __syncthreads();
  if (!active || tl >= Nc)
    return;
  const int i = tl / (ndc * ndc), j = (tl / ndc) % ndc, k = tl % ndc;
  const double* w = sw + lc * Nf;
  double v = 0.0;
  for (int a = 0; a < ndf; ++a)
  {
    double va = 0.0;
    for (int b = 0; b < ndf; ++b)
    {
      double vb = 0.0;
      for (int c = 0; c < ndf; ++c)
        vb += sM[c * ndc + k] * w[(a * ndf + b) * ndf + c];
      va += sM[b * ndc + j] * vb;
    }
    v += sM[a * ndc + i] * va;
  }
  atomicAdd(&coarse[dmc[(size_t)cell * Nc + tl]], v); // src/interpolate.hpp:84
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
int interp_prolong(pmg_interpolator ip, double* coarse, double* fine, hipStream_t s)
{
  const size_t shm = sizeof(double) * (ip->ndf * ip->ndc + (size_t)ip->cpb * ip->Nc);
  PMG_TRY(pmg_scatter_fwd_begin(ip->lc, coarse, (pmg_stream)s)); // src/interpolate.hpp:202
  if (ip->n_l > 0)
    prolong_kernel<<<(ip->n_l + ip->cpb - 1) / ip->cpb, ip->threads, shm, s>>>(
        ip->n_l, ip->lcells, ip->cpb, ip->ndc, ip->ndf, ip->dmc, ip->dmf, ip->M1, coarse, fine);
  PMG_TRY(pmg_scatter_fwd_end(ip->lc, coarse, (pmg_stream)s)); // :217
  if (ip->n_b > 0) // own grid size: the reference reuses the interior grid (quirk Q3)
    prolong_kernel<<<(ip->n_b + ip->cpb - 1) / ip->cpb, ip->threads, shm, s>>>(
        ip->n_b, ip->bcells, ip->cpb, ip->ndc, ip->ndf, ip->dmc, ip->dmf, ip->M1, coarse, fine);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

int interp_restrict(pmg_interpolator ip, double* fine, double* coarse, hipStream_t s)
{
  const size_t shm = sizeof(double) * (ip->ndf * ip->ndc + (size_t)ip->cpb * ip->Nf);
  PMG_TRY(pmg_scatter_fwd_begin(ip->lf, fine, (pmg_stream)s));             // :264
  PMG_HIP(hipMemsetAsync(coarse, 0, sizeof(double) * ip->lc->total(), s)); // :270
  if (ip->n_l > 0)
    restrict_kernel<<<(ip->n_l + ip->cpb - 1) / ip->cpb, ip->threads, shm, s>>>(
        ip->n_l, ip->lcells, ip->cpb, ip->ndc, ip->ndf, ip->dmc, ip->dmf, ip->M1, ip->inv_mult,
        fine, coarse);
  PMG_TRY(pmg_scatter_fwd_end(ip->lf, fine, (pmg_stream)s)); // :281
  if (ip->n_b > 0)
    restrict_kernel<<<(ip->n_b + ip->cpb - 1) / ip->cpb, ip->threads, shm, s>>>(
        ip->n_b, ip->bcells, ip->cpb, ip->ndc, ip->ndf, ip->dmc, ip->dmf, ip->M1, ip->inv_mult,
        fine, coarse);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}
} // namespace pmg

extern "C" int pmg_interpolator_create(pmg_interpolator* out, pmg_layout layout_coarse,
                                       pmg_layout layout_fine, int degree_coarse, int degree_fine,
                                       int32_t ncells, const int32_t* dofmap_coarse,
                                       const int32_t* dofmap_fine, const int32_t* lcells,
                                       int32_t n_lcells, const int32_t* bcells, int32_t n_bcells,
                                       pmg_stream stream)
{
  PMG_REQUIRE(out && layout_coarse && layout_fine, "pmg_interpolator_create: NULL handle");
  PMG_REQUIRE(degree_coarse >= 1 && degree_fine > degree_coarse && degree_fine <= PMG_MAX_DEGREE,
              "pmg_interpolator_create: need 1 <= degree_coarse < degree_fine <= %d",
              PMG_MAX_DEGREE);
  PMG_REQUIRE(ncells >= 0 && n_lcells >= 0 && n_bcells >= 0 && n_lcells + n_bcells <= ncells,
              "pmg_interpolator_create: bad cell counts");
  PMG_REQUIRE(ncells == 0 || (dofmap_coarse && dofmap_fine),
              "pmg_interpolator_create: NULL dofmap");
  for (int i = 0; i < n_lcells; ++i)
    PMG_REQUIRE(lcells[i] >= 0 && lcells[i] < ncells,
                "pmg_interpolator_create: lcells out of range");
  for (int i = 0; i < n_bcells; ++i)
    PMG_REQUIRE(bcells[i] >= 0 && bcells[i] < ncells,
                "pmg_interpolator_create: bcells out of range");
  hipStream_t s = S(stream);
  auto* ip = new pmg_interpolator_s;
  ip->lc = layout_coarse;
  ip->lf = layout_fine;
  ip->pc = degree_coarse;
  ip->pf = degree_fine;
  ip->ndc = degree_coarse + 1;
  ip->ndf = degree_fine + 1;
  ip->Nc = ip->ndc * ip->ndc * ip->ndc;
  ip->Nf = ip->ndf * ip->ndf * ip->ndf;
  ip->ncells = ncells;
  ip->dmc = dofmap_coarse;
  ip->dmf = dofmap_fine;
  ip->cpb = 256 / ip->Nf > 0 ? 256 / ip->Nf : 1;
  ip->threads = ((ip->cpb * ip->Nf + 63) / 64) * 64;

  std::vector<double> xc(ip->ndc), wc(ip->ndc), xf(ip->ndf), wf(ip->ndf), M1(ip->ndf * ip->ndc);
  gll_table(ip->ndc, xc.data(), wc.data());
  gll_table(ip->ndf, xf.data(), wf.data());
  lagrange_eval_table(ip->ndc, xc.data(), ip->ndf, xf.data(), M1.data());
  PMG_TRY(upload(&ip->M1, M1.data(), M1.size(), s));

  bool identity = (n_lcells == ncells);
  for (int i = 0; identity && i < n_lcells; ++i)
    identity = (lcells[i] == i);
  ip->n_l = n_lcells;
  ip->n_b = n_bcells;
  if (!identity && n_lcells > 0)
    PMG_TRY(upload(&ip->lcells, lcells, n_lcells, s));
  if (n_bcells > 0)
    PMG_TRY(upload(&ip->bcells, bcells, n_bcells, s));

  // multiplicity of every fine dof over all local cells, ghosts included (:172-178)
  const int nf_total = layout_fine->total();
  PMG_HIP(hipMalloc(&ip->inv_mult, sizeof(double) * (nf_total ? nf_total : 1)));
  PMG_HIP(hipMemsetAsync(ip->inv_mult, 0, sizeof(double) * nf_total, s));
  const long long n = (long long)ncells * ip->Nf;
  if (n > 0)
  {
    long long blocks = (n + 255) / 256;
    count_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(n, dofmap_fine,
                                                                      ip->inv_mult);
    invert_kernel<<<(nf_total + 255) / 256, 256, 0, s>>>(nf_total, ip->inv_mult);
    PMG_HIP(hipGetLastError());
  }
  PMG_HIP(hipStreamSynchronize(s));
  *out = ip;
  return PMG_OK;
}

extern "C" int pmg_interpolator_destroy(pmg_interpolator ip)
{
  if (!ip)
    return PMG_OK;
  (void)hipFree(ip->M1);
  (void)hipFree(ip->inv_mult);
  (void)hipFree(ip->lcells);
  (void)hipFree(ip->bcells);
  delete ip;
  return PMG_OK;
}

extern "C" int pmg_interpolator_interpolate(pmg_interpolator ip, double* coarse, double* fine,
                                            pmg_stream stream)
{
  PMG_REQUIRE(ip && coarse && fine, "pmg_interpolator_interpolate: NULL argument");
  return interp_prolong(ip, coarse, fine, S(stream));
}

extern "C" int pmg_interpolator_reverse_interpolate(pmg_interpolator ip, double* fine,
                                                    double* coarse, pmg_stream stream)
{
  PMG_REQUIRE(ip && coarse && fine, "pmg_interpolator_reverse_interpolate: NULL argument");
  return interp_restrict(ip, fine, coarse, S(stream));
}
