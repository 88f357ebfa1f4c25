// p-transfer between two tensor-product GLL spaces on the same cells.
// Replaces src/interpolate.hpp (interpolate_Q1Q2 :21-45, interpolate_Q2Q1 :60-87,
// Interpolator :93-329).  The reference runs one *thread* per cell over a CSR of
// the dense N_f x N_c cell matrix; here a workgroup takes several cells, one
// thread per fine dof, the cell's values staged in LDS and the cell matrix
// applied as the tensor product of the 1-D table M1 (nd_f x nd_c) it is built
// from (entries with |v| <= 1e-12 dropped, :119-135).
#include "common.hpp"
#include "patches.hpp"

#include <algorithm>
#include <cstdlib>
#include <map>

using namespace pmg;

namespace pmg
{
PatchView laplacian_patches(pmg_laplacian op);
pmg_layout laplacian_layout(pmg_laplacian op);
}

struct pmg_interpolator_s
{
  pmg_layout lc = nullptr, lf = nullptr;
  int pc = 0, pf = 0, ndc = 0, ndf = 0, Nc = 0, Nf = 0;
  int32_t ncells = 0;
  const int32_t* dmc = nullptr; // caller-owned (ascending node order), or the two copies below
  const int32_t* dmf = nullptr;
  int32_t* dmc_own = nullptr;   // ascending copies of dofmaps given in another cell-local node order
  int32_t* dmf_own = nullptr;   // (pmg_interpolator_create_ordered)
  double* M1 = nullptr;       // [ndf][ndc]
  double* inv_mult = nullptr; // [fine total], 1/multiplicity (src/interpolate.hpp:172-178)
  int32_t* lcells = nullptr;  // nullptr = identity
  int32_t* bcells = nullptr;
  int32_t n_l = 0, n_b = 0;
  int cpb = 1, threads = 64;
  // ---- patch path: shares the cell patches (grouping, colours, launch order,
  // fine dof lists and local maps) of the fine-level operator
  bool patched = false;
  PatchView fv;
  int32_t* cpoff = nullptr;    // [npatch+1] coarse dof lists of the same patches
  uint32_t* cpdofs = nullptr;  // sorted coarse dofs, PD_ACC = an earlier launch wrote it
  int32_t* clmap_id = nullptr; // [npatch]
  uint16_t* clmaps = nullptr;  // [table][K*Nc] position of (slot, coarse local dof)
  uint8_t* pmult = nullptr;    // [fine pdofs entries] multiplicity of the fine dof (:172-178)
  int cmax_m = 0;
  int pwaves = 4;
  size_t pshm = 0;
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

// ---------------------------------------------------------------------------
// Patch form of the two transfers.  One workgroup per patch of the fine-level
// operator; each wavefront takes whole cells and applies the cell matrix as three
// 1-D contractions through a wave-private LDS scratch (nothing but a compiler fence
// between the stages: LDS executes a wave's instructions in order).
//   prolongation: every patch dof is written once, by the patch that touches it
//     first in launch order -- so the correction u += P u_c needs no temporary and
//     all patches run in one launch;
//   restriction: contributions are summed in LDS; a patch adds its sums to the coarse
//     vector with one atomic per patch coarse dof (long runs: a few dozen 64-byte
//     requests per patch; the reference issues one FP64 atomicAdd per (cell, coarse
//     dof), src/interpolate.hpp:84).  A coloured write-back like the operator's is
//     available in the kernel (atomic_out = 0).
__device__ __forceinline__ void tfence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void tbarrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ int ftab(int ndf, int a, int b, int c) // layer-major tables, patches.hpp
{
  return c * ndf * ndf + a * ndf + b;
}

struct TransferArgs
{
  int first, ndc, ndf, K, max_mf, max_mc;
  int nt; // stream the fine vector (nt loads): its level does not fit the MALL anyway
  const int32_t *poff, *lmap_id, *pncell, *cpoff, *clmap_id;
  const uint32_t *pdofs, *cpdofs;
  const uint16_t *lmaps, *clmaps;
  const uint8_t* pmult;
  const double* M1;
};

// Round 4, from the in-kernel stamps of the operator (profiles/kernel_tuning_r04.md): under load every DEPENDENT global
// load costs ~2 us, also when it hits in L2, so what a workgroup needs at its end is requested at its start --
// the prolongation re-reads nothing behind its contraction (the fine dof list, and with `add` the fine values it
// adds to, are in registers by then), the restriction has its coarse list in registers.  Fine cells of at most 32
// dofs (degree 2) are taken TWO per wavefront pass, one per half-wavefront: 54 of 64 lanes busy instead of 27.
constexpr int transfer_cpw(int ndf) { return ndf * ndf * ndf <= 32 ? 2 : 1; } // cells per wavefront pass
constexpr int TRANSFER_LIST_ITER = 6; // list entries per thread held in registers (6 x 512 threads >= any patch list)

template <int NDC, int NDF>
__global__ void prolong_patch_kernel(TransferArgs A, const double* __restrict__ coarse,
                                     double* __restrict__ fine, int add)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int ndc = NDC, ndf = NDF, Nc = ndc * ndc * ndc, Nf = ndf * ndf * ndf;
  constexpr int n1 = ndf * ndc * ndc, n2 = ndf * ndf * ndc;
  constexpr int CPW = transfer_cpw(NDF), HL = 64 / CPW; // cells per wavefront pass, lanes per cell
  constexpr int IT = TRANSFER_LIST_ITER;
  double* sM = smem;                  // [ndf*ndc]
  double* sc = sM + ndf * ndc;        // [max_mc] coarse values of the patch
  double* sf = sc + A.max_mc;         // [max_mf] fine values of the patch
  double* scratch = sf + A.max_mf;    // per wave and cell of the pass: uc[Nc] t1[n1] t2[n2]
  const int p = A.first + blockIdx.x, t = threadIdx.x, nthr = blockDim.x;
  const int off = A.poff[p], Mf = A.poff[p + 1] - off;
  const int coff = A.cpoff[p], Mc = A.cpoff[p + 1] - coff;
  const int nc = A.pncell[p];
  // what the write-back needs, requested first: the fine list ...
  uint32_t m[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k)
  {
    const int i = t + k * nthr;
    m[k] = A.pdofs[off + (i < Mf ? i : Mf - 1)];
  }
  for (int i = t; i < ndf * ndc; i += nthr)
    sM[i] = A.M1[i];
  for (int i = t; i < Mc; i += nthr)
    sc[i] = coarse[A.cpdofs[coff + i] & PD_MASK];
  // ... and the values the correction is added to (they arrive under the contraction)
  double v[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k)
    v[k] = !add ? 0.0 : A.nt ? __builtin_nontemporal_load(fine + (m[k] & PD_MASK)) : fine[m[k] & PD_MASK];
  tbarrier();
  const int wave = t >> 6, lane = t & 63, nw = nthr >> 6;
  const int half = lane / HL, ll = lane - half * HL; // the lane's cell of the pass, its lane inside the cell
  double* uc = scratch + (size_t)(wave * CPW + half) * (Nc + n1 + n2);
  double* t1 = uc + Nc;
  double* t2 = t1 + n1;
  const uint16_t* cl = A.clmaps + (size_t)A.clmap_id[p] * A.K * Nc;
  const uint16_t* fl = A.lmaps + (size_t)A.lmap_id[p] * A.K * Nf;
  // the (cell, local dof) -> patch position tables of the NEXT cell are fetched while the
  // current one is computed, so the cell loop itself touches only LDS
  constexpr int FP = (Nf + HL - 1) / HL, CP = (Nc + HL - 1) / HL;
  int fcur[FP], ccur[CP], fnxt[FP], cnxt[CP];
  auto fetch = [&](int slot, int* fi, int* ci) {
    const int sl = slot < nc ? slot : nc - 1; // (a half-wavefront without a cell re-reads the last one's tables)
#pragma unroll
    for (int j = 0; j < CP; ++j)
    {
      const int o = ll + HL * j;
      ci[j] = cl[(size_t)sl * Nc + (o < Nc ? o : Nc - 1)];
    }
#pragma unroll
    for (int j = 0; j < FP; ++j)
    {
      const int o = ll + HL * j, oc = o < Nf ? o : Nf - 1;
      const int a = oc / (ndf * ndf), r = oc - a * ndf * ndf, b = r / ndf, c = r - b * ndf;
      fi[j] = fl[(size_t)sl * Nf + ftab(ndf, a, b, c)];
    }
  };
  if (wave * CPW < nc)
    fetch(wave * CPW + half, fcur, ccur);
  for (int slot0 = wave * CPW; slot0 < nc; slot0 += nw * CPW)
  {
    const int slot = slot0 + half;
    const bool mine = slot < nc;
    if (slot0 + nw * CPW < nc)
      fetch(slot + nw * CPW, fnxt, cnxt);
#pragma unroll
    for (int j = 0; j < CP; ++j)
    {
      const int o = ll + HL * j;
      if (o < Nc)
        uc[o] = sc[ccur[j]];
    }
    tfence();
    for (int o = ll; o < n1; o += HL) // (a, j, k): sum over i
    {
      const int a = o / (ndc * ndc), jk = o - a * ndc * ndc;
      double w = 0.0;
      #pragma unroll
      for (int i = 0; i < ndc; ++i)
        w += sM[a * ndc + i] * uc[i * ndc * ndc + jk];
      t1[o] = w;
    }
    tfence();
    for (int o = ll; o < n2; o += HL) // (a, b, k): sum over j
    {
      const int a = o / (ndf * ndc), r = o - a * ndf * ndc, b = r / ndc, k = r - b * ndc;
      double w = 0.0;
      #pragma unroll
      for (int j = 0; j < ndc; ++j)
        w += sM[b * ndc + j] * t1[(a * ndc + j) * ndc + k];
      t2[o] = w;
    }
    tfence();
#pragma unroll
    for (int jj = 0; jj < FP; ++jj) // (a, b, c): sum over k
    {
      const int o = ll + HL * jj;
      if (o < Nf)
      {
        const int a = o / (ndf * ndf), r = o - a * ndf * ndf, b = r / ndf, c = r - b * ndf;
        double w = 0.0;
        #pragma unroll
        for (int k = 0; k < ndc; ++k)
          w += sM[c * ndc + k] * t2[(a * ndf + b) * ndc + k];
        if (mine)
          sf[fcur[jj]] = w; // shared dofs: identical values
      }
    }
    tfence();
#pragma unroll
    for (int j = 0; j < CP; ++j)
      ccur[j] = cnxt[j];
#pragma unroll
    for (int j = 0; j < FP; ++j)
      fcur[j] = fnxt[j];
  }
  tbarrier();
#pragma unroll
  for (int k = 0; k < IT; ++k)
  {
    const int i = t + k * nthr;
    if (i < Mf && !(m[k] & PD_ACC)) // this patch is the first (only) writer of the dof
      fine[m[k] & PD_MASK] = v[k] + sf[i]; // src/interpolate.hpp:42 (+ src/pmg.hpp:129 when add)
  }
  for (int i = t + IT * nthr; i < Mf; i += nthr) // (lists longer than the registers hold: few threads per workgroup)
  {
    const uint32_t mm = A.pdofs[off + i];
    if (!(mm & PD_ACC))
      fine[mm & PD_MASK] = (add ? fine[mm & PD_MASK] : 0.0) + sf[i];
  }
}

template <int NDC, int NDF>
__global__ void restrict_patch_kernel(TransferArgs A, const double* __restrict__ fine,
                                      const double* __restrict__ fine_sub, double* __restrict__ coarse,
                                      int atomic_out)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int ndc = NDC, ndf = NDF, Nc = ndc * ndc * ndc, Nf = ndf * ndf * ndf;
  constexpr int n1 = ndf * ndc * ndc, n2 = ndf * ndf * ndc;
  constexpr int CPW = transfer_cpw(NDF), HL = 64 / CPW;
  double* sM = smem;
  double* sc = sM + ndf * ndc;     // [max_mc] coarse accumulators
  double* sf = sc + A.max_mc;      // [max_mf] weighted fine values
  double* scratch = sf + A.max_mf; // per wave and cell of the pass: w[Nf] t2[n2] t1[n1]
  const int p = A.first + blockIdx.x, t = threadIdx.x, nthr = blockDim.x;
  const int off = A.poff[p], Mf = A.poff[p + 1] - off;
  const int coff = A.cpoff[p], Mc = A.cpoff[p + 1] - coff;
  const int nc = A.pncell[p];
  // the coarse list, for the write-back at the end (one entry per thread: coarse patches are small; longer lists
  // fall back to re-reading)
  const uint32_t cm = A.cpdofs[coff + (t < Mc ? t : Mc - 1)];
  for (int i = t; i < ndf * ndc; i += nthr)
    sM[i] = A.M1[i];
  // four independent (index -> value) load chains per thread and pass; clamped indices
  // keep every load unconditional
  for (int i0 = t; i0 < Mf; i0 += 4 * nthr)
  {
    uint32_t m[4];
    uint8_t mu[4];
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const int i = i0 + k * nthr;
      const int ic = off + (i < Mf ? i : Mf - 1);
      m[k] = A.pdofs[ic];
      mu[k] = A.pmult[ic];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      v[k] = A.nt ? __builtin_nontemporal_load(fine + (m[k] & PD_MASK)) : fine[m[k] & PD_MASK];
    if (fine_sub) // the residual r - q formed here instead of in a pass of its own (src/chebyshev.hpp:77)
    {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        v[k] -= A.nt ? __builtin_nontemporal_load(fine_sub + (m[k] & PD_MASK)) : fine_sub[m[k] & PD_MASK];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const int i = i0 + k * nthr;
      if (i < Mf)
        sf[i] = v[k] / (double)mu[k]; // src/interpolate.hpp:81-82
    }
  }
  {
    // onto the earlier colours (coloured write-back only)
    const double c0 = (!atomic_out && (cm & PD_ACC)) ? coarse[cm & PD_MASK] : 0.0;
    if (t < Mc)
      sc[t] = c0;
    for (int i = t + nthr; i < Mc; i += nthr)
    {
      const uint32_t mm = A.cpdofs[coff + i];
      sc[i] = (!atomic_out && (mm & PD_ACC)) ? coarse[mm & PD_MASK] : 0.0;
    }
  }
  tbarrier();
  const int wave = t >> 6, lane = t & 63, nw = nthr >> 6;
  const int half = lane / HL, ll = lane - half * HL;
  double* w = scratch + (size_t)(wave * CPW + half) * (Nf + n1 + n2);
  double* t2 = w + Nf;
  double* t1 = t2 + n2;
  const uint16_t* cl = A.clmaps + (size_t)A.clmap_id[p] * A.K * Nc;
  const uint16_t* fl = A.lmaps + (size_t)A.lmap_id[p] * A.K * Nf;
  constexpr int FP = (Nf + HL - 1) / HL, CP = (Nc + HL - 1) / HL;
  int fcur[FP], ccur[CP], fnxt[FP], cnxt[CP]; // position tables, fetched one cell ahead (see prolong_patch_kernel)
  auto fetch = [&](int slot, int* fi, int* ci) {
    const int sl = slot < nc ? slot : nc - 1;
#pragma unroll
    for (int j = 0; j < CP; ++j)
    {
      const int o = ll + HL * j;
      ci[j] = cl[(size_t)sl * Nc + (o < Nc ? o : Nc - 1)];
    }
#pragma unroll
    for (int j = 0; j < FP; ++j)
    {
      const int o = ll + HL * j, oc = o < Nf ? o : Nf - 1;
      const int a = oc / (ndf * ndf), r = oc - a * ndf * ndf, b = r / ndf, c = r - b * ndf;
      fi[j] = fl[(size_t)sl * Nf + ftab(ndf, a, b, c)];
    }
  };
  if (wave * CPW < nc)
    fetch(wave * CPW + half, fcur, ccur);
  for (int slot0 = wave * CPW; slot0 < nc; slot0 += nw * CPW)
  {
    const int slot = slot0 + half;
    const bool mine = slot < nc;
    if (slot0 + nw * CPW < nc)
      fetch(slot + nw * CPW, fnxt, cnxt);
#pragma unroll
    for (int j = 0; j < FP; ++j)
    {
      const int o = ll + HL * j;
      if (o < Nf)
        w[o] = sf[fcur[j]];
    }
    tfence();
    for (int o = ll; o < n2; o += HL) // (a, b, k): sum over c
    {
      const int ab = o / ndc, k = o - ab * ndc;
      double v = 0.0;
      #pragma unroll
      for (int c = 0; c < ndf; ++c)
        v += sM[c * ndc + k] * w[ab * ndf + c];
      t2[o] = v;
    }
    tfence();
    for (int o = ll; o < n1; o += HL) // (a, j, k): sum over b
    {
      const int a = o / (ndc * ndc), r = o - a * ndc * ndc, j = r / ndc, k = r - j * ndc;
      double v = 0.0;
      #pragma unroll
      for (int b = 0; b < ndf; ++b)
        v += sM[b * ndc + j] * t2[(a * ndf + b) * ndc + k];
      t1[o] = v;
    }
    tfence();
#pragma unroll
    for (int jj = 0; jj < CP; ++jj) // (i, j, k): sum over a
    {
      const int o = ll + HL * jj;
      if (o < Nc)
      {
        const int i = o / (ndc * ndc), jk = o - i * ndc * ndc;
        double v = 0.0;
        #pragma unroll
        for (int a = 0; a < ndf; ++a)
          v += sM[a * ndc + i] * t1[a * ndc * ndc + jk];
        if (mine)
          atomicAdd(&sc[ccur[jj]], v); // in LDS
      }
    }
    tfence();
#pragma unroll
    for (int j = 0; j < CP; ++j)
      ccur[j] = cnxt[j];
#pragma unroll
    for (int j = 0; j < FP; ++j)
      fcur[j] = fnxt[j];
  }
  tbarrier();
  if (atomic_out) // single launch over all patches, coarse zero-filled beforehand
  {
    if (t < Mc)
      atomicAdd(&coarse[cm & PD_MASK], sc[t]);
    for (int i = t + nthr; i < Mc; i += nthr)
      atomicAdd(&coarse[A.cpdofs[coff + i] & PD_MASK], sc[i]);
  }
  else
  {
    if (t < Mc)
      coarse[cm & PD_MASK] = sc[t];
    for (int i = t + nthr; i < Mc; i += nthr)
      coarse[A.cpdofs[coff + i] & PD_MASK] = sc[i];
  }
}

// (coarse nd, fine nd) -> kernel instantiation
#define PMG_FOR_PAIRS(X)                                                                            \
  X(2, 3) X(2, 4) X(2, 5) X(2, 6) X(2, 7) X(2, 8) X(2, 9) X(3, 4) X(3, 5) X(3, 6) X(3, 7) X(3, 8)   \
  X(3, 9) X(4, 5) X(4, 6) X(4, 7) X(4, 8) X(4, 9) X(5, 6) X(5, 7) X(5, 8) X(5, 9) X(6, 7) X(6, 8)   \
  X(6, 9) X(7, 8) X(7, 9) X(8, 9)

int launch_prolong_patch(int ndc, int ndf, int grid, int threads, size_t shm, hipStream_t s,
                         const TransferArgs& A, const double* coarse, double* fine, int add)
{
#define X(C, F)                                                                                    \
  if (ndc == C && ndf == F)                                                                        \
  {                                                                                                \
    prolong_patch_kernel<C, F><<<grid, threads, shm, s>>>(A, coarse, fine, add);                   \
    return PMG_OK;                                                                                 \
  }
  PMG_FOR_PAIRS(X)
#undef X
  return fail(PMG_ERR_INVALID, "unsupported degree pair");
}

int launch_restrict_patch(int ndc, int ndf, int grid, int threads, size_t shm, hipStream_t s,
                          const TransferArgs& A, const double* fine, const double* fine_sub, double* coarse,
                          int atomic_out)
{
#define X(C, F)                                                                                    \
  if (ndc == C && ndf == F)                                                                        \
  {                                                                                                \
    restrict_patch_kernel<C, F><<<grid, threads, shm, s>>>(A, fine, fine_sub, coarse, atomic_out); \
    return PMG_OK;                                                                                 \
  }
  PMG_FOR_PAIRS(X)
#undef X
  return fail(PMG_ERR_INVALID, "unsupported degree pair");
}

int set_patch_kernel_lds(int ndc, int ndf, int bytes)
{
#define X(C, F)                                                                                    \
  if (ndc == C && ndf == F)                                                                        \
  {                                                                                                \
    PMG_HIP(hipFuncSetAttribute((const void*)prolong_patch_kernel<C, F>,                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));               \
    PMG_HIP(hipFuncSetAttribute((const void*)restrict_patch_kernel<C, F>,                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));               \
    return PMG_OK;                                                                                 \
  }
  PMG_FOR_PAIRS(X)
#undef X
  return fail(PMG_ERR_INVALID, "unsupported degree pair");
}

// multiplicity of every fine patch dof, as a byte next to pdofs
__global__ void patch_mult_kernel(long long n, const uint32_t* __restrict__ pdofs,
                                  const double* __restrict__ inv_mult, uint8_t* __restrict__ pm)
{
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
  {
    const double im = inv_mult[pdofs[i] & PD_MASK];
    pm[i] = (uint8_t)(im > 0.0 ? 1.0 / im + 0.5 : 1.0);
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

namespace
{
TransferArgs make_args(pmg_interpolator ip, int first)
{
  TransferArgs A;
  A.first = first;
  A.ndc = ip->ndc;
  A.ndf = ip->ndf;
  A.K = ip->fv.K;
  A.max_mf = ip->fv.max_m;
  A.max_mc = ip->cmax_m;
  A.nt = ip->lf->total() >= (4 << 20) ? 1 : 0; // same rule as the smoother kernels (vector.hip)
  A.poff = ip->fv.poff;
  A.lmap_id = ip->fv.lmap_id;
  A.pncell = ip->fv.pncell;
  A.cpoff = ip->cpoff;
  A.clmap_id = ip->clmap_id;
  A.pdofs = ip->fv.pdofs;
  A.cpdofs = ip->cpdofs;
  A.lmaps = ip->fv.lmaps;
  A.clmaps = ip->clmaps;
  A.pmult = ip->pmult;
  A.M1 = ip->M1;
  return A;
}

// the fine operator is a small level's (all launches merged) and the layout's exchange is one launch
bool whole_exchange(pmg_interpolator ip, pmg_layout l)
{
  return ip->fv.merged && l->num_ghosts > 0 && layout_exchanges_whole(l);
}

// patches of the interior cell list come first in launch order
int interior_patches(pmg_interpolator ip)
{
  const auto& lf = *ip->fv.launch_first;
  return ip->fv.n_launch_l < (int)lf.size() ? lf[ip->fv.n_launch_l] : ip->fv.npatch;
}

int prolong_patched(pmg_interpolator ip, double* coarse, double* fine, int add, hipStream_t s)
{
  const int n_int = interior_patches(ip), n_all = ip->fv.npatch;
  if (whole_exchange(ip, ip->lc)) // a small level: the exchange whole, then all patches in one launch (laplacian.hip)
  {
    PMG_TRY(scatter_fwd_whole(ip->lc, coarse, s));
    if (n_all > 0)
      PMG_TRY(launch_prolong_patch(ip->ndc, ip->ndf, n_all, ip->pwaves * 64, ip->pshm, s, make_args(ip, 0), coarse,
                                   fine, add));
    PMG_HIP(hipGetLastError());
    return PMG_OK;
  }
  PMG_TRY(pmg_scatter_fwd_begin(ip->lc, coarse, (pmg_stream)s)); // src/interpolate.hpp:202
  if (n_int > 0)
    PMG_TRY(launch_prolong_patch(ip->ndc, ip->ndf, n_int, ip->pwaves * 64, ip->pshm, s, make_args(ip, 0),
                                 coarse, fine, add));
  PMG_TRY(pmg_scatter_fwd_end(ip->lc, coarse, (pmg_stream)s)); // :217
  if (n_all > n_int)
    PMG_TRY(launch_prolong_patch(ip->ndc, ip->ndf, n_all - n_int, ip->pwaves * 64, ip->pshm, s,
                                 make_args(ip, n_int), coarse, fine, add));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

int restrict_patched(pmg_interpolator ip, double* fine, const double* fine_sub, double* coarse, hipStream_t s)
{
  // One launch per cell list; the patch sums go to the (small) coarse vector with
  // FP64 atomics: a patch issues a few dozen 64-byte atomic requests (its coarse
  // dofs form long runs), three orders of magnitude fewer than one per
  // (cell, coarse dof) as in src/interpolate.hpp:84, and 8 colour launches of a
  // ~30 us kernel would cost more than they save.
  // fine_sub (optional, layouts without ghosts only): restrict fine - fine_sub.
  const int n_int = interior_patches(ip), n_all = ip->fv.npatch;
  PMG_REQUIRE(!fine_sub || ip->lf->num_ghosts == 0, "restriction of a difference needs a layout without ghosts");
  if (whole_exchange(ip, ip->lf))
  {
    PMG_TRY(scatter_fwd_whole(ip->lf, fine, s));
    launch_zero(ip->lc->total(), coarse, s);
    if (n_all > 0)
      PMG_TRY(launch_restrict_patch(ip->ndc, ip->ndf, n_all, ip->pwaves * 64, ip->pshm, s, make_args(ip, 0), fine,
                                    fine_sub, coarse, 1));
    PMG_HIP(hipGetLastError());
    return PMG_OK;
  }
  PMG_TRY(pmg_scatter_fwd_begin(ip->lf, fine, (pmg_stream)s));             // :264
  launch_zero(ip->lc->total(), coarse, s); // :270
  if (n_int > 0)
    PMG_TRY(launch_restrict_patch(ip->ndc, ip->ndf, n_int, ip->pwaves * 64, ip->pshm, s,
                                  make_args(ip, 0), fine, fine_sub, coarse, 1));
  PMG_TRY(pmg_scatter_fwd_end(ip->lf, fine, (pmg_stream)s)); // :281
  if (n_all > n_int)
    PMG_TRY(launch_restrict_patch(ip->ndc, ip->ndf, n_all - n_int, ip->pwaves * 64, ip->pshm, s,
                                  make_args(ip, n_int), fine, fine_sub, coarse, 1));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}
} // namespace

namespace pmg
{
bool interp_is_patched(pmg_interpolator ip) { return ip->patched; }

// fine += P coarse in one pass (src/pmg.hpp:123-129 fused); patch path only
int interp_prolong_add(pmg_interpolator ip, double* coarse, double* fine, hipStream_t s)
{
  PMG_REQUIRE(ip->patched, "interp_prolong_add needs the patch path");
  return prolong_patched(ip, coarse, fine, 1, s);
}

// can the restriction form the residual r - q itself?  (patch path, no halo to refresh)
bool interp_restricts_difference(pmg_interpolator ip) { return ip->patched && ip->lf->num_ghosts == 0; }

// coarse = R (fine - fine_sub)
int interp_restrict_difference(pmg_interpolator ip, double* fine, const double* fine_sub, double* coarse,
                               hipStream_t s)
{
  PMG_REQUIRE(interp_restricts_difference(ip), "interp_restrict_difference: not available for this interpolator");
  return restrict_patched(ip, fine, fine_sub, coarse, s);
}

int interp_prolong(pmg_interpolator ip, double* coarse, double* fine, hipStream_t s)
{
  if (ip->patched)
    return prolong_patched(ip, coarse, fine, 0, s);
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
  if (ip->patched)
    return restrict_patched(ip, fine, nullptr, coarse, s);
  const size_t shm = sizeof(double) * (ip->ndf * ip->ndc + (size_t)ip->cpb * ip->Nf);
  PMG_TRY(pmg_scatter_fwd_begin(ip->lf, fine, (pmg_stream)s));             // :264
  launch_zero(ip->lc->total(), coarse, s); // :270
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
  HandleGuard<pmg_interpolator> guard(ip, pmg_interpolator_destroy);
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
  *out = guard.release();
  return PMG_OK;
}

extern "C" int pmg_interpolator_create_with_operator(
    pmg_interpolator* out, pmg_layout layout_coarse, pmg_layout layout_fine, int degree_coarse,
    int degree_fine, int32_t ncells, const int32_t* dofmap_coarse, const int32_t* dofmap_fine,
    const int32_t* lcells, int32_t n_lcells, const int32_t* bcells, int32_t n_bcells,
    pmg_laplacian fine_operator, pmg_stream stream)
{
  PMG_TRY(pmg_interpolator_create(out, layout_coarse, layout_fine, degree_coarse, degree_fine, ncells,
                                  dofmap_coarse, dofmap_fine, lcells, n_lcells, bcells, n_bcells,
                                  stream));
  if (!fine_operator)
    return PMG_OK;
  pmg_interpolator ip = *out;
  *out = nullptr; // handed back only when the patch data below is complete
  HandleGuard<pmg_interpolator> guard(ip, pmg_interpolator_destroy);
  hipStream_t s = S(stream);
  PMG_REQUIRE(laplacian_layout(fine_operator) == layout_fine,
              "pmg_interpolator_create_with_operator: operator is not on the fine layout");
  ip->fv = laplacian_patches(fine_operator);
  PMG_REQUIRE(ip->fv.P == degree_fine,
              "pmg_interpolator_create_with_operator: operator degree %d != fine degree %d",
              ip->fv.P, degree_fine);
  const PatchView& v = ip->fv;
  const int K = v.K, Nc = ip->Nc, np = v.npatch;
  const int nc_total = layout_coarse->total();

  std::vector<int32_t> h_dmc((size_t)ncells * Nc);
  PMG_HIP(hipMemcpyAsync(h_dmc.data(), dofmap_coarse, sizeof(int32_t) * h_dmc.size(),
                         hipMemcpyDeviceToHost, s));
  PMG_HIP(hipStreamSynchronize(s));

  // launch index of every patch
  std::vector<int32_t> launch_of(np, 0);
  for (size_t l = 0; l < v.launch_first->size(); ++l)
    for (int q = 0; q < (*v.launch_count)[l]; ++q)
      launch_of[(*v.launch_first)[l] + q] = (int32_t)l;

  std::vector<int32_t> cpoff(np + 1, 0), clmap_id(np, 0), first(nc_total, INT32_MAX);
  std::vector<uint32_t> cpdofs;
  std::vector<uint16_t> clmaps;
  std::map<std::vector<uint16_t>, int32_t> uniq;
  std::vector<std::vector<int32_t>> lists(np);
  int cmax = 1;
  for (int p = 0; p < np; ++p)
  {
    std::vector<int32_t>& d = lists[p];
    const int nc = (*v.pncell_h)[p];
    for (int sl = 0; sl < nc; ++sl)
    {
      const int32_t cell = (*v.pcell_h)[(size_t)p * K + sl];
      PMG_REQUIRE(cell >= 0 && cell < ncells, "pmg_interpolator_create_with_operator: cell out of range");
      for (int k = 0; k < Nc; ++k)
      {
        const int32_t dof = h_dmc[(size_t)cell * Nc + k];
        PMG_REQUIRE(dof >= 0 && dof < nc_total, "coarse dofmap entry %d out of range", dof);
        d.push_back(dof);
      }
    }
    std::sort(d.begin(), d.end());
    d.erase(std::unique(d.begin(), d.end()), d.end());
    PMG_REQUIRE(d.size() <= 65535, "coarse patch too large");
    cmax = std::max(cmax, (int)d.size());
    for (int32_t dof : d)
      first[dof] = std::min(first[dof], launch_of[p]);
  }
  for (int p = 0; p < np; ++p)
  {
    const std::vector<int32_t>& d = lists[p];
    for (int32_t dof : d)
      cpdofs.push_back((uint32_t)dof | (first[dof] != launch_of[p] ? PD_ACC : 0u));
    cpoff[p + 1] = (int32_t)cpdofs.size();
    std::vector<uint16_t> lm((size_t)K * Nc, 0);
    const int nc = (*v.pncell_h)[p];
    for (int sl = 0; sl < nc; ++sl)
    {
      const int32_t cell = (*v.pcell_h)[(size_t)p * K + sl];
      for (int k = 0; k < Nc; ++k)
        lm[(size_t)sl * Nc + k] = (uint16_t)(
            std::lower_bound(d.begin(), d.end(), h_dmc[(size_t)cell * Nc + k]) - d.begin());
    }
    auto it = uniq.find(lm);
    if (it == uniq.end())
    {
      it = uniq.emplace(lm, (int32_t)uniq.size()).first;
      clmaps.insert(clmaps.end(), lm.begin(), lm.end());
    }
    clmap_id[p] = it->second;
  }
  ip->cmax_m = cmax;
  PMG_TRY(upload(&ip->cpoff, cpoff.data(), cpoff.size(), s));
  PMG_TRY(upload(&ip->cpdofs, cpdofs.data(), cpdofs.size(), s));
  PMG_TRY(upload(&ip->clmap_id, clmap_id.data(), clmap_id.size(), s));
  PMG_TRY(upload(&ip->clmaps, clmaps.data(), clmaps.size(), s));
  PMG_HIP(hipMalloc(&ip->pmult, v.npdofs ? v.npdofs : 1));
  if (v.npdofs > 0)
  {
    long long blocks = (v.npdofs + 255) / 256;
    patch_mult_kernel<<<(int)(blocks > 4096 ? 4096 : blocks), 256, 0, s>>>(v.npdofs, v.pdofs,
                                                                           ip->inv_mult, ip->pmult);
    PMG_HIP(hipGetLastError());
  }
  // LDS: table + coarse list + fine list + per-wave scratch
  const int ndc = ip->ndc, ndf = ip->ndf;
  const size_t per_wave = (size_t)transfer_cpw(ndf) * ((size_t)ip->Nf + ndf * ndc * ndc + ndf * ndf * ndc);
  const size_t base = (size_t)ndf * ndc + cmax + v.max_m;
  int waves = 8;
  if (const char* e = std::getenv("PMG_TRANSFER_WAVES")) // tuning: waves per patch (1 .. 16)
    waves = std::max(1, std::min(16, std::atoi(e)));
  while (waves > 1 && 8 * (base + waves * per_wave) > 64 * 1024)
    --waves;
  ip->pwaves = waves;
  ip->pshm = 8 * (base + waves * per_wave);
  PMG_REQUIRE(ip->pshm <= 160 * 1024, "transfer kernels need %zu bytes of LDS", ip->pshm);
  if (ip->pshm > 48 * 1024)
    PMG_TRY(set_patch_kernel_lds(ndc, ndf, (int)ip->pshm));
  PMG_HIP(hipStreamSynchronize(s));
  ip->patched = true;
  *out = guard.release();
  return PMG_OK;
}

// The same for dofmaps in the caller's cell-local node order (pmg_amd.h "cell-local node order"): ascending copies
// are made once and owned by the handle; the 1-D interpolation table and every kernel stay ascending.
extern "C" int pmg_interpolator_create_ordered(pmg_interpolator* out, pmg_layout layout_coarse, pmg_layout layout_fine,
                                               int degree_coarse, int degree_fine, int32_t ncells,
                                               const int32_t* dofmap_coarse, const int32_t* dofmap_fine,
                                               const int32_t* lcells, int32_t n_lcells, const int32_t* bcells,
                                               int32_t n_bcells, pmg_laplacian fine_operator, int node_order,
                                               const int32_t* custom_coarse, const int32_t* custom_fine,
                                               pmg_stream stream)
{
  PMG_REQUIRE(out, "pmg_interpolator_create_ordered: NULL handle");
  PMG_REQUIRE(degree_coarse >= 1 && degree_fine > degree_coarse && degree_fine <= PMG_MAX_DEGREE,
              "pmg_interpolator_create: need 1 <= degree_coarse < degree_fine <= %d", PMG_MAX_DEGREE);
  PMG_REQUIRE(ncells >= 0 && (ncells == 0 || (dofmap_coarse && dofmap_fine)), "pmg_interpolator_create: NULL dofmap");
  std::vector<int32_t> pc, pf;
  PMG_TRY(node_permutation(node_order, degree_coarse, custom_coarse, pc));
  PMG_TRY(node_permutation(node_order, degree_fine, custom_fine, pf));
  hipStream_t s = S(stream);
  int32_t* own[2] = {nullptr, nullptr};
  const int32_t* use[2] = {dofmap_coarse, dofmap_fine};
  auto release = [&] {
    (void)hipFree(own[0]);
    (void)hipFree(own[1]);
  };
  for (int which = 0; which < 2; ++which)
  {
    const std::vector<int32_t>& p1 = which ? pf : pc;
    if (is_identity(p1) || ncells == 0)
      continue;
    const int nd = (int)p1.size(), N = nd * nd * nd;
    const std::vector<int32_t> p3 = cell_permutation(nd, p1);
    int32_t* p3_d = nullptr;
    int rc = upload(&p3_d, p3.data(), p3.size(), s);
    if (rc == PMG_OK && hipMalloc(&own[which], sizeof(int32_t) * (size_t)ncells * N) != hipSuccess)
      rc = fail(PMG_ERR_HIP, "pmg_interpolator_create_ordered: out of device memory");
    if (rc == PMG_OK)
      rc = permute_rows_i32(ncells, N, p3_d, use[which], own[which], s);
    if (rc == PMG_OK && hipStreamSynchronize(s) != hipSuccess)
      rc = fail(PMG_ERR_HIP, "pmg_interpolator_create_ordered: permutation failed");
    (void)hipFree(p3_d);
    if (rc != PMG_OK)
    {
      release();
      return rc;
    }
    use[which] = own[which];
  }
  const int rc = pmg_interpolator_create_with_operator(out, layout_coarse, layout_fine, degree_coarse, degree_fine,
                                                       ncells, use[0], use[1], lcells, n_lcells, bcells, n_bcells,
                                                       fine_operator, stream);
  if (rc != PMG_OK)
  {
    release();
    return rc;
  }
  (*out)->dmc_own = own[0];
  (*out)->dmf_own = own[1];
  return PMG_OK;
}

extern "C" int pmg_interpolator_interpolate_add(pmg_interpolator ip, double* coarse, double* fine,
                                                pmg_stream stream)
{
  PMG_REQUIRE(ip && coarse && fine, "pmg_interpolator_interpolate_add: NULL argument");
  if (ip->patched)
    return interp_prolong_add(ip, coarse, fine, S(stream));
  return fail(PMG_ERR_INVALID, "pmg_interpolator_interpolate_add needs an interpolator created "
                               "with pmg_interpolator_create_with_operator");
}

extern "C" int pmg_interpolator_destroy(pmg_interpolator ip)
{
  if (!ip)
    return PMG_OK;
  (void)hipFree(ip->dmc_own);
  (void)hipFree(ip->dmf_own);
  (void)hipFree(ip->cpoff);
  (void)hipFree(ip->cpdofs);
  (void)hipFree(ip->clmap_id);
  (void)hipFree(ip->clmaps);
  (void)hipFree(ip->pmult);
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
