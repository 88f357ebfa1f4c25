// Distributed-vector layer: layout handle, halo pack/unpack, BLAS-1 and the
// fused smoother / CG passes.  Replaces acc::Vector and the free functions of
// src/vector.hpp (pack/unpack :24-55, scatter :186-286, BLAS-1 :333-454).
//
// Every kernel is a pure HBM stream: 16 B per lane (double2) when the operands
// are 16-byte aligned, grid capped at 8 blocks per CU and grid-strided.
// Nothing synchronises the host except the reductions that must return a value.
#include "common.hpp"

using namespace pmg;

namespace
{
constexpr int EW_THREADS = 256;
constexpr int EW_MAX_BLOCKS = 256 * 8;

inline int ew_blocks(long long work)
{
  long long b = (work + EW_THREADS - 1) / EW_THREADS;
  if (b < 1)
    b = 1;
  return (int)(b > EW_MAX_BLOCKS ? EW_MAX_BLOCKS : b);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- halo ----
__global__ void pack_kernel(int n, const int32_t* __restrict__ idx, const double* __restrict__ in,
                            double* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[i] = in[idx[i]];
}

__global__ void unpack_kernel(int n, const int32_t* __restrict__ idx, const double* __restrict__ in,
                              double* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[idx[i]] = in[i];
}

__global__ void unpack_add_kernel(int n, const int32_t* __restrict__ idx,
                                  const double* __restrict__ in, double* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    atomicAdd(&out[idx[i]], in[i]);
}

// the same through a position list of the staging buffer (padded per-neighbour segments, comm.hip)
__global__ void pack_pos_kernel(int n, const int32_t* __restrict__ idx, const int32_t* __restrict__ pos,
                                const double* __restrict__ in, double* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[pos[i]] = in[idx[i]];
}
__global__ void unpack_pos_kernel(int n, const int32_t* __restrict__ idx, const int32_t* __restrict__ pos,
                                  const double* __restrict__ in, double* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[idx[i]] = in[pos[i]];
}
__global__ void unpack_add_pos_kernel(int n, const int32_t* __restrict__ idx, const int32_t* __restrict__ pos,
                                      const double* __restrict__ in, double* __restrict__ out)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    atomicAdd(&out[idx[i]], in[pos[i]]);
}

// ---- generic element-wise launcher: F::apply(i, args...) on doubles ----
// pairs [0, n2) plus, when `tail` >= 0, the single trailing element (odd lengths)
template <typename F>
__global__ void ew_kernel2(int n2, int tail, F f)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x)
    f.pair(i);
  if (tail >= 0 && blockIdx.x == 0 && threadIdx.x == 0)
    f.one(tail);
}
template <typename F>
__global__ void ew_kernel1(int begin, int n, F f)
{
  for (int i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    f.one(i);
}

template <typename F>
void ew_launch(int n, bool vec_ok, F f, hipStream_t s)
{
  if (n <= 0)
    return;
  if (vec_ok)
  {
    const int n2 = n / 2;
    ew_kernel2<<<ew_blocks(n2), EW_THREADS, 0, s>>>(n2, (n & 1) ? n - 1 : -1, f);
  }
  else
    ew_kernel1<<<ew_blocks(n), EW_THREADS, 0, s>>>(0, n, f);
}


// F over the owned entries [0, n_owned), and q[0, n_total) = 0 behind it: the vector kernel that
// consumes the operator's last output also clears it for the next application of an operator whose
// launch accumulates with atomics (a small level's merged launch) -- no zero-fill kernel of its own.
template <typename F>
struct ThenClearF
{
  F f;
  double* q;
  int n_owned;
  __device__ void pair(int i) const
  {
    if (2 * i + 1 < n_owned)
    {
      f.pair(i);
      reinterpret_cast<double2*>(q)[i] = make_double2(0.0, 0.0);
    }
    else
    {
      one(2 * i);
      one(2 * i + 1);
    }
  }
  __device__ void one(int i) const
  {
    if (i < n_owned)
      f.one(i);
    q[i] = 0.0;
  }
};
template <typename F>
void ew_launch_clear(int n_owned, int n_total, bool vec_ok, F f, double* q, hipStream_t s)
{
  ew_launch(n_total, vec_ok && aligned16(q), ThenClearF<F>{f, q, n_owned}, s);
}

#define D2(p) reinterpret_cast<double2*>(p)
#define CD2(p) reinterpret_cast<const double2*>(p)

struct SetF
{
  double* x;
  double v;
  __device__ void pair(int i) const { D2(x)[i] = make_double2(v, v); }
  __device__ void one(int i) const { x[i] = v; }
};
struct ScaleF
{
  double* x;
  double a;
  __device__ void pair(int i) const
  {
    double2 t = D2(x)[i];
    t.x *= a;
    t.y *= a;
    D2(x)[i] = t;
  }
  __device__ void one(int i) const { x[i] *= a; }
};
struct CopyF
{
  double* d;
  const double* s;
  __device__ void pair(int i) const { D2(d)[i] = CD2(s)[i]; }
  __device__ void one(int i) const { d[i] = s[i]; }
};
struct AxpyF
{ // r = a x + y   (src/vector.hpp:398-407)
  double* r;
  double a;
  const double* x;
  const double* y;
  __device__ void pair(int i) const
  {
    double2 vx = CD2(x)[i], vy = CD2(y)[i];
    D2(r)[i] = make_double2(vx.x * a + vy.x, vx.y * a + vy.y);
  }
  __device__ void one(int i) const { r[i] = x[i] * a + y[i]; }
};
struct MulF
{ // w = x .* y  (src/vector.hpp:438-447)
  double* w;
  const double* x;
  const double* y;
  __device__ void pair(int i) const
  {
    double2 vx = CD2(x)[i], vy = CD2(y)[i];
    D2(w)[i] = make_double2(vx.x * vy.x, vx.y * vy.y);
  }
  __device__ void one(int i) const { w[i] = x[i] * y[i]; }
};
// Cache policy of the smoother's vector kernels.  On a level whose vectors do not fit the
// MALL anyway (NT = true, chosen by length in the launchers below) every operand that is
// not consumed by the very next kernel is streamed with the nt hint; z, the input of the
// next operator application, keeps the default policy.  Measured on the 64^3 p = 4 -> 2 -> 1
// V-cycle: 6.01 -> 5.64 ms.  Small levels keep the default policy (their vectors stay
// resident between kernels).
typedef double ntv2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 ld2(const double* p, int i)
{
  if constexpr (NT)
  {
    ntv2 v = __builtin_nontemporal_load(reinterpret_cast<const ntv2*>(p) + i);
    return make_double2(v.x, v.y);
  }
  else
    return CD2(p)[i];
}
template <bool NT>
__device__ __forceinline__ void st2(double* p, int i, double2 v)
{
  if constexpr (NT)
  {
    ntv2 w;
    w.x = v.x;
    w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<ntv2*>(p) + i);
  }
  else
    D2(p)[i] = v;
}

template <bool NT>
struct ChebInitF
{ // r = b - q ; z = c0 * dinv * r   (src/chebyshev.hpp:57,67-68); q == nullptr: r = b
  double* r;
  double* z;
  const double* b;
  const double* q;
  const double* dinv;
  double c0;
  __device__ void pair(int i) const
  {
    double2 vb = ld2<NT>(b, i), vd = ld2<NT>(dinv, i);
    double2 vr = vb;
    if (q)
    {
      double2 vq = ld2<NT>(q, i);
      vr.x -= vq.x;
      vr.y -= vq.y;
    }
    st2<NT>(r, i, vr);
    D2(z)[i] = make_double2(c0 * vd.x * vr.x, c0 * vd.y * vr.y);
  }
  __device__ void one(int i) const
  {
    double vr = b[i] - (q ? q[i] : 0.0);
    r[i] = vr;
    z[i] = c0 * dinv[i] * vr;
  }
};
// The iterate absorbs every correction z_{i+1} in the kernel that computes it (x is streamed by that
// kernel anyway), so no pass is needed for "x += z" after the last operator application:
//   step 1:      x = (x + z_1) + z_2   (the sums in the order of src/chebyshev.hpp:73)
//   step i >= 2: x += z_{i+1}
template <bool NT, bool BOTH>
struct ChebStepF
{ // r -= q ; z_new = c1 z + c2 dinv r ; x += (z if BOTH) + z_new   (src/chebyshev.hpp:73-83)
  double* x;
  double* r;
  double* z;
  const double* q;
  const double* dinv;
  double c1, c2;
  int x_final; // 1, 2: the last correction: x is gathered next (apply / prolongation), keep it in cache;
               // 2: nobody reads r and z after this step (no residual wanted): they are not written
  __device__ void pair(int i) const
  {
    double2 vx = ld2<NT>(x, i), vr = ld2<NT>(r, i), vz = D2(z)[i];
    double2 vq = ld2<NT>(q, i), vd = ld2<NT>(dinv, i);
    if constexpr (BOTH)
    {
      vx.x += vz.x;
      vx.y += vz.y;
    }
    vr.x -= vq.x;
    vr.y -= vq.y;
    vz.x = c1 * vz.x + c2 * vd.x * vr.x;
    vz.y = c1 * vz.y + c2 * vd.y * vr.y;
    vx.x += vz.x;
    vx.y += vz.y;
    if (x_final)
      D2(x)[i] = vx;
    else
      st2<NT>(x, i, vx);
    if (x_final != 2)
    {
      st2<NT>(r, i, vr);
      D2(z)[i] = vz;
    }
  }
  __device__ void one(int i) const
  {
    double vz = z[i], vx = x[i];
    if constexpr (BOTH)
      vx += vz;
    double vr = r[i] - q[i];
    vz = c1 * vz + c2 * dinv[i] * vr;
    if (x_final != 2)
    {
      r[i] = vr;
      z[i] = vz;
    }
    x[i] = vx + vz;
  }
};
template <bool NT>
struct ChebFirstF
{ // first step from x == 0:  r -= q ; z_2 = c1 z_1 + c2 dinv r ; x = z_1 + z_2
  double* x;
  double* r;
  double* z;
  const double* q;
  const double* dinv;
  double c1, c2;
  int x_final;
  __device__ void pair(int i) const
  {
    double2 vr = ld2<NT>(r, i), vz = D2(z)[i];
    double2 vq = ld2<NT>(q, i), vd = ld2<NT>(dinv, i);
    double2 vx = vz;
    vr.x -= vq.x;
    vr.y -= vq.y;
    vz.x = c1 * vz.x + c2 * vd.x * vr.x;
    vz.y = c1 * vz.y + c2 * vd.y * vr.y;
    if (x_final)
      D2(x)[i] = make_double2(vx.x + vz.x, vx.y + vz.y);
    else
      st2<NT>(x, i, make_double2(vx.x + vz.x, vx.y + vz.y));
    if (x_final != 2)
    {
      st2<NT>(r, i, vr);
      D2(z)[i] = vz;
    }
  }
  __device__ void one(int i) const
  {
    double vz = z[i];
    const double vx = vz;
    double vr = r[i] - q[i];
    vz = c1 * vz + c2 * dinv[i] * vr;
    if (x_final != 2)
    {
      r[i] = vr;
      z[i] = vz;
    }
    x[i] = vx + vz;
  }
};
template <bool NT>
struct ChebResidualF
{ // after the last application when the residual is wanted: r -= q   (src/chebyshev.hpp:77)
  double* r;
  const double* q;
  __device__ void pair(int i) const
  {
    double2 vr = ld2<NT>(r, i), vq = ld2<NT>(q, i);
    D2(r)[i] = make_double2(vr.x - vq.x, vr.y - vq.y); // r is gathered next (restriction)
  }
  __device__ void one(int i) const { r[i] -= q[i]; }
};
template <bool NT>
struct ChebLastF
{ // the only step of a one-step smoother when x and the residual are wanted:  x (+)= z ; r -= q
  double* x;
  double* r;
  const double* z;
  const double* q;
  int assign; // x = z (first step from x == 0) instead of x += z
  __device__ void pair(int i) const
  {
    double2 vr = ld2<NT>(r, i), vz = ld2<NT>(z, i), vq = ld2<NT>(q, i);
    double2 vx = assign ? make_double2(0.0, 0.0) : ld2<NT>(x, i);
    D2(x)[i] = make_double2(vx.x + vz.x, vx.y + vz.y); // x is gathered next (prolongation)
    st2<NT>(r, i, make_double2(vr.x - vq.x, vr.y - vq.y));
  }
  __device__ void one(int i) const
  {
    x[i] = (assign ? 0.0 : x[i]) + z[i];
    r[i] -= q[i];
  }
};
template <bool NT>
struct AddF
{ // x += z
  double* x;
  const double* z;
  __device__ void pair(int i) const
  {
    double2 vx = ld2<NT>(x, i), vz = ld2<NT>(z, i);
    D2(x)[i] = make_double2(vx.x + vz.x, vx.y + vz.y); // x is gathered next (apply / prolongation)
  }
  __device__ void one(int i) const { x[i] += z[i]; }
};
struct MaskBcF
{ // b *= (1 - bc)   (src/pmg.hpp:100-103)
  double* b;
  const int8_t* bc;
  __device__ void pair(int i) const
  {
    if (bc[2 * i])
      b[2 * i] = 0.0;
    if (bc[2 * i + 1])
      b[2 * i + 1] = 0.0;
  }
  __device__ void one(int i) const
  {
    if (bc[i])
      b[i] = 0.0;
  }
};
struct CgUpdateF
{ // x += alpha p ; r -= alpha y ; y = dinv r   (src/cg.hpp:186-192), alpha = rnorm / *d_py (:182)
  double* x;
  double* r;
  double* y;
  const double* p;
  const double* dinv;
  double rnorm;
  const double* d_py;
  __device__ void pair(int i) const
  {
    const double alpha = rnorm / *d_py;
    double2 vx = D2(x)[i], vr = D2(r)[i], vy = D2(y)[i];
    double2 vp = CD2(p)[i], vd = CD2(dinv)[i];
    vx.x += alpha * vp.x;
    vx.y += alpha * vp.y;
    vr.x -= alpha * vy.x;
    vr.y -= alpha * vy.y;
    D2(x)[i] = vx;
    D2(r)[i] = vr;
    D2(y)[i] = make_double2(vd.x * vr.x, vd.y * vr.y);
  }
  __device__ void one(int i) const
  {
    const double alpha = rnorm / *d_py;
    x[i] += alpha * p[i];
    double vr = r[i] - alpha * y[i];
    r[i] = vr;
    y[i] = dinv[i] * vr;
  }
};
struct CgUpdate2F
{ // x += alpha p ; r -= alpha y   (src/cg.hpp:186-189) when the preconditioner is not the diagonal
  double* x;
  double* r;
  const double* p;
  const double* y;
  double rnorm;
  const double* d_py;
  __device__ void pair(int i) const
  {
    const double alpha = rnorm / *d_py;
    double2 vx = D2(x)[i], vr = D2(r)[i], vy = CD2(y)[i], vp = CD2(p)[i];
    D2(x)[i] = make_double2(vx.x + alpha * vp.x, vx.y + alpha * vp.y);
    D2(r)[i] = make_double2(vr.x - alpha * vy.x, vr.y - alpha * vy.y);
  }
  __device__ void one(int i) const
  {
    const double alpha = rnorm / *d_py;
    x[i] += alpha * p[i];
    r[i] -= alpha * y[i];
  }
};
struct CgDirectionF
{ // p = beta p + y   (src/cg.hpp:196,211), beta = (*d_new - *d_sub) / rnorm (d_sub: flexible variant)
  double* p;
  const double* y;
  double rnorm;
  const double* d_new;
  const double* d_sub;
  __device__ void pair(int i) const
  {
    const double beta = (*d_new - (d_sub ? *d_sub : 0.0)) / rnorm;
    double2 vp = D2(p)[i], vy = CD2(y)[i];
    D2(p)[i] = make_double2(beta * vp.x + vy.x, beta * vp.y + vy.y);
  }
  __device__ void one(int i) const
  {
    const double beta = (*d_new - (d_sub ? *d_sub : 0.0)) / rnorm;
    p[i] = beta * p[i] + y[i];
  }
};

// ---- reductions: block partials, then one block folds them (deterministic) ----
__device__ inline double wave_sum(double v)
{
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_down(v, off, 64);
  return v;
}
__device__ inline double wave_max(double v)
{
  for (int off = 32; off > 0; off >>= 1)
    v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

template <bool MAX>
__device__ inline double block_reduce(double v)
{
  __shared__ double sm[RED_THREADS / 64];
  v = MAX ? wave_max(v) : wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0)
    sm[w] = v;
  __syncthreads();
  if (w == 0)
  {
    v = lane < RED_THREADS / 64 ? sm[lane] : (MAX ? 0.0 : 0.0);
    v = MAX ? wave_max(v) : wave_sum(v);
  }
  return v;
}

__global__ void dot_partial_kernel(int n, const double* __restrict__ a,
                                   const double* __restrict__ b, double* __restrict__ partials,
                                   int vec)
{
  double acc = 0.0;
  if (vec)
  {
    int n2 = n / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x)
    {
      double2 va = CD2(a)[i], vb = CD2(b)[i];
      acc += va.x * vb.x + va.y * vb.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
      acc += a[n - 1] * b[n - 1];
  }
  else
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
      acc += a[i] * b[i];
  acc = block_reduce<false>(acc);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = acc;
}

__global__ void absmax_partial_kernel(int n, const double* __restrict__ a,
                                      double* __restrict__ partials)
{
  double acc = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc = fmax(acc, fabs(a[i]));
  acc = block_reduce<true>(acc);
  if (threadIdx.x == 0)
    partials[blockIdx.x] = acc;
}

template <bool MAX>
__global__ void fold_kernel(int nb, const double* __restrict__ partials, double* __restrict__ out)
{
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x)
    acc = MAX ? fmax(acc, partials[i]) : acc + partials[i];
  acc = block_reduce<MAX>(acc);
  if (threadIdx.x == 0)
    out[0] = acc;
}
} // namespace

namespace pmg
{
// a level streams (nt policy of the smoother kernels) when its vectors cannot stay in the
// 256 MB MALL between kernels: from 4 M dofs (32 MB per vector, eight vectors in play)
static inline bool streams(int n) { return n >= (4 << 20); }

void launch_axpy(int n, double* r, double alpha, const double* x, const double* y, hipStream_t s)
{
  ew_launch(n, aligned16(r) && aligned16(x) && aligned16(y), AxpyF{r, alpha, x, y}, s);
}
void launch_pointwise(int n, double* w, const double* x, const double* y, hipStream_t s)
{
  ew_launch(n, aligned16(w) && aligned16(x) && aligned16(y), MulF{w, x, y}, s);
}
// clear_q (optional): the operator's output vector, zeroed over [0, n_total) behind the update (ThenClearF)
void launch_cheb_init(int n, double* r, double* z, const double* b, const double* q,
                      const double* dinv, double c0, hipStream_t s, double* clear_q, int n_total)
{
  bool v = aligned16(r) && aligned16(z) && aligned16(b) && aligned16(dinv) && (!q || aligned16(q));
  if (clear_q)
  {
    if (streams(n))
      ew_launch_clear(n, n_total, v, ChebInitF<true>{r, z, b, q, dinv, c0}, clear_q, s);
    else
      ew_launch_clear(n, n_total, v, ChebInitF<false>{r, z, b, q, dinv, c0}, clear_q, s);
  }
  else if (streams(n))
    ew_launch(n, v, ChebInitF<true>{r, z, b, q, dinv, c0}, s);
  else
    ew_launch(n, v, ChebInitF<false>{r, z, b, q, dinv, c0}, s);
}
void launch_cheb_step(int n, double* x, double* r, double* z, const double* q, const double* dinv,
                      double c1, double c2, bool both, int x_final, hipStream_t s, double* clear_q, int n_total)
{
  bool v = aligned16(x) && aligned16(r) && aligned16(z) && aligned16(q) && aligned16(dinv);
  const int xf = x_final;
  if (clear_q) // small levels only (merged-launch operators): no streaming variants needed
  {
    if (both)
      ew_launch_clear(n, n_total, v, ChebStepF<false, true>{x, r, z, q, dinv, c1, c2, xf}, clear_q, s);
    else
      ew_launch_clear(n, n_total, v, ChebStepF<false, false>{x, r, z, q, dinv, c1, c2, xf}, clear_q, s);
    return;
  }
  if (streams(n))
  {
    if (both)
      ew_launch(n, v, ChebStepF<true, true>{x, r, z, q, dinv, c1, c2, xf}, s);
    else
      ew_launch(n, v, ChebStepF<true, false>{x, r, z, q, dinv, c1, c2, xf}, s);
  }
  else if (both)
    ew_launch(n, v, ChebStepF<false, true>{x, r, z, q, dinv, c1, c2, xf}, s);
  else
    ew_launch(n, v, ChebStepF<false, false>{x, r, z, q, dinv, c1, c2, xf}, s);
}
void launch_cheb_residual(int n, double* r, const double* q, hipStream_t s)
{
  if (streams(n))
    ew_launch(n, aligned16(r) && aligned16(q), ChebResidualF<true>{r, q}, s);
  else
    ew_launch(n, aligned16(r) && aligned16(q), ChebResidualF<false>{r, q}, s);
}
void launch_cheb_first(int n, double* x, double* r, double* z, const double* q, const double* dinv,
                       double c1, double c2, int x_final, hipStream_t s, double* clear_q, int n_total)
{
  bool v = aligned16(x) && aligned16(r) && aligned16(z) && aligned16(q) && aligned16(dinv);
  const int xf = x_final;
  if (clear_q)
    ew_launch_clear(n, n_total, v, ChebFirstF<false>{x, r, z, q, dinv, c1, c2, xf}, clear_q, s);
  else if (streams(n))
    ew_launch(n, v, ChebFirstF<true>{x, r, z, q, dinv, c1, c2, xf}, s);
  else
    ew_launch(n, v, ChebFirstF<false>{x, r, z, q, dinv, c1, c2, xf}, s);
}
void launch_cheb_last(int n, double* x, double* r, const double* z, const double* q, bool assign,
                      hipStream_t s)
{
  bool v = aligned16(x) && aligned16(r) && aligned16(z) && aligned16(q);
  if (streams(n))
    ew_launch(n, v, ChebLastF<true>{x, r, z, q, assign ? 1 : 0}, s);
  else
    ew_launch(n, v, ChebLastF<false>{x, r, z, q, assign ? 1 : 0}, s);
}
void launch_add(int n, double* x, const double* z, hipStream_t s)
{
  if (streams(n))
    ew_launch(n, aligned16(x) && aligned16(z), AddF<true>{x, z}, s);
  else
    ew_launch(n, aligned16(x) && aligned16(z), AddF<false>{x, z}, s);
}
// x[0..n) = 0 in ONE kernel (hipMemsetAsync issues two, ~5 us apiece on a small level)
void launch_zero(int n, double* x, hipStream_t s) { ew_launch(n, aligned16(x), SetF{x, 0.0}, s); }
void launch_mask_bc(int n, double* b, const int8_t* bc, hipStream_t s)
{
  ew_launch(n, true, MaskBcF{b, bc}, s);
}
void launch_cg_update(int n, double* x, double* r, double* y, const double* p, const double* dinv,
                      double rnorm, const double* d_py, hipStream_t s)
{
  bool v = aligned16(x) && aligned16(r) && aligned16(y) && aligned16(p) && aligned16(dinv);
  ew_launch(n, v, CgUpdateF{x, r, y, p, dinv, rnorm, d_py}, s);
}
void launch_cg_update2(int n, double* x, double* r, const double* p, const double* y, double rnorm,
                       const double* d_py, hipStream_t s)
{
  bool v = aligned16(x) && aligned16(r) && aligned16(y) && aligned16(p);
  ew_launch(n, v, CgUpdate2F{x, r, p, y, rnorm, d_py}, s);
}
void launch_cg_direction(int n, double* p, const double* y, double rnorm, const double* d_new,
                         const double* d_sub, hipStream_t s)
{
  ew_launch(n, aligned16(p) && aligned16(y), CgDirectionF{p, y, rnorm, d_new, d_sub}, s);
}

double* red_slot(pmg_layout l, int slot) { return l->d_partials + RED_BLOCKS + slot; }

// local (this rank) dot of the owned entries -> result slot, stream-ordered
int dot_async(pmg_layout l, const double* a, const double* b, int slot, hipStream_t s)
{
  PMG_REQUIRE(slot >= 0 && slot < RED_SLOTS, "dot_async: bad slot");
  int n = l->size_local;
  int nb = ew_blocks((n + 1) / 2);
  if (nb > RED_BLOCKS)
    nb = RED_BLOCKS;
  int vec = aligned16(a) && aligned16(b);
  dot_partial_kernel<<<nb, RED_THREADS, 0, s>>>(n, a, b, l->d_partials, vec);
  fold_kernel<false><<<1, RED_THREADS, 0, s>>>(nb, l->d_partials, red_slot(l, slot));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

// Sum (or maximise) the result slots [slot, slot + n) over the ranks; afterwards the DEVICE slots
// hold the global values, so kernels may consume them without a host round trip.  With a
// communicator this is one stream-ordered ncclAllReduce; with the caller's callback (MPI) the
// values travel through the host, which costs a stream synchronisation.
int reduce_slots_async(pmg_layout l, int slot, int n, bool max, hipStream_t s)
{
  PMG_REQUIRE(slot >= 0 && n >= 1 && slot + n <= RED_SLOTS, "reduce_slots_async: bad slot range");
  if (l->comm)
    return comm_allreduce(l, red_slot(l, slot), n, max, s);
  pmg_allreduce_fn fn = max ? l->allreduce_max : l->allreduce;
  if (!fn)
    return PMG_OK; // single rank
  PMG_HIP(hipMemcpyAsync(l->h_result, red_slot(l, slot), sizeof(double) * n, hipMemcpyDeviceToHost, s));
  PMG_HIP(hipStreamSynchronize(s));
  if (fn(l->user, l->h_result, n) != 0)
    return fail(PMG_ERR_INVALID, "allreduce callback failed");
  PMG_HIP(hipMemcpyAsync(red_slot(l, slot), l->h_result, sizeof(double) * n, hipMemcpyHostToDevice, s));
  PMG_HIP(hipStreamSynchronize(s)); // h_result is reused by the next reduction
  return PMG_OK;
}

// ... and bring them to the host: the one host synchronisation of a reduction
int fetch_slots(pmg_layout l, int slot, int n, double* host_out, hipStream_t s)
{
  PMG_REQUIRE(slot >= 0 && n >= 1 && slot + n <= RED_SLOTS, "fetch_slots: bad slot range");
  PMG_HIP(hipMemcpyAsync(l->h_result, red_slot(l, slot), sizeof(double) * n, hipMemcpyDeviceToHost, s));
  PMG_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i)
    host_out[i] = l->h_result[i];
  return PMG_OK;
}

// inner_product of src/vector.hpp:334-352: local dot, all-reduce, host value
int dot_host(pmg_layout l, const double* a, const double* b, double* result, hipStream_t s)
{
  PMG_TRY(dot_async(l, a, b, 0, s));
  PMG_TRY(reduce_slots_async(l, 0, 1, false, s));
  return fetch_slots(l, 0, 1, result, s);
}
} // namespace pmg

// ------------------------------------------------------------------ C ABI --
extern "C" int pmg_layout_create(pmg_layout* out, int32_t size_local, int32_t num_ghosts,
                                 int32_t n_send, const int32_t* send_indices, double* send_buffer,
                                 int32_t n_recv, const int32_t* recv_indices, double* recv_buffer,
                                 pmg_exchange_fn exchange, pmg_allreduce_fn allreduce_sum,
                                 void* user)
{
  PMG_REQUIRE(out, "pmg_layout_create: out is NULL");
  PMG_REQUIRE(size_local >= 0 && num_ghosts >= 0 && n_send >= 0 && n_recv >= 0,
              "pmg_layout_create: negative size");
  PMG_REQUIRE(n_recv <= num_ghosts, "pmg_layout_create: n_recv (%d) > num_ghosts (%d)", n_recv,
              num_ghosts);
  PMG_REQUIRE(n_send == 0 || (send_indices && send_buffer),
              "pmg_layout_create: send arrays missing");
  PMG_REQUIRE(n_recv == 0 || (recv_indices && recv_buffer),
              "pmg_layout_create: recv arrays missing");
  // (a layout with neighbours needs an exchange: the callback here, or pmg_layout_set_comm later;
  // the scatter calls check)
  // with a callback every scatter calls it (it is a collective on the caller's
  // side), even when this rank shares no dof with anybody
  auto* l = new pmg_layout_s;
  l->size_local = size_local;
  l->num_ghosts = num_ghosts;
  l->n_send = n_send;
  l->n_recv = n_recv;
  l->send_idx = send_indices;
  l->recv_idx = recv_indices;
  l->send_buf = send_buffer;
  l->recv_buf = recv_buffer;
  l->exchange = exchange;
  l->allreduce = allreduce_sum;
  l->user = user;
  hipError_t e = hipMalloc(&l->d_partials, sizeof(double) * (RED_BLOCKS + RED_SLOTS));
  if (e == hipSuccess)
    e = hipHostMalloc(&l->h_result, sizeof(double) * RED_SLOTS, hipHostMallocDefault);
  if (e != hipSuccess)
  {
    delete l;
    return fail(PMG_ERR_HIP, "pmg_layout_create: %s", hipGetErrorString(e));
  }
  *out = l;
  return PMG_OK;
}

extern "C" int pmg_layout_destroy(pmg_layout l)
{
  if (!l)
    return PMG_OK;
  (void)hipFree(l->d_partials);
  (void)hipHostFree(l->h_result);
  if (l->ev_packed)
    (void)hipEventDestroy(l->ev_packed);
  if (l->ev_arrived)
    (void)hipEventDestroy(l->ev_arrived);
  window_destroy(l);
  (void)hipFree(l->c_send);
  (void)hipFree(l->c_recv);
  (void)hipFree(l->send_pos);
  (void)hipFree(l->recv_pos);
  delete l;
  return PMG_OK;
}

extern "C" int32_t pmg_layout_size_local(pmg_layout l) { return l ? l->size_local : -1; }
extern "C" int32_t pmg_layout_num_ghosts(pmg_layout l) { return l ? l->num_ghosts : -1; }
extern "C" int pmg_layout_set_allreduce_max(pmg_layout l, pmg_allreduce_fn allreduce_max)
{
  PMG_REQUIRE(l, "pmg_layout_set_allreduce_max: NULL layout");
  l->allreduce_max = allreduce_max;
  return PMG_OK;
}

// src/vector.hpp:186-207
// forward scatters issued on the layout since its creation (a captured cycle counts once, at capture)
extern "C" long long pmg_layout_forward_scatters(pmg_layout l) { return l ? l->fwd_scatters : -1; }

extern "C" int pmg_scatter_fwd_begin(pmg_layout l, const double* x, pmg_stream stream)
{
  PMG_REQUIRE(l && x, "pmg_scatter_fwd_begin: NULL argument");
  l->fwd_scatters++;
  if (l->win)
    return window_exchange_begin(l, false, x, S(stream));
  if (!l->comm && !l->exchange)
  {
    PMG_REQUIRE(l->n_send == 0 && l->n_recv == 0,
                "scatter: the layout has neighbours but neither a communicator nor an exchange callback");
    return PMG_OK; // single rank
  }
  if (l->n_send > 0)
  {
    if (l->comm)
      pack_pos_kernel<<<ew_blocks(l->n_send), EW_THREADS, 0, S(stream)>>>(l->n_send, l->send_idx, l->send_pos, x,
                                                                          l->c_send);
    else
      pack_kernel<<<ew_blocks(l->n_send), EW_THREADS, 0, S(stream)>>>(l->n_send, l->send_idx, x, l->send_buf);
  }
  PMG_HIP(hipGetLastError());
  if (l->comm)
    return comm_exchange_begin(l, false, S(stream));
  if (l->exchange(l->user, 0, stream) != 0)
    return fail(PMG_ERR_INVALID, "exchange callback (begin) failed");
  return PMG_OK;
}

// src/vector.hpp:209-238
extern "C" int pmg_scatter_fwd_end(pmg_layout l, double* x, pmg_stream stream)
{
  PMG_REQUIRE(l && x, "pmg_scatter_fwd_end: NULL argument");
  if (l->win)
    return window_exchange_end(l, false, x, S(stream));
  if (!l->comm && !l->exchange)
    return PMG_OK; // single rank
  if (l->comm)
    PMG_TRY(comm_exchange_end(l, S(stream)));
  else if (l->exchange(l->user, 1, stream) != 0)
    return fail(PMG_ERR_INVALID, "exchange callback (end) failed");
  if (l->n_recv > 0)
  {
    if (l->comm)
      unpack_pos_kernel<<<ew_blocks(l->n_recv), EW_THREADS, 0, S(stream)>>>(l->n_recv, l->recv_idx, l->recv_pos,
                                                                            l->c_recv, x + l->size_local);
    else
      unpack_kernel<<<ew_blocks(l->n_recv), EW_THREADS, 0, S(stream)>>>(l->n_recv, l->recv_idx, l->recv_buf,
                                                                        x + l->size_local);
  }
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

namespace pmg
{
int scatter_fwd_whole(pmg_layout l, double* x, hipStream_t s)
{
  if (l->win)
  {
    l->fwd_scatters++;
    return window_exchange_whole(l, x, s);
  }
  PMG_TRY(pmg_scatter_fwd_begin(l, x, (pmg_stream)s));
  return pmg_scatter_fwd_end(l, x, (pmg_stream)s);
}
bool layout_exchanges_whole(pmg_layout l)
{
  if (!l->win)
    return false;
  const char* e = std::getenv("PMG_FUSED_EXCHANGE");
  return !(e && e[0] == '0');
}
} // namespace pmg

// src/vector.hpp:249-267: ghosts are packed into the *recv* buffer and travel
// back to their owners, who receive them in the *send* buffer.
extern "C" int pmg_scatter_rev_begin(pmg_layout l, const double* x, pmg_stream stream)
{
  PMG_REQUIRE(l && x, "pmg_scatter_rev_begin: NULL argument");
  if (l->win)
    return window_exchange_begin(l, true, x, S(stream));
  if (!l->comm && !l->exchange)
  {
    PMG_REQUIRE(l->n_send == 0 && l->n_recv == 0,
                "scatter: the layout has neighbours but neither a communicator nor an exchange callback");
    return PMG_OK; // single rank
  }
  if (l->n_recv > 0)
  {
    if (l->comm)
      pack_pos_kernel<<<ew_blocks(l->n_recv), EW_THREADS, 0, S(stream)>>>(l->n_recv, l->recv_idx, l->recv_pos,
                                                                          x + l->size_local, l->c_recv);
    else
      pack_kernel<<<ew_blocks(l->n_recv), EW_THREADS, 0, S(stream)>>>(l->n_recv, l->recv_idx, x + l->size_local,
                                                                      l->recv_buf);
  }
  PMG_HIP(hipGetLastError());
  if (l->comm)
    return comm_exchange_begin(l, true, S(stream));
  if (l->exchange(l->user, 2, stream) != 0)
    return fail(PMG_ERR_INVALID, "exchange callback (rev begin) failed");
  return PMG_OK;
}

// src/vector.hpp:270-286
extern "C" int pmg_scatter_rev_end(pmg_layout l, double* x, pmg_stream stream)
{
  PMG_REQUIRE(l && x, "pmg_scatter_rev_end: NULL argument");
  if (l->win)
    return window_exchange_end(l, true, x, S(stream));
  if (!l->comm && !l->exchange)
    return PMG_OK; // single rank
  if (l->comm)
    PMG_TRY(comm_exchange_end(l, S(stream)));
  else if (l->exchange(l->user, 3, stream) != 0)
    return fail(PMG_ERR_INVALID, "exchange callback (rev end) failed");
  if (l->n_send > 0)
  {
    if (l->comm)
      unpack_add_pos_kernel<<<ew_blocks(l->n_send), EW_THREADS, 0, S(stream)>>>(l->n_send, l->send_idx, l->send_pos,
                                                                                l->c_send, x);
    else
      unpack_add_kernel<<<ew_blocks(l->n_send), EW_THREADS, 0, S(stream)>>>(l->n_send, l->send_idx, l->send_buf, x);
  }
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_vec_set(pmg_layout l, double* x, double value, pmg_stream stream)
{
  PMG_REQUIRE(l && x, "pmg_vec_set: NULL argument");
  ew_launch(l->total(), aligned16(x), SetF{x, value}, S(stream));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_vec_scale(pmg_layout l, double* x, double alpha, pmg_stream stream)
{
  PMG_REQUIRE(l && x, "pmg_vec_scale: NULL argument");
  ew_launch(l->total(), aligned16(x), ScaleF{x, alpha}, S(stream));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_vec_copy(pmg_layout l, double* dst, const double* src, pmg_stream stream)
{
  PMG_REQUIRE(l && dst && src, "pmg_vec_copy: NULL argument");
  ew_launch(l->size_local, aligned16(dst) && aligned16(src), CopyF{dst, src}, S(stream));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_vec_axpy(pmg_layout l, double* r, double alpha, const double* x,
                            const double* y, pmg_stream stream)
{
  PMG_REQUIRE(l && r && x && y, "pmg_vec_axpy: NULL argument");
  launch_axpy(l->size_local, r, alpha, x, y, S(stream));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_vec_pointwise_mult(pmg_layout l, double* w, const double* x, const double* y,
                                      pmg_stream stream)
{
  PMG_REQUIRE(l && w && x && y, "pmg_vec_pointwise_mult: NULL argument");
  launch_pointwise(l->size_local, w, x, y, S(stream));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_vec_inner_product(pmg_layout l, const double* a, const double* b,
                                     double* result, pmg_stream stream)
{
  PMG_REQUIRE(l && a && b && result, "pmg_vec_inner_product: NULL argument");
  return dot_host(l, a, b, result, S(stream));
}

extern "C" int pmg_vec_squared_norm(pmg_layout l, const double* a, double* result,
                                    pmg_stream stream)
{
  return pmg_vec_inner_product(l, a, a, result, stream);
}

extern "C" int pmg_vec_norm(pmg_layout l, const double* a, int norm_type, double* result,
                            pmg_stream stream)
{
  PMG_REQUIRE(l && a && result, "pmg_vec_norm: NULL argument");
  if (norm_type == 0)
  {
    double v;
    PMG_TRY(dot_host(l, a, a, &v, S(stream)));
    *result = sqrt(v);
    return PMG_OK;
  }
  PMG_REQUIRE(norm_type == 1, "Norm type not supported"); // src/vector.hpp:388
  PMG_REQUIRE(l->comm || !l->allreduce || l->allreduce_max,
              "pmg_vec_norm: linf over several ranks needs pmg_layout_set_allreduce_max");
  int nb = ew_blocks(l->size_local);
  if (nb > RED_BLOCKS)
    nb = RED_BLOCKS;
  absmax_partial_kernel<<<nb, RED_THREADS, 0, S(stream)>>>(l->size_local, a, l->d_partials);
  fold_kernel<true><<<1, RED_THREADS, 0, S(stream)>>>(nb, l->d_partials, red_slot(l, 0));
  PMG_HIP(hipGetLastError());
  PMG_TRY(reduce_slots_async(l, 0, 1, true, S(stream)));
  double v;
  PMG_TRY(fetch_slots(l, 0, 1, &v, S(stream)));
  *result = v;
  return PMG_OK;
}
