// FP64 issue rates on MI355X: v_fma_f64 against v_mfma_f64_16x16x4_f64, and the one contraction of the stiffness
// kernel an MFMA could take at nd = 8.   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
//
// (1) peak: independent chains, no memory.  (2) one direction of one cell at nd = 8, out[i][jk] = sum_m D[i][m] u[m][jk]
// (512 outputs x 8 terms), operands already where each form wants them:
//   VALU: a lane owns the column jk (64 lanes = the cell), u[0..7] and D in registers: 64 v_fma_f64 per lane;
//   MFMA: rows = 16 columns jk, K = m in two steps of 4, N = i (8 of the tile's 16 columns carry D^T, the rest are zero):
//         4 row tiles x 2 steps = 8 v_mfma_f64_16x16x4_f64.
// Neither form is charged for moving its operands into place (the column form needs none for this direction; the MFMA
// form needs u transposed across lanes: A[row = jk][k = m] has lane = 16 (m % 4) + jk % 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void fma_peak(double* out, int iters, double a, double b)
{
  double c[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    c[k] = threadIdx.x + k;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      c[k] = __builtin_fma(a, c[k], b);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    s += c[k];
  if (s == 12345.678)
    out[0] = s;
}

__global__ void mfma_peak(double* out, int iters, double a, double b)
{
  double4_t c[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    c[k] = double4_t{(double)threadIdx.x, 1.0, 2.0, (double)k};
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[k], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    s += c[k].x + c[k].y + c[k].z + c[k].w;
  if (s == 12345.678)
    out[0] = s;
}

// one direction of `cells` cells per wave, VALU column form
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) contract_valu(double* out, int cells, const double* __restrict__ Dg, int zero)
{
  double D[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int m = 0; m < 8; ++m)
      D[i][m] = Dg[i * 8 + m + (threadIdx.x & zero)]; // vector registers, as the column kernel keeps its table rows
  double u[8], acc = 0;
#pragma unroll
  for (int m = 0; m < 8; ++m)
    u[m] = threadIdx.x * 0.001 + m;
  for (int c = 0; c < cells; ++c)
  {
    double r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
      r[i] = D[i][0] * u[0];
#pragma unroll
      for (int m = 1; m < 8; ++m)
        r[i] = __builtin_fma(D[i][m], u[m], r[i]);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
      u[m] = r[m] * 0.125; // the next cell's input depends on this one's output: nothing can be hoisted
    acc += r[0];
  }
  if (acc == 12345.678)
    out[0] = acc;
}

// the same contraction as 8 MFMAs per cell: 4 row tiles (16 columns jk each) x 2 K steps
__global__ void contract_mfma(double* out, int cells, const double* __restrict__ Dg)
{
  const int lane = threadIdx.x & 63;
  // B[k][n] = D^T[m = 4 s + k][i = n] = D[n][4 s + k], n < 8; lane = 16 k + n
  double B[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
    B[s] = (lane & 15) < 8 ? Dg[(lane & 15) * 8 + 4 * s + (lane >> 4)] : 0.0;
  // A[row][k] of tile t, step s: u[m = 4 s + k][jk = 16 t + row]; lane = 16 k + row: one value per (t, s)
  double A[4][2], acc = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      A[t][s] = lane * 0.001 + t + 4 * s;
  for (int c = 0; c < cells; ++c)
  {
    double4_t r[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
    {
      r[t] = double4_t{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 2; ++s)
        r[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[t][s], B[s], r[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
    {
      A[t][0] = r[t].x * 0.125;
      A[t][1] = r[t].y * 0.125;
    }
    acc += (r[0].z + r[1].z) + (r[2].w + r[3].w); // every tile's result is used
  }
  if (acc == 12345.678)
    out[0] = acc;
}

template <typename F>
static double time_ms(F launch)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main()
{
  double *out, *D;
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&D, 64 * 8));
  double h[64];
  for (int i = 0; i < 64; ++i)
    h[i] = 0.01 * (i % 9) - 0.03;
  CK(hipMemcpy(D, h, sizeof(h), hipMemcpyHostToDevice));
  const int blocks = 256 * 8, threads = 256; // 8 workgroups of 4 waves per CU
  const double waves = (double)blocks * threads / 64;
  const int iters = 20000;
  double ms = time_ms([&] { fma_peak<<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); });
  printf("v_fma_f64, 8 chains per lane:            %7.1f TFLOP/s\n", waves * iters * 8.0 * 64 * 2 / ms / 1e9);
  ms = time_ms([&] { mfma_peak<<<blocks, threads>>>(out, iters, 1.0000001, 1e-9); });
  printf("v_mfma_f64_16x16x4_f64, 4 chains:        %7.1f TFLOP/s\n", waves * iters * 4.0 * 2048 / ms / 1e9);
  const int cells = 20000;
  ms = time_ms([&] { contract_valu<<<blocks, threads>>>(out, cells, D, 0); });
  const double useful = waves * cells * 512.0 * 8 * 2;
  printf("nd = 8, one direction of a cell, VALU column form (64 FMAs per lane):     %7.1f TFLOP/s useful\n", useful / ms / 1e9);
  ms = time_ms([&] { contract_mfma<<<blocks, threads>>>(out, cells, D); });
  printf("nd = 8, one direction of a cell, 8 x v_mfma_f64_16x16x4 (half-empty N): %7.1f TFLOP/s useful\n", useful / ms / 1e9);
  return 0;
}
