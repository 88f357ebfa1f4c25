// The x / y contractions of the stiffness kernel at nd = 5, two ways of getting the other lanes' values: through a
// wave-private LDS slice (what every kernel form of this library does: 5 ds_read_b64 per value and direction) or with
// DPP row shifts inside the wavefront (2 x 8 v_mov_b32 dpp + 9 v_fma_f64 with 9 coefficient registers per value and
// direction, groups of five lanes laid out three to a 16-lane row).   hipcc --offload-arch=gfx950 -O3 tools/dpp_probe.hip
//
// Sixteen wavefronts per compute unit (four 256-thread workgroups), every unit busy, a dependent chain of `iters`
// "layers" per wavefront; a layer is two contractions (forward x and y; the backward pair costs the same):
//   LL both through LDS (the kernel today)     LD one through LDS, one with DPP     DD both with DPP
// The LDS pipe is shared by the four SIMDs of a unit, the vector ALUs are not: what does a unit sustain?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true); // bound_ctrl: lanes outside the row read zero
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// out[b] = sum_m D[b][m] v[m] inside the lane's group of five, via the slice
__device__ __forceinline__ double contract_lds(double v, double* sl, int lane, int gbase, const double (&D)[5])
{
  sl[lane] = v;
  wave_fence();
  double acc = 0.0;
#pragma unroll
  for (int m = 0; m < 5; ++m)
    acc += D[m] * sl[gbase + m];
  wave_fence();
  return acc;
}

// the same with row shifts: C[s + 4] = D[b][b + s] where 0 <= b + s < 5, else 0
__device__ __forceinline__ double contract_dpp(double v, const double (&C)[9])
{
  double acc = C[4] * v;
  acc += C[0] * dpp_move<0x114>(v); // row_shr:4  -> v[b - 4]
  acc += C[1] * dpp_move<0x113>(v);
  acc += C[2] * dpp_move<0x112>(v);
  acc += C[3] * dpp_move<0x111>(v);
  acc += C[5] * dpp_move<0x101>(v); // row_shl:1  -> v[b + 1]
  acc += C[6] * dpp_move<0x102>(v);
  acc += C[7] * dpp_move<0x103>(v);
  acc += C[8] * dpp_move<0x104>(v);
  return acc;
}

template <int FORM> // 0 = LL, 1 = LD, 2 = DD
__global__ void __launch_bounds__(256) layers(double* out, const double* __restrict__ Dg, int iters, int check)
{
  __shared__ double sl_all[4 * 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int row = lane >> 4, inrow = lane & 15, g = inrow / 5, b = inrow - 5 * g; // lane 15 of a row: g = 3, idle
  const bool live = inrow < 15;
  const int gbase = row * 16 + g * 5;
  double* sl = sl_all + wave * 64;
  double D[5], C[9];
#pragma unroll
  for (int m = 0; m < 5; ++m)
    D[m] = live ? Dg[b * 5 + m] : 0.0;
#pragma unroll
  for (int s = -4; s <= 4; ++s)
    C[s + 4] = (live && b + s >= 0 && b + s < 5) ? Dg[b * 5 + b + s] : 0.0;
  double v = live ? 1.0 + 0.01 * lane : 0.0;
  for (int it = 0; it < iters; ++it)
  {
    double w;
    if constexpr (FORM == 0)
    {
      w = contract_lds(v, sl, lane, gbase, D);
      v = contract_lds(w, sl, lane, gbase, D);
    }
    else if constexpr (FORM == 1)
    {
      w = contract_lds(v, sl, lane, gbase, D);
      v = contract_dpp(w, C);
    }
    else
    {
      w = contract_dpp(v, C);
      v = contract_dpp(w, C);
    }
  }
  if (check)
    out[(size_t)blockIdx.x * 256 + t] = v;
  else if (v == 12345.678)
    out[0] = v;
}

template <typename F>
double timeit(F f, int reps = 5)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * 4, iters = 4000;
  // a contraction matrix with spectral radius < 1 (the chain must not overflow): 0.18 everywhere, 0.2 on the diagonal
  double Dh[25];
  for (int i = 0; i < 25; ++i)
    Dh[i] = (i % 6 == 0) ? 0.2 : 0.18 - 0.01 * (i % 5);
  double *Dg, *out, *ref;
  CK(hipMalloc(&Dg, sizeof(Dh)));
  CK(hipMemcpy(Dg, Dh, sizeof(Dh), hipMemcpyHostToDevice));
  CK(hipMalloc(&out, sizeof(double) * blocks * 256));
  CK(hipMalloc(&ref, sizeof(double) * blocks * 256));
  // the three forms compute the same numbers (a short chain, compared lane by lane)
  layers<0><<<blocks, 256>>>(ref, Dg, 7, 1);
  double worst[3] = {0, 0, 0};
  static double h0[1 << 20], h1[1 << 20];
  CK(hipMemcpy(h0, ref, sizeof(double) * 1024, hipMemcpyDeviceToHost));
  layers<1><<<blocks, 256>>>(out, Dg, 7, 1);
  CK(hipMemcpy(h1, out, sizeof(double) * 1024, hipMemcpyDeviceToHost));
  for (int i = 0; i < 1024; ++i)
    worst[1] = fmax(worst[1], fabs(h1[i] - h0[i]) / (fabs(h0[i]) + 1e-300));
  layers<2><<<blocks, 256>>>(out, Dg, 7, 1);
  CK(hipMemcpy(h1, out, sizeof(double) * 1024, hipMemcpyDeviceToHost));
  for (int i = 0; i < 1024; ++i)
    worst[2] = fmax(worst[2], fabs(h1[i] - h0[i]) / (fabs(h0[i]) + 1e-300));
  printf("agreement with the LDS form after 7 layers: LD %.2e, DD %.2e (relative)\n", worst[1], worst[2]);
  const double ms0 = timeit([&] { layers<0><<<blocks, 256>>>(out, Dg, iters, 0); });
  const double ms1 = timeit([&] { layers<1><<<blocks, 256>>>(out, Dg, iters, 0); });
  const double ms2 = timeit([&] { layers<2><<<blocks, 256>>>(out, Dg, iters, 0); });
  const double clk = prop.clockRate * 1e3; // Hz
  auto report = [&](const char* name, double ms) {
    // per compute unit: 16 wavefronts x iters layers x 2 contractions
    const double per_unit = ms * 1e-3 / (16.0 * iters * 2.0);
    printf("%s: %8.3f ms   %6.1f ns per contraction of a wavefront (16 per unit in flight) = %5.1f unit clocks at %.2f GHz\n", name,
           ms, 16.0 * per_unit * 1e9, per_unit * clk, clk * 1e-9);
  };
  printf("%d units, %d workgroups of 256, %d layers of two contractions per wavefront\n", cus, blocks, iters);
  report("LL (both through LDS)   ", ms0);
  report("LD (one LDS, one DPP)   ", ms1);
  report("DD (both DPP)           ", ms2);
  return 0;
}
