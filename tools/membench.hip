// Read-bandwidth micro-benchmarks on MI355X: what the memory system delivers for
// the access shapes of the stiffness kernel.  hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// (a) grid-stride linear read, 16 B/lane
__global__ void read_linear(const double2* __restrict__ a, size_t n, double* out)
{
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 12345.678)
    out[0] = s;
}

// (b) each wave streams its own contiguous chunk of CH bytes in pieces of 50 lanes x 16 B (2 x 400 B
//     segments 6000 B apart), DEPTH pieces in flight, like the column kernel
template <int DEPTH>
__global__ void read_waves(const double2* __restrict__ a, size_t nchunks, double* out)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  double s = 0;
  for (size_t chunk = (size_t)blockIdx.x * nw + wave; chunk < nchunks; chunk += (size_t)gridDim.x * nw)
  {
    // chunk = 2 cells x 6000 B = 750 double2; lane l<25 -> cell 0, 25..49 -> cell 1
    const int l = lane < 50 ? lane : 49;
    const double2* base = a + chunk * 750 + (l / 25) * 375 + (l % 25);
    double2 v[DEPTH * 3];
#pragma unroll
    for (int d = 0; d < DEPTH * 3; ++d)
      v[d] = base[d * 25];
#pragma unroll
    for (int k = 0; k < 15; k += 3)
    {
#pragma unroll
      for (int j = 0; j < 3; ++j)
      {
        s += v[(k + j) % (DEPTH * 3)].x + v[(k + j) % (DEPTH * 3)].y;
        if (k + j + DEPTH * 3 < 15)
          v[(k + j) % (DEPTH * 3)] = base[(k + j + DEPTH * 3) * 25];
      }
    }
  }
  if (s == 12345.678)
    out[0] = s;
}

// (d) skeleton of the column kernel's G stream: one workgroup of 8 waves per patch (192 KB of G),
//     LDS_BYTES of LDS to pin the residency, wave w takes items w and w + 8, layer k + 1 in flight
template <int LDS_BYTES, int DEPTH>
__global__ void read_patch(const double2* __restrict__ a, double* out)
{
  __shared__ double lds[LDS_BYTES / 8 + 512];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (LDS_BYTES > 0)
    lds[t] = t;
  __syncthreads();
  double s = LDS_BYTES > 0 ? lds[(t * 7) % 512] : 0.0;
  const int l = lane < 50 ? lane : 49;
  for (int it = wave; it < 16; it += 8)
  {
    const double2* base = a + (size_t)blockIdx.x * 12000 + (size_t)it * 750 + (l / 25) * 375 + (l % 25);
    double2 v[DEPTH * 3];
#pragma unroll
    for (int d = 0; d < DEPTH * 3; ++d)
      v[d] = base[d * 25];
#pragma unroll
    for (int k = 0; k < 15; k += 3)
    {
#pragma unroll
      for (int j = 0; j < 3; ++j)
      {
        s += v[(k + j) % (DEPTH * 3)].x + v[(k + j) % (DEPTH * 3)].y;
        if (k + j + DEPTH * 3 < 15)
          v[(k + j) % (DEPTH * 3)] = base[(k + j + DEPTH * 3) * 25];
      }
    }
  }
  if (s == 12345.678)
    out[0] = s;
}

// (e) block tail: every workgroup busy-waits `spin` wall-clock ticks (100 MHz counter), then writes
//     NST x 512 x 8 bytes and exits; 52 KB of LDS pins 2 workgroups per CU
template <int NST>
__global__ void store_tail(double* __restrict__ y, int spin_ticks)
{
  __shared__ double lds[53248 / 8];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks)
    __builtin_amdgcn_s_sleep(8);
  double v = lds[(threadIdx.x * 7) % 512];
#pragma unroll
  for (int k = 0; k < NST; ++k)
    y[(size_t)blockIdx.x * (NST * 512) + k * 512 + threadIdx.x] = v + k;
}

// (c) copy
__global__ void copy_linear(const double2* __restrict__ a, double2* __restrict__ b, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    b[i] = a[i];
}

template <typename F>
double timeit(F f, int reps = 10)
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
  const size_t bytes = 1572864000; // the G tensor of 64^3 p4
  const size_t n = bytes / 16;
  double2 *a, *b;
  double* out;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 1, bytes));
  for (int blocks : {2048, 4096, 8192})
    for (int threads : {256, 512})
    {
      double ms = timeit([&] { read_linear<<<blocks, threads>>>(a, n, out); });
      printf("read_linear  blocks %5d threads %4d : %7.1f us  %6.0f GB/s\n", blocks, threads, ms * 1e3, bytes / ms / 1e6);
    }
  {
    double ms = timeit([&] { copy_linear<<<4096, 256>>>(a, b, n); });
    printf("copy_linear                          : %7.1f us  %6.0f GB/s (read+write)\n", ms * 1e3, 2 * bytes / ms / 1e6);
  }
  const size_t nchunks = bytes / 12000;
  for (int blocks : {512, 1024, 2048, 8192})
    for (int threads : {256, 512})
    {
      double ms1 = timeit([&] { read_waves<1><<<blocks, threads>>>(a, nchunks, out); });
      double ms2 = timeit([&] { read_waves<2><<<blocks, threads>>>(a, nchunks, out); });
      double ms5 = timeit([&] { read_waves<5><<<blocks, threads>>>(a, nchunks, out); });
      printf("read_waves   blocks %5d threads %4d : depth1 %6.0f GB/s  depth2 %6.0f GB/s  depth5 %6.0f GB/s\n", blocks,
             threads, bytes / ms1 / 1e6, bytes / ms2 / 1e6, bytes / ms5 / 1e6);
    }
  {
    const int npatch = (int)(bytes / 192000);
    double m;
    m = timeit([&] { read_patch<0, 1><<<npatch, 512>>>(a, out); });
    printf("read_patch   LDS  0 KB depth1 : %7.1f us %6.0f GB/s\n", m * 1e3, bytes / m / 1e6);
    m = timeit([&] { read_patch<53248, 1><<<npatch, 512>>>(a, out); });
    printf("read_patch   LDS 52 KB depth1 : %7.1f us %6.0f GB/s\n", m * 1e3, bytes / m / 1e6);
    m = timeit([&] { read_patch<53248, 2><<<npatch, 512>>>(a, out); });
    printf("read_patch   LDS 52 KB depth2 : %7.1f us %6.0f GB/s\n", m * 1e3, bytes / m / 1e6);
    m = timeit([&] { read_patch<53248, 5><<<npatch, 512>>>(a, out); });
    printf("read_patch   LDS 52 KB depth5 : %7.1f us %6.0f GB/s\n", m * 1e3, bytes / m / 1e6);
    m = timeit([&] { read_patch<32768, 1><<<npatch, 512>>>(a, out); });
    printf("read_patch   LDS 32 KB depth1 : %7.1f us %6.0f GB/s\n", m * 1e3, bytes / m / 1e6);
    // 8 launches of 1/8 of the patches, like the colours
    m = timeit([&] { for (int c = 0; c < 8; ++c) read_patch<53248, 1><<<npatch / 8, 512>>>(a + (size_t)c * (npatch / 8) * 12000, out); });
    printf("read_patch   LDS 52 KB depth1, 8 launches : %7.1f us %6.0f GB/s\n", m * 1e3, bytes / m / 1e6);
  }
  {
    double* y = (double*)b;
    for (int spin : {500, 1000, 2000})
    {
      double m0 = timeit([&] { store_tail<0><<<8192, 512>>>(y, spin); });
      double m6 = timeit([&] { store_tail<6><<<8192, 512>>>(y, spin); });
      printf("store_tail   spin %5.1f us : no stores %7.1f us, 6 stores/thread %7.1f us (+%5.2f us per workgroup round)\n",
             spin / 100.0, m0 * 1e3, m6 * 1e3, (m6 - m0) * 1e3 / 16.0);
    }
  }
  return 0;
}
