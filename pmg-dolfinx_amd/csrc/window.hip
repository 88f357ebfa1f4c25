// Halo windows: the neighbour exchange as direct stores into the neighbours' memory.
//
// The second way the library can move a halo between the GPUs of one node (the first is the
// grouped ncclSend / ncclRecv of comm.hip).  It replaces the same piece of the reference,
// dolfinx's Scatterer over GPU-aware MPI (src/vector.hpp:186-238), but needs no transport at all
// at run time: every rank owns a *window* (device memory exported with hipIpcGetMemHandle and
// mapped by its neighbours, who reach it over xGMI) and
//
//   begin : ONE kernel on the compute stream gathers the owned values and stores them straight
//           into the neighbours' windows, then raises an "arrived" flag in each neighbour's
//           flag block;
//   (the interior-cell kernels run on the compute stream meanwhile)
//   end   : ONE kernel waits for the neighbours' "arrived" flags, copies the window into the
//           ghost entries and raises a "consumed" flag at each neighbour.
//
// Two kernels per exchange, the same number as the pack / unpack pair of the RCCL route, and
// nothing else: no second stream, no events, no ~115 us of host work per grouped RCCL call
// (DESIGN.md section 6), no proxy thread, and the exchange is ordinary kernel launches to a
// hipGraph capture on any runtime.
//
// Protocol (per layout and direction d: 0 = owner -> ghost, 1 = ghost -> owner).  Exchanges are
// numbered 1, 2, ... on the device (a captured graph replays kernel arguments, so the number
// cannot be one); exchange s uses slot s & 1 of every window.
//   sender : waits until every neighbour has consumed exchange s - 2 (the previous tenant of the
//            slot), stores, fences at system scope, and the last block to finish writes s to
//            arrived[d][me] at every neighbour;
//   receiver: every block waits for arrived[d][k] >= s from all neighbours k, fences, reads its
//            own window; the last block to finish writes s to consumed[d][me] at every neighbour.
// A rank can therefore run at most two exchanges ahead of a neighbour, and no wait can starve:
// the waits are bounded (PMG_WINDOW_TIMEOUT_MS, default 5000) and a timeout is reported by the
// next call on the layout instead of hanging the GPU.
//
// The reductions still need a collective: a layout with windows keeps its communicator (or its
// callbacks) for those.
#include "common.hpp"

#include <cstdlib>
#include <cstring>

using namespace pmg;

namespace
{
constexpr int NBMAX = PMG_WINDOW_MAX_NEIGHBORS;
// flag block (uint64 words), written by the neighbours unless noted
constexpr int F_ARRIVED = 0;           // [2][NBMAX]
constexpr int F_CONSUMED = 2 * NBMAX;  // [2][NBMAX]
constexpr int F_LOCAL = 4 * NBMAX;     // written by the owner only:
constexpr int L_SENT = 0;              //   [2] number of the last exchange sent
constexpr int L_GOT = 2;               //   [2] number of the last exchange received
constexpr int L_PACK_DONE = 4;         //   [2] blocks of the running pack kernel that have finished
constexpr int L_UNPACK_DONE = 6;       //   [2] the same for the unpack kernel
static_assert(F_LOCAL + 8 == PMG_WINDOW_FLAG_WORDS, "flag block size");

struct WindowDev // device copy of what the kernels need
{
  int n = 0;                // neighbours
  double* win = nullptr;    // my window: [2 slots][stride]
  uint64_t* flags = nullptr;
  long long stride = 0, region[2] = {0, 0}; // start of the region direction d is received in
  double* nb_win[NBMAX];
  uint64_t* nb_flags[NBMAX];
  long long nb_stride[NBMAX], nb_off[2][NBMAX];
  int nb_slot[NBMAX];
  int* err = nullptr; // pinned host memory: 0, or PMG_WINDOW_ERR_* | direction << 8
  long long timeout_ticks = 0; // of wall_clock64 (100 MHz)
};

// Counters are polled with RELAXED system-scope loads (sc0 sc1: served by memory, past the caches).  An acquire would
// add a buffer_inv sc0 sc1 -- an invalidation of this XCD's L2 -- to every poll; it is not needed, because everything
// that is read after the counter (the window) is itself read with system-scope loads, and they are issued only after
// the counter's value has come back.
__device__ inline uint64_t load_sys(const uint64_t* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// a release-only fence at system scope (write the L2 back, wait): __threadfence_system() is acquire + release and
// would also invalidate this XCD's L2 each time
__device__ inline void release_sys() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
__device__ inline void store_sys(uint64_t* p, uint64_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline uint64_t load_dev(const uint64_t* p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait until *p >= target; false after the time limit (the caller records the error and goes on:
// every wave of the grid still reaches its end).
__device__ inline bool wait_flag(const uint64_t* p, uint64_t target, long long limit)
{
  if (load_sys(p) >= target)
    return true;
  const long long t0 = wall_clock64();
  while (load_sys(p) < target)
  {
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > limit)
      return false;
  }
  return true;
}

// Flags and the device-side sequence numbers are written with RELAXED stores behind ONE explicit release fence per block
// (round 4, read at the ISA level: every release STORE is a buffer_wbl2 of its own, and the acquire-release count-in
// adds a write-back and an invalidate -- five write-backs per put, three and an invalidate per get, 1.7 us each, in a
// kernel that moves a few kilobytes).
__device__ inline void store_sys_relaxed(uint64_t* p, uint64_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ inline void store_dev_relaxed(uint64_t* p, uint64_t v)
{
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifdef PMG_STAMPS // diagnostic build: readings of the 100 MHz clock by thread 0 of block 0 of the last window kernel
__device__ unsigned long long g_wstamp[16];
#define WSTAMP(i)                                                                                                     \
  do                                                                                                                  \
  {                                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0)                                                                          \
      g_wstamp[i] = __builtin_amdgcn_s_memrealtime();                                                                 \
  } while (0)
#else
#define WSTAMP(i)
#endif
constexpr int WBATCH = 8; // entries a thread moves per pass: their loads are all issued before the first store

// in[idx[i]] -> dst[slot][i], the entry's place in its neighbour's window (forward: the send list over the owned
// entries; reverse: the receive list over the ghost entries, `in` already offset)
__device__ __forceinline__ void window_put_body(const WindowDev* __restrict__ wp, int d, int n,
                                                const int32_t* __restrict__ idx, double* const* __restrict__ dst,
                                                const double* __restrict__ in)
{
  const WindowDev& w = *wp;
  __shared__ uint64_t s_seq;
  uint64_t* local = w.flags + F_LOCAL;
  if (threadIdx.x < 64) // first wave: lane k looks after neighbour k
  {
    const uint64_t s = load_dev(&local[L_SENT + d]) + 1;
    if ((int)threadIdx.x < w.n && s > 2)
      if (!wait_flag(&w.flags[F_CONSUMED + d * NBMAX + threadIdx.x], s - 2, w.timeout_ticks))
        *w.err = PMG_WINDOW_ERR_SLOT_BUSY | (d << 8);
    if (threadIdx.x == 0)
      s_seq = s;
  }
  __syncthreads();
  WSTAMP(1); // sequence number read, slot free
  const uint64_t s = s_seq;
  double* const* to = dst + (long long)(s & 1) * n;
  const int stride = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += WBATCH * stride)
  {
    double v[WBATCH];
    double* t[WBATCH];
#pragma unroll
    for (int k = 0; k < WBATCH; ++k)
    {
      const int i = i0 + k * stride;
      const int ic = i < n ? i : n - 1;
      t[k] = to[ic];
      v[k] = in[idx[ic]];
    }
#pragma unroll
    for (int k = 0; k < WBATCH; ++k)
      if (i0 + k * stride < n)
        *t[k] = v[k];
  }
  // One release per block, by the thread that counts the block in (a fence per thread writes the L2 back a thousand
  // times per exchange: measured 48 us per exchange against 20).  The barrier alone does not order the OTHER waves'
  // stores before it -- at workgroup scope the compiler waits for LDS only -- so every wave first waits for the
  // acknowledgement of its own stores.
  WSTAMP(2); // stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  WSTAMP(3); // stores acknowledged, all waves
  __shared__ int s_last;
  if (threadIdx.x == 0)
  {
    release_sys(); // this block's data: written back and acknowledged before anything below is issued
    WSTAMP(4); // release fence done
    if (gridDim.x == 1)
      s_last = 1;
    else
    {
      // (acquire-release at agent scope: the block that counts in last thereby acquires what the others released
      // before their counts, so its flag stores are ordered after every block's data by the memory model and not
      // only by issue order -- ADVICE r03)
      const unsigned long long prev = __hip_atomic_fetch_add((unsigned long long*)&local[L_PACK_DONE + d], 1ull,
                                                             __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      s_last = prev == (unsigned long long)gridDim.x - 1;
      if (s_last)
        release_sys(); // cumulativity: what it acquired, released to the neighbours with the flags
    }
  }
  __syncthreads();
  if (s_last && threadIdx.x < 64)
  {
    if ((int)threadIdx.x < w.n)
      store_sys_relaxed(&w.nb_flags[threadIdx.x][F_ARRIVED + d * NBMAX + w.nb_slot[threadIdx.x]], s);
    if (threadIdx.x == 0)
    {
      if (gridDim.x > 1)
        store_dev_relaxed(&local[L_PACK_DONE + d], 0ull);
      store_dev_relaxed(&local[L_SENT + d], s); // read by the next kernel of this stream only
    }
  }
  WSTAMP(5); // flags stored
}

__global__ void window_put_kernel(const WindowDev* __restrict__ wp, int d, int n,
                                  const int32_t* __restrict__ idx, double* const* __restrict__ dst,
                                  const double* __restrict__ in)
{
  window_put_body(wp, d, n, idx, dst, in);
}

// my window -> out[idx[i]] (assign: ghosts; add: owned entries)
// No fence at all on this side: the window is read with system-scope loads (served by memory), the values have come
// back before the stores that use them are issued, and those stores are for the next kernel of this stream; the
// "consumed" flag only tells the neighbour that the loads are done.
template <bool ADD>
__device__ __forceinline__ void window_get_body(const WindowDev* __restrict__ wp, int d, int n,
                                                const int32_t* __restrict__ idx, const int32_t* __restrict__ pos,
                                                double* __restrict__ out)
{
  const WindowDev& w = *wp;
  __shared__ uint64_t s_seq;
  uint64_t* local = w.flags + F_LOCAL;
  if (threadIdx.x < 64)
  {
    const uint64_t e = load_dev(&local[L_GOT + d]) + 1;
    if ((int)threadIdx.x < w.n)
      if (!wait_flag(&w.flags[F_ARRIVED + d * NBMAX + threadIdx.x], e, w.timeout_ticks))
        *w.err = PMG_WINDOW_ERR_NO_ARRIVAL | (d << 8);
    if (threadIdx.x == 0)
      s_seq = e;
  }
  __syncthreads(); // nothing of the window is read before the first wave has seen the flags
  WSTAMP(6); // arrival seen
  const uint64_t e = s_seq;
  const double* src = w.win + (long long)(e & 1) * w.stride + w.region[d];
  const int stride = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += WBATCH * stride)
  {
    double v[WBATCH];
    int32_t o[WBATCH], ps[WBATCH];
#pragma unroll
    for (int k = 0; k < WBATCH; ++k)
    {
      const int i = i0 + k * stride;
      const int ic = i < n ? i : n - 1;
      o[k] = idx[ic];
      ps[k] = pos[ic];
    }
#pragma unroll
    for (int k = 0; k < WBATCH; ++k) // (all addresses first: the loads below are then issued back to back)
      // system scope: the value a neighbour stored, not a line this XCD's L2 kept from two exchanges ago
      v[k] = __hip_atomic_load(&src[ps[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
    for (int k = 0; k < WBATCH; ++k)
      if (i0 + k * stride < n)
      {
        if (ADD)
          atomicAdd(&out[o[k]], v[k]);
        else
          out[o[k]] = v[k];
      }
  }
  __syncthreads(); // every wave's loads have returned (their values were stored)
  WSTAMP(7); // unpacked
  __shared__ int s_last;
  if (threadIdx.x == 0)
  {
    if (gridDim.x == 1)
      s_last = 1;
    else
    {
      const unsigned long long prev = __hip_atomic_fetch_add((unsigned long long*)&local[L_UNPACK_DONE + d], 1ull,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = prev == (unsigned long long)gridDim.x - 1;
    }
  }
  __syncthreads();
  if (s_last && threadIdx.x < 64)
  {
    if ((int)threadIdx.x < w.n)
      store_sys_relaxed(&w.nb_flags[threadIdx.x][F_CONSUMED + d * NBMAX + w.nb_slot[threadIdx.x]], e);
    if (threadIdx.x == 0)
    {
      if (gridDim.x > 1)
        store_dev_relaxed(&local[L_UNPACK_DONE + d], 0ull);
      store_dev_relaxed(&local[L_GOT + d], e);
    }
  }
}

template <bool ADD>
__global__ void window_get_kernel(const WindowDev* __restrict__ wp, int d, int n,
                                  const int32_t* __restrict__ idx, const int32_t* __restrict__ pos,
                                  double* __restrict__ out)
{
  window_get_body<ADD>(wp, d, n, idx, pos, out);
}

// The whole owner -> ghost exchange in ONE launch (round 4): every block puts its share, then waits for the
// neighbours' data and unpacks its share.  For levels so small that a kernel is shorter than the gap between two
// dependent launches (config 3's per-GPU share: 32^3 cells), where putting before and getting behind the interior
// cells' launch hides nothing and costs a launch.  All blocks of the grid are resident together (at most 64),
// so a block that polls never keeps a block that still has to put from running -- on this rank; between ranks every
// rank puts before it polls.
constexpr int WHOLE_THREADS = 1024;
__global__ void __launch_bounds__(WHOLE_THREADS)
    window_exchange_kernel(const WindowDev* __restrict__ wp, int n_send,
                                       const int32_t* __restrict__ send_idx, double* const* __restrict__ dst,
                                       const double* __restrict__ in, int n_recv,
                                       const int32_t* __restrict__ recv_idx, const int32_t* __restrict__ recv_pos,
                                       double* __restrict__ ghosts)
{
  WSTAMP(0); // entry
  window_put_body(wp, 0, n_send, send_idx, dst, in);
  __syncthreads();
  window_get_body<false>(wp, 0, n_recv, recv_idx, recv_pos, ghosts);
  WSTAMP(8); // end
}
#ifdef PMG_STAMPS
extern "C" int pmg_debug_read_window_stamps(unsigned long long* out)
{
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamp), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif

constexpr size_t ALIGN = 32; // doubles: every neighbour's segment starts on a 256-byte boundary

// padded segment starts of one side of the plan; returns the padded length
size_t segment_offsets(const int32_t* counts, int n, std::vector<size_t>& off)
{
  off.assign((size_t)n, 0);
  size_t at = 0;
  for (int k = 0; k < n; ++k)
  {
    off[(size_t)k] = at;
    at = (at + (size_t)counts[k] + ALIGN - 1) / ALIGN * ALIGN;
  }
  return at;
}

// few, fat blocks: every block pays one system-scope fence and one look at the flags
constexpr int PUT_THREADS = 256;
int put_blocks(int n)
{
  static const int per_thread = [] { // tuning: PMG_WINDOW_ENTRIES_PER_THREAD
    const char* e = std::getenv("PMG_WINDOW_ENTRIES_PER_THREAD");
    return e ? std::max(1, std::atoi(e)) : 8;
  }();
  long long b = ((long long)n + per_thread * PUT_THREADS - 1) / (per_thread * PUT_THREADS);
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}
} // namespace

struct pmg_window_s
{
  WindowDev host;           // what the device copy holds
  WindowDev* dev = nullptr; // device copy
  int* err = nullptr;       // pinned
  // where every list entry goes in its neighbour's window, per put direction: [2 slots][n]
  double **fwd_dst = nullptr, **rev_dst = nullptr;
  // places of the list entries in my own window regions
  int32_t *recv_pos = nullptr, *send_pos = nullptr;
};

namespace pmg
{
void window_destroy(pmg_layout l)
{
  pmg_window_s* w = l->win;
  if (!w)
    return;
  (void)hipFree(w->dev);
  (void)hipHostFree(w->err);
  for (void* p : {(void*)w->fwd_dst, (void*)w->rev_dst, (void*)w->recv_pos, (void*)w->send_pos})
    (void)hipFree(p);
  delete w;
  l->win = nullptr;
}

static int window_check(pmg_layout l)
{
  const int e = *(volatile int*)l->win->err;
  if (e == 0)
    return PMG_OK;
  const int d = e >> 8;
  return fail(PMG_ERR_HIP,
              (e & 0xff) == PMG_WINDOW_ERR_NO_ARRIVAL
                  ? "halo window: a neighbour's data (%s) did not arrive within the time limit (PMG_WINDOW_TIMEOUT_MS)"
                  : "halo window: a neighbour did not consume the previous exchange (%s) within the time limit "
                    "(PMG_WINDOW_TIMEOUT_MS)",
              d ? "ghost -> owner" : "owner -> ghost");
}

int window_exchange_begin(pmg_layout l, bool reverse, const double* x, hipStream_t s)
{
  pmg_window_s* w = l->win;
  PMG_TRY(window_check(l));
  if (w->host.n == 0)
    return PMG_OK;
  const int d = reverse ? 1 : 0;
  const int n = reverse ? l->n_recv : l->n_send;
  window_put_kernel<<<put_blocks(n), PUT_THREADS, 0, s>>>(w->dev, d, n, reverse ? l->recv_idx : l->send_idx,
                                                          reverse ? w->rev_dst : w->fwd_dst,
                                                          reverse ? x + l->size_local : x);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

// owner -> ghost, begin and end in one launch (window_exchange_kernel)
int window_exchange_whole(pmg_layout l, double* x, hipStream_t s)
{
  pmg_window_s* w = l->win;
  PMG_TRY(window_check(l));
  if (w->host.n == 0)
    return PMG_OK;
  const long long n = std::max(l->n_send, l->n_recv);
  // Entries per block of 1 024 threads (tuning: PMG_WINDOW_ENTRIES_PER_BLOCK).  In-kernel clock readings
  // (-DPMG_STAMPS, tools/stamp_exchange.py) show where the kernel's time is: not in its fences (0.4 us) or flags
  // (0.4 - 1 us each way) but in the passes of its two copy loops, 1.6 - 3.3 us each (list -> gather -> store, all
  // dependent).  One pass of four entries per thread, paid for with a count-in once there are several blocks:
  // 16 384 / 8 192 / 4 096 / 2 048 entries per block = 13.5 / 11.3 / 9.7 / 9.6 us for the 12 675 entries of a degree-2
  // level, 18.6 / 12.2 / 10.6 / 10.9 us for the 83 205 of a degree-4 level, 1.018 / 0.965 / 0.948 / 0.944 ms per cycle
  // of a 32^3 share.
  static const int per_block = [] {
    const char* e = std::getenv("PMG_WINDOW_ENTRIES_PER_BLOCK");
    return e ? std::max(256, std::atoi(e)) : 4096;
  }();
  const int blocks = (int)std::min<long long>(64, std::max<long long>(1, (n + per_block - 1) / per_block));
  window_exchange_kernel<<<blocks, WHOLE_THREADS, 0, s>>>(w->dev, l->n_send, l->send_idx, w->fwd_dst, x, l->n_recv,
                                                        l->recv_idx, w->recv_pos, x + l->size_local);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

int window_exchange_end(pmg_layout l, bool reverse, double* x, hipStream_t s)
{
  pmg_window_s* w = l->win;
  if (w->host.n == 0)
    return PMG_OK;
  if (reverse)
    window_get_kernel<true><<<put_blocks(l->n_send), PUT_THREADS, 0, s>>>(w->dev, 1, l->n_send, l->send_idx,
                                                                          w->send_pos, x);
  else
    window_get_kernel<false><<<put_blocks(l->n_recv), PUT_THREADS, 0, s>>>(w->dev, 0, l->n_recv, l->recv_idx,
                                                                           w->recv_pos, x + l->size_local);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}
} // namespace pmg

// ---- window memory ----
static int g_window_fine_grained = -1; // what the last pmg_window_alloc of this process obtained
extern "C" int pmg_window_fine_grained(void) { return g_window_fine_grained; }
static_assert(sizeof(hipIpcMemHandle_t) == PMG_WINDOW_HANDLE_BYTES, "pmg window handle size");

// The protocol (relaxed polling, release-only fences, no L2 invalidation per poll) needs memory in which a peer GPU's
// stores become visible to a kernel that is already running here: fine-grained device memory.  Ordinary
// (coarse-grained) memory is NOT a silent fall-back: between two GPUs the reader can be served stale lines
// from its L2 -- wrong halo values, not a timeout.  It is accepted only on request, PMG_WINDOW_ALLOW_COARSE=1, for runs
// in which all ranks share ONE device (a rehearsal; one L2 path) on a runtime that cannot allocate or export
// fine-grained memory.
static bool window_coarse_allowed()
{
  const char* e = std::getenv("PMG_WINDOW_ALLOW_COARSE");
  return e && e[0] == '1';
}

extern "C" int pmg_window_alloc(size_t bytes, void** ptr, char* handle)
{
  PMG_REQUIRE(ptr && handle && bytes > 0, "pmg_window_alloc: bad argument");
  void* p = nullptr;
  hipIpcMemHandle_t h;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
  const char* what = "hipExtMallocWithFlags(hipDeviceMallocFinegrained)";
  if (e == hipSuccess)
  {
    what = "hipMemset";
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess)
      e = hipDeviceSynchronize();
    if (e == hipSuccess)
    {
      what = "hipIpcGetMemHandle of fine-grained memory";
      e = hipIpcGetMemHandle(&h, p);
    }
  }
  g_window_fine_grained = e == hipSuccess ? 1 : 0;
  if (e != hipSuccess)
  {
    (void)hipGetLastError();
    (void)hipFree(p);
    p = nullptr;
    if (!window_coarse_allowed())
      return fail(PMG_ERR_HIP,
                  "pmg_window_alloc: %s failed (%s): no fine-grained device memory for the window.  Ordinary memory is "
                  "not a safe substitute between GPUs (stale reads); set PMG_WINDOW_ALLOW_COARSE=1 only when all ranks "
                  "share one device", what, hipGetErrorString(e));
    PMG_HIP(hipMalloc(&p, bytes));
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess)
      e = hipDeviceSynchronize();
    if (e == hipSuccess)
      e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess)
    {
      (void)hipFree(p);
      return fail(PMG_ERR_HIP, "pmg_window_alloc: %s", hipGetErrorString(e));
    }
  }
  std::memcpy(handle, &h, sizeof(h));
  *ptr = p;
  return PMG_OK;
}

extern "C" int pmg_window_open(const char* handle, void** ptr)
{
  PMG_REQUIRE(ptr && handle, "pmg_window_open: NULL argument");
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  PMG_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
  return PMG_OK;
}

extern "C" int pmg_window_close(void* ptr)
{
  if (ptr)
    PMG_HIP(hipIpcCloseMemHandle(ptr));
  return PMG_OK;
}

extern "C" int pmg_window_free(void* ptr)
{
  if (ptr)
    PMG_HIP(hipFree(ptr));
  return PMG_OK;
}

extern "C" int pmg_layout_window_describe(int32_t n_neighbors, const int32_t* send_counts,
                                          const int32_t* recv_counts, int64_t* window_doubles,
                                          int64_t* fwd_offsets, int64_t* rev_offsets)
{
  PMG_REQUIRE(n_neighbors >= 0 && n_neighbors <= NBMAX,
              "halo windows serve at most %d neighbours per rank (this one has %d): use the RCCL exchange", NBMAX,
              n_neighbors);
  PMG_REQUIRE(window_doubles && (n_neighbors == 0 || (send_counts && recv_counts && fwd_offsets && rev_offsets)),
              "pmg_layout_window_describe: NULL argument");
  std::vector<size_t> so, ro;
  const size_t rlen = segment_offsets(recv_counts, n_neighbors, ro);
  const size_t slen = segment_offsets(send_counts, n_neighbors, so);
  for (int k = 0; k < n_neighbors; ++k)
  {
    fwd_offsets[k] = (int64_t)ro[(size_t)k];          // where neighbour k's owned values land (my ghosts)
    rev_offsets[k] = (int64_t)(rlen + so[(size_t)k]); // where neighbour k's ghost values land (my owned entries)
  }
  const size_t stride = rlen + slen;
  *window_doubles = (int64_t)(2 * (stride ? stride : ALIGN));
  return PMG_OK;
}

extern "C" int pmg_layout_set_windows(pmg_layout l, int32_t n_neighbors, const int32_t* send_counts,
                                      const int32_t* recv_counts, double* window, uint64_t* flags,
                                      double* const* nb_window, uint64_t* const* nb_flags,
                                      const int64_t* nb_window_doubles, const int64_t* nb_fwd_offset,
                                      const int64_t* nb_rev_offset, const int32_t* nb_slot)
{
  PMG_REQUIRE(l && window && flags, "pmg_layout_set_windows: NULL argument");
  PMG_REQUIRE(n_neighbors >= 0 && n_neighbors <= NBMAX,
              "halo windows serve at most %d neighbours per rank (this one has %d): use the RCCL exchange", NBMAX,
              n_neighbors);
  PMG_REQUIRE(n_neighbors == 0 || (send_counts && recv_counts && nb_window && nb_flags && nb_window_doubles
                                   && nb_fwd_offset && nb_rev_offset && nb_slot),
              "pmg_layout_set_windows: neighbour arrays missing");
  long long ns = 0, nr = 0;
  for (int k = 0; k < n_neighbors; ++k)
  {
    PMG_REQUIRE(send_counts[k] >= 0 && recv_counts[k] >= 0, "pmg_layout_set_windows: negative count");
    PMG_REQUIRE(nb_window[k] && nb_flags[k], "pmg_layout_set_windows: neighbour %d has no window", k);
    PMG_REQUIRE(nb_slot[k] >= 0 && nb_slot[k] < NBMAX, "pmg_layout_set_windows: bad slot %d", nb_slot[k]);
    // what I store must fit the neighbour's window: it receives my send list in its forward region ...
    PMG_REQUIRE(nb_fwd_offset[k] >= 0 && nb_rev_offset[k] >= 0
                    && 2 * (nb_fwd_offset[k] + send_counts[k]) <= nb_window_doubles[k]
                    && 2 * (nb_rev_offset[k] + recv_counts[k]) <= nb_window_doubles[k],
                "pmg_layout_set_windows: my segment does not fit neighbour %d's window", k);
    ns += send_counts[k];
    nr += recv_counts[k];
  }
  PMG_REQUIRE(ns == l->n_send && nr == l->n_recv,
              "pmg_layout_set_windows: per-neighbour counts (%lld, %lld) do not add up to the layout's n_send, "
              "n_recv (%d, %d)", ns, nr, l->n_send, l->n_recv);
  window_destroy(l);
  auto* w = new pmg_window_s;
  l->win = w;
  const int rc = [&]() -> int { // (a failure below must not leave a half-built attachment on the layout)
  std::vector<size_t> so, ro;
  const size_t rlen = segment_offsets(recv_counts, n_neighbors, ro);
  const size_t slen = segment_offsets(send_counts, n_neighbors, so);
  WindowDev& h = w->host;
  h.n = n_neighbors;
  h.win = window;
  h.flags = flags;
  h.stride = (long long)(rlen + slen ? rlen + slen : ALIGN);
  h.region[0] = 0;
  h.region[1] = (long long)rlen;
  for (int k = 0; k < n_neighbors; ++k)
  {
    h.nb_win[k] = nb_window[k];
    h.nb_flags[k] = nb_flags[k];
    h.nb_stride[k] = nb_window_doubles[k] / 2;
    h.nb_off[0][k] = nb_fwd_offset[k];
    h.nb_off[1][k] = nb_rev_offset[k];
    h.nb_slot[k] = nb_slot[k];
  }
  long long ms = 5000;
  if (const char* e = std::getenv("PMG_WINDOW_TIMEOUT_MS"))
    ms = std::atoll(e) > 0 ? std::atoll(e) : ms;
  h.timeout_ticks = ms * 100000ll; // wall_clock64 counts at 100 MHz
  PMG_HIP(hipHostMalloc(&w->err, sizeof(int), hipHostMallocMapped));
  *w->err = 0;
  h.err = w->err;
  PMG_HIP(hipMalloc(&w->dev, sizeof(WindowDev)));
  PMG_HIP(hipMemcpy(w->dev, &h, sizeof(WindowDev), hipMemcpyHostToDevice));
  // per-entry tables
  auto upload = [](const std::vector<int32_t>& v, int32_t** d) -> int {
    PMG_HIP(hipMalloc(d, sizeof(int32_t) * (v.empty() ? 1 : v.size())));
    if (!v.empty())
      PMG_HIP(hipMemcpy(*d, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice));
    return PMG_OK;
  };
  std::vector<int32_t> pos;
  std::vector<double*> dst;
  auto side = [&](const int32_t* counts, const std::vector<size_t>& off, int d) {
    pos.clear();
    size_t total = 0;
    for (int k = 0; k < n_neighbors; ++k)
      total += (size_t)counts[k];
    dst.assign(2 * total, nullptr);
    size_t at = 0;
    for (int k = 0; k < n_neighbors; ++k)
      for (int32_t i = 0; i < counts[k]; ++i, ++at)
      {
        pos.push_back((int32_t)(off[(size_t)k] + (size_t)i));
        for (int slot = 0; slot < 2; ++slot)
          dst[(size_t)slot * total + at] = h.nb_win[k] + slot * h.nb_stride[k] + h.nb_off[d][k] + i;
      }
  };
  auto upload_dst = [&](double*** d) -> int {
    PMG_HIP(hipMalloc(d, sizeof(double*) * (dst.empty() ? 1 : dst.size())));
    if (!dst.empty())
      PMG_HIP(hipMemcpy(*d, dst.data(), sizeof(double*) * dst.size(), hipMemcpyHostToDevice));
    return PMG_OK;
  };
  side(send_counts, so, 0); // the send list is put forward and received into in reverse (region[1] + send_pos)
  PMG_TRY(upload_dst(&w->fwd_dst));
  PMG_TRY(upload(pos, &w->send_pos));
  side(recv_counts, ro, 1);
  PMG_TRY(upload_dst(&w->rev_dst));
  PMG_TRY(upload(pos, &w->recv_pos));
  return PMG_OK;
  }();
  if (rc != PMG_OK)
    window_destroy(l);
  return rc;
}

// ==================================================================================================
// A communicator made of windows: the reductions (and the set-up gathers) without a transport library.
//
// Every rank owns one window of PMG_COMM_WINDOW_BYTES which all ranks map; an exchange of up to
// PMG_COMM_WINDOW_CHUNK doubles per rank is two kernels on the caller's stream, as for the halo:
//   put : stores my values into slot [number & 1][my rank] of EVERY rank's window (my own included), one system-scope
//         fence per workgroup, and the last workgroup raises arrived[slot][my rank] = number at every rank;
//   get : waits for arrived[slot][r] >= number from all ranks r, then combines the R contributions in rank order
//         (sum or maximum: the same bits on every rank) or lays them side by side (gather).
// The exchange number lives on the device, so a captured all-reduce replays correctly.  Two slots suffice without
// acknowledgements: a rank writes exchange n + 2 only after its get of n + 1 returned, i.e. after every rank had put
// n + 1, which each of them does only after its own get of n -- provided a rank issues its puts and gets in one order
// (one stream at a time), which is also what makes the ranks agree on the numbering.
// Longer vectors (the replicated coarse solve sums whole level vectors) go chunk by chunk.
namespace
{
constexpr int WC_MAXR = PMG_COMM_WINDOW_MAX_RANKS;
constexpr int WC_CHUNK = PMG_COMM_WINDOW_CHUNK; // doubles per rank and exchange
struct CommWindow              // the layout of a rank's window
{
  uint64_t arrived[2][WC_MAXR];
  uint64_t sent, got, put_done, get_done; // owner only
  uint64_t pad[4];
  double values[2][WC_MAXR][WC_CHUNK];
};
static_assert(sizeof(CommWindow) == PMG_COMM_WINDOW_BYTES, "communicator window size");

struct WCommDev
{
  int rank = 0, nranks = 1;
  CommWindow* win[WC_MAXR]; // win[rank] is mine
  int* err = nullptr;
  long long timeout_ticks = 0;
};

__global__ void wcomm_put_kernel(const WCommDev* __restrict__ wp, int m, const double* __restrict__ src)
{
  const WCommDev& w = *wp;
  CommWindow* mine = w.win[w.rank];
  const uint64_t n = load_dev(&mine->sent) + 1;
  const int slot = (int)(n & 1);
  const long long total = (long long)w.nranks * m;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
  {
    const int r = (int)(i / m), j = (int)(i - (long long)r * m);
    w.win[r]->values[slot][w.rank][j] = src[j];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every wave's stores acknowledged before the block is counted in
  __syncthreads();
  __shared__ int s_last;
  if (threadIdx.x == 0)
  {
    release_sys();
    const unsigned long long prev = __hip_atomic_fetch_add((unsigned long long*)&mine->put_done, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == (unsigned long long)gridDim.x - 1;
  }
  __syncthreads();
  if (s_last && threadIdx.x < 64)
  {
    release_sys();
    if ((int)threadIdx.x < w.nranks)
      store_sys(&w.win[threadIdx.x]->arrived[slot][w.rank], n);
    if (threadIdx.x == 0)
    {
      __hip_atomic_store(&mine->put_done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&mine->sent, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// MODE 0: sum, 1: maximum (dst[j] over the ranks, in rank order), 2: gather (dst[r * m + j])
template <int MODE>
__global__ void wcomm_get_kernel(const WCommDev* __restrict__ wp, int m, double* __restrict__ dst)
{
  const WCommDev& w = *wp;
  CommWindow* mine = w.win[w.rank];
  __shared__ uint64_t s_n;
  if (threadIdx.x < 64)
  {
    const uint64_t n = load_dev(&mine->got) + 1;
    if ((int)threadIdx.x < w.nranks)
      if (!wait_flag(&mine->arrived[n & 1][threadIdx.x], n, w.timeout_ticks))
        *w.err = PMG_WINDOW_ERR_NO_ARRIVAL;
    if (threadIdx.x == 0)
      s_n = n;
  }
  __syncthreads();
  const uint64_t n = s_n;
  const int slot = (int)(n & 1);
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x)
  {
    double acc = 0.0;
    for (int r = 0; r < w.nranks; ++r)
    {
      const double v = __hip_atomic_load(&mine->values[slot][r][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (MODE == 2)
        dst[(size_t)r * m + j] = v;
      else if (MODE == 1)
        acc = r == 0 ? v : fmax(acc, v);
      else
        acc = r == 0 ? v : acc + v;
    }
    if (MODE != 2)
      dst[j] = acc;
  }
  __syncthreads();
  __shared__ int s_last;
  if (threadIdx.x == 0)
  {
    const unsigned long long prev = __hip_atomic_fetch_add((unsigned long long*)&mine->get_done, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == (unsigned long long)gridDim.x - 1;
    if (s_last)
    {
      __hip_atomic_store(&mine->get_done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&mine->got, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int wc_blocks(long long items)
{
  const long long b = (items + 4 * 256 - 1) / (4 * 256);
  return (int)(b < 1 ? 1 : (b > 256 ? 256 : b));
}
} // namespace

struct pmg_window_comm_s
{
  WCommDev host;
  WCommDev* dev = nullptr;
  int* err = nullptr;          // pinned
  double* stage = nullptr;     // device: the gather's staging, WC_CHUNK * (1 + WC_MAXR) doubles
};

namespace pmg
{
void wcomm_destroy(pmg_comm c)
{
  pmg_window_comm_s* w = c->wcomm;
  if (!w)
    return;
  (void)hipFree(w->dev);
  (void)hipFree(w->stage);
  (void)hipHostFree(w->err);
  delete w;
  c->wcomm = nullptr;
}

static int wcomm_check(pmg_window_comm_s* w)
{
  if (*(volatile int*)w->err == 0)
    return PMG_OK;
  return fail(PMG_ERR_HIP, "communicator windows: a rank's contribution did not arrive within the time limit "
                           "(PMG_WINDOW_TIMEOUT_MS)");
}

int wcomm_allreduce(pmg_comm c, double* d_values, int n, bool max, hipStream_t s)
{
  pmg_window_comm_s* w = c->wcomm;
  PMG_TRY(wcomm_check(w));
  for (int o = 0; o < n; o += WC_CHUNK)
  {
    const int m = n - o < WC_CHUNK ? n - o : WC_CHUNK;
    wcomm_put_kernel<<<wc_blocks((long long)c->nranks * m), 256, 0, s>>>(w->dev, m, d_values + o);
    if (max)
      wcomm_get_kernel<1><<<wc_blocks(m), 256, 0, s>>>(w->dev, m, d_values + o);
    else
      wcomm_get_kernel<0><<<wc_blocks(m), 256, 0, s>>>(w->dev, m, d_values + o);
  }
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

// blocking, set-up only: on the null stream, so it is ordered against everything the caller has in flight
int wcomm_allgather(pmg_comm c, const void* send, size_t bytes, void* recv)
{
  pmg_window_comm_s* w = c->wcomm;
  PMG_TRY(wcomm_check(w));
  const size_t chunk_bytes = sizeof(double) * WC_CHUNK;
  std::vector<double> host((size_t)WC_CHUNK * (size_t)c->nranks);
  // A set-up gather runs on the null stream.  That does NOT order it against work still queued on streams created
  // with hipStreamNonBlocking (torch's side streams, a caller's own) -- and the exchange number lives in device memory,
  // read when each kernel runs: a gather that overtook an all-reduce of this communicator still queued on such a
  // stream would number its exchange differently on different ranks.  So the device is drained first;
  // this is set-up code (window handles, unique ids), never on the hot path.
  PMG_HIP(hipDeviceSynchronize());
  for (size_t o = 0; o < bytes; o += chunk_bytes)
  {
    const size_t b = bytes - o < chunk_bytes ? bytes - o : chunk_bytes;
    const int m = (int)((b + 7) / 8);
    std::vector<double> in((size_t)m, 0.0);
    std::memcpy(in.data(), static_cast<const char*>(send) + o, b);
    PMG_HIP(hipMemcpy(w->stage, in.data(), sizeof(double) * m, hipMemcpyHostToDevice));
    wcomm_put_kernel<<<wc_blocks((long long)c->nranks * m), 256, 0, nullptr>>>(w->dev, m, w->stage);
    wcomm_get_kernel<2><<<wc_blocks(m), 256, 0, nullptr>>>(w->dev, m, w->stage + WC_CHUNK);
    PMG_HIP(hipGetLastError());
    PMG_HIP(hipMemcpy(host.data(), w->stage + WC_CHUNK, sizeof(double) * m * (size_t)c->nranks, hipMemcpyDeviceToHost));
    PMG_TRY(wcomm_check(w));
    for (int r = 0; r < c->nranks; ++r)
      std::memcpy(static_cast<char*>(recv) + (size_t)r * bytes + o, host.data() + (size_t)r * m, b);
  }
  return PMG_OK;
}
} // namespace pmg

extern "C" int pmg_comm_create_windows(pmg_comm* out, int rank, int nranks, void* const* windows)
{
  PMG_REQUIRE(out && windows && nranks >= 1 && nranks <= WC_MAXR && rank >= 0 && rank < nranks,
              "pmg_comm_create_windows: bad argument (at most %d ranks)", WC_MAXR);
  for (int r = 0; r < nranks; ++r)
    PMG_REQUIRE(windows[r], "pmg_comm_create_windows: rank %d has no window", r);
  auto* c = new pmg_comm_s;
  HandleGuard<pmg_comm> guard(c, pmg_comm_destroy);
  c->rank = rank;
  c->nranks = nranks;
  auto* w = new pmg_window_comm_s;
  c->wcomm = w;
  w->host.rank = rank;
  w->host.nranks = nranks;
  for (int r = 0; r < nranks; ++r)
    w->host.win[r] = static_cast<CommWindow*>(windows[r]);
  long long ms = 5000;
  if (const char* e = std::getenv("PMG_WINDOW_TIMEOUT_MS"))
    ms = std::atoll(e) > 0 ? std::atoll(e) : ms;
  w->host.timeout_ticks = ms * 100000ll;
  PMG_HIP(hipHostMalloc(&w->err, sizeof(int), hipHostMallocMapped));
  *w->err = 0;
  w->host.err = w->err;
  PMG_HIP(hipMalloc(&w->dev, sizeof(WCommDev)));
  PMG_HIP(hipMemcpy(w->dev, &w->host, sizeof(WCommDev), hipMemcpyHostToDevice));
  PMG_HIP(hipMalloc(&w->stage, sizeof(double) * WC_CHUNK * (1 + (size_t)WC_MAXR)));
  *out = guard.release();
  return PMG_OK;
}
