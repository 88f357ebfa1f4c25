// Shared definitions of the pmg_amd library (gfx950 only).
#pragma once

#include "../../include/pmg_amd.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace pmg
{
extern thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...);

#define PMG_HIP(call)                                                                              \
  do                                                                                               \
  {                                                                                                \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return pmg::fail(PMG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),         \
                       __FILE__, __LINE__);                                                        \
  } while (0)

#define PMG_TRY(call)                                                                              \
  do                                                                                               \
  {                                                                                                \
    int rc_ = (call);                                                                              \
    if (rc_ != PMG_OK)                                                                             \
      return rc_;                                                                                  \
  } while (0)

#define PMG_REQUIRE(cond, ...)                                                                     \
  do                                                                                               \
  {                                                                                                \
    if (!(cond))                                                                                   \
      return pmg::fail(PMG_ERR_INVALID, __VA_ARGS__);                                              \
  } while (0)

inline hipStream_t S(pmg_stream s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int MAXND = PMG_MAX_DEGREE + 1;

// 1-D tables on the host (tables.cpp)
void gll_table(int n, double* x, double* w);
void lagrange_derivative_table(int n, const double* x, double* D);
void lagrange_eval_table(int nc, const double* xc, int nf, const double* xf, double* M);

// Cell-local node order (pmg_amd.h "cell-local node order"; tables.hip).  perm1d[j] = ascending position of the
// caller's 1-D node j; node_permutation validates / builds it, cell_permutation expands it to the nd^3 cell-local
// numbers (perm3[t_caller] = t_ascending, t = ja*nd^2 + jb*nd + jc).
int node_permutation(int node_order, int degree, const int32_t* custom, std::vector<int32_t>& perm1d);
std::vector<int32_t> cell_permutation(int nd, const std::vector<int32_t>& perm1d);
inline bool is_identity(const std::vector<int32_t>& p)
{
  for (size_t i = 0; i < p.size(); ++i)
    if (p[i] != (int32_t)i)
      return false;
  return true;
}
// out[row * n + perm[t]] (x width) = in[row * n + t] (x width) on the device; perm is a device array of n entries
int permute_rows_i32(long long nrows, int n, const int32_t* perm_d, const int32_t* in, int32_t* out, hipStream_t s);
int permute_rows_f64(long long nrows, int n, int width, const int32_t* perm_d, const double* in, double* out,
                     hipStream_t s);

// Kernel-argument copy of a small dense table (<= 9x9), passed by value.
struct Table
{
  double v[MAXND * MAXND];
};
} // namespace pmg

// Destroys a half-built handle when a constructor returns early through
// PMG_REQUIRE / PMG_TRY / PMG_HIP; release() hands the finished handle over.
template <typename H>
struct HandleGuard
{
  H h;
  int (*destroy)(H);
  HandleGuard(H handle, int (*d)(H)) : h(handle), destroy(d) {}
  HandleGuard(const HandleGuard&) = delete;
  HandleGuard& operator=(const HandleGuard&) = delete;
  ~HandleGuard()
  {
    if (h)
      destroy(h);
  }
  H release()
  {
    H t = h;
    h = nullptr;
    return t;
  }
};

struct ncclComm;
struct pmg_window_s;
struct pmg_window_comm_s;

// One process's communicator: RCCL (comm.hip) or, without any transport library, a window of device memory that
// every rank has mapped (window.hip: reductions and set-up gathers as direct stores + flags; the halo of its layouts
// then has to be halo windows as well).
struct pmg_comm_s
{
  ncclComm* comm = nullptr;
  pmg_window_comm_s* wcomm = nullptr;
  int rank = 0, nranks = 1;
  hipStream_t stream = nullptr; // every RCCL call of this communicator is issued here, in program order
  hipEvent_t ev_in = nullptr, ev_out = nullptr; // reductions: compute stream -> comm stream -> compute stream
  bool reduced_eagerly = false; // an all-reduce has been issued outside a capture (see comm_capture_ready)
};

struct pmg_layout_s
{
  int32_t size_local = 0, num_ghosts = 0, n_send = 0, n_recv = 0;
  // native communicator (comm.hip): when set, the exchange and the reductions are issued by the
  // library on RCCL and the callbacks below are not used
  pmg_comm_s* comm = nullptr;
  std::vector<int32_t> nb_rank, nb_send, nb_recv; // neighbour ranks and per-neighbour counts
  hipEvent_t ev_packed = nullptr, ev_arrived = nullptr;
  // staging of the native exchange (owned): every neighbour's segment starts on a 256-byte boundary -- RCCL moves a
  // 7-segment halo 22 % faster than with segments that are only 8-byte aligned (tools/time_halo_alignment.py);
  // *_pos[i] = place of list entry i in the padded buffer, *_off[k] = start of neighbour k's segment
  double *c_send = nullptr, *c_recv = nullptr;
  int32_t *send_pos = nullptr, *recv_pos = nullptr;
  std::vector<size_t> send_off, recv_off;
  // halo windows (window.hip): when set, the neighbour exchange is direct stores into the neighbours' windows;
  // the reductions still go through `comm` or the callbacks
  pmg_window_s* win = nullptr;
  bool exchange_inline = false; // the exchange in flight was issued on the compute stream (graph capture)
  bool exchanged_eagerly = false; // an exchange of this layout has been issued outside a capture (comm_capture_ready)
  const int32_t* send_idx = nullptr;
  const int32_t* recv_idx = nullptr;
  double* send_buf = nullptr;
  double* recv_buf = nullptr;
  pmg_exchange_fn exchange = nullptr;
  pmg_allreduce_fn allreduce = nullptr;
  pmg_allreduce_fn allreduce_max = nullptr;
  void* user = nullptr;
  // reduction scratch (owned)
  double* d_partials = nullptr; // [RED_BLOCKS] block partials + [RED_SLOTS] results
  double* h_result = nullptr;   // pinned, [RED_SLOTS]
  long long fwd_scatters = 0; // forward scatters issued (pmg_layout_forward_scatters: exchange bookkeeping tests)
  bool multi_rank() const { return comm != nullptr || allreduce != nullptr; }
  int32_t total() const { return size_local + num_ghosts; }
};

namespace pmg
{
constexpr int RED_BLOCKS = 1024;
constexpr int RED_THREADS = 256;
constexpr int RED_SLOTS = 8;

// comm.hip -- native (RCCL) exchange and reductions of a layout that has a communicator
int comm_exchange_begin(pmg_layout l, bool reverse, hipStream_t s);
int comm_exchange_end(pmg_layout l, hipStream_t s);
int comm_allreduce(pmg_layout l, double* d_values, int n, bool max, hipStream_t s);
int comm_allreduce(pmg_comm c, double* d_values, int n, bool max, hipStream_t s);
bool comm_capture_ready(pmg_layout l, bool with_allreduce);

// window.hip -- the exchange of a layout that has halo windows (x: the whole vector, owned entries first)
int window_exchange_begin(pmg_layout l, bool reverse, const double* x, hipStream_t s);
int window_exchange_end(pmg_layout l, bool reverse, double* x, hipStream_t s);
int window_exchange_whole(pmg_layout l, double* x, hipStream_t s);
// owner -> ghost exchange of x, complete on return of the stream: one launch on a window layout, begin + end otherwise
int scatter_fwd_whole(pmg_layout l, double* x, hipStream_t s);
// is the exchange of this layout one that a small level should take whole, in front of ONE launch over all its cells,
// rather than split around the interior cells' launch?  (halo windows; PMG_FUSED_EXCHANGE=0 in the environment: never)
bool layout_exchanges_whole(pmg_layout l);
void window_destroy(pmg_layout l);
// ... and the reductions / set-up gathers of a communicator made of windows
int wcomm_allreduce(pmg_comm c, double* d_values, int n, bool max, hipStream_t s);
int wcomm_allgather(pmg_comm c, const void* send, size_t bytes, void* recv);
void wcomm_destroy(pmg_comm c);

// Profiling ranges (roctx, bound at run time; no-ops when libroctx64 is absent).  The reference
// annotates each CG iteration (src/amd_gpu.hpp:236-252, src/cg.hpp:174,219); here every phase of
// the V-cycle carries a range as well, so a rocprofv3 --marker-trace reads like the algorithm.
void range_push(const char* name);
void range_pop();
struct Range
{
  explicit Range(const char* name) { range_push(name); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
  ~Range() { range_pop(); }
};

// solvers.hip -- the 4th-kind Chebyshev / Jacobi smoother (src/chebyshev.hpp:46-91) and the CG loop
// (src/cg.hpp:147-222) over ANY operator and preconditioner given as callables (the matrix-free
// Laplacian, an assembled CSR level of the AMG hierarchy; the diagonal, a V-cycle, an AMG cycle)
struct ChebWork
{
  double *r = nullptr, *z = nullptr, *q = nullptr;
};
using ApplyFn = std::function<int(double* in, double* out)>;
using PrecondFn = std::function<int(double* z, const double* r)>;
int cg_iterate(pmg_cg cg, const ApplyFn& A, const double* dinv, const PrecondFn* M, bool flexible, double* x,
               const double* b, int* iterations, hipStream_t s);
enum : int
{
  ResidualNone = 0,    // only x is wanted
  ResidualUpdated = 1, // w.r = b - A x on return
  ResidualSplit = 2    // b - A x = w.r - w.q is left to the consumer when *split comes back true
};
// A_zeroed (optional): the same operator for an output vector that is already zero over [0, n_total) -- given for
// an operator whose launch accumulates with atomics; the smoother's vector kernels then clear w.q behind themselves
// and every application after the first goes through A_zeroed (no zero-fill kernels).
// Ghost bookkeeping of the iterate (several ranks; round 4): every operator application refreshes the ghost entries of
// its INPUT (src/laplacian.hpp:378,425), so the ghosts of z are current right behind A z.  With track_ghosts the
// smoother adds them to the ghost entries of x there (a kernel over the ghost range only) -- after a smooth that
// returns its residual, in which every correction has been applied to, x then has current ghosts without an exchange
// of its own.  A_first (optional): the operator for the FIRST application, A x, when the caller knows that x's ghosts
// are current already (no exchange).
int cheb_iterate(const ChebWork& w, const ApplyFn& A, const double* dinv, int n, double lmax, int max_iter,
                 double* x, const double* b, int need_r, bool x_zero, hipStream_t s, bool* split = nullptr,
                 const ApplyFn* A_zeroed = nullptr, int n_total = 0, bool track_ghosts = false,
                 const ApplyFn* A_first = nullptr);

// vector.hip -- stream-ordered building blocks used by the solvers
// local dot of the owned entries into the result slot `slot` of the layout (device)
int dot_async(pmg_layout l, const double* a, const double* b, int slot, hipStream_t s);
double* red_slot(pmg_layout l, int slot);
// sum / maximise the slots [slot, slot + n) over the ranks; the device slots then hold the global values
int reduce_slots_async(pmg_layout l, int slot, int n, bool max, hipStream_t s);
// ... and bring them to the host: the one host synchronisation of a reduction
int fetch_slots(pmg_layout l, int slot, int n, double* host_out, hipStream_t s);
int dot_host(pmg_layout l, const double* a, const double* b, double* result, hipStream_t s);
void launch_axpy(int n, double* r, double alpha, const double* x, const double* y, hipStream_t s);
void launch_pointwise(int n, double* w, const double* x, const double* y, hipStream_t s);
// Chebyshev fused passes (src/chebyshev.hpp:57-83)
// clear_q (optional): the operator's output, zeroed over [0, n_total) behind the update (see ThenClearF)
void launch_cheb_init(int n, double* r, double* z, const double* b, const double* q,
                      const double* dinv, double c0, hipStream_t s, double* clear_q = nullptr, int n_total = 0);
void launch_cheb_step(int n, double* x, double* r, double* z, const double* q, const double* dinv,
                      double c1, double c2, bool both, int x_final, hipStream_t s, double* clear_q = nullptr,
                      int n_total = 0);
void launch_cheb_residual(int n, double* r, const double* q, hipStream_t s);
void launch_add(int n, double* x, const double* z, hipStream_t s);
void launch_cheb_last(int n, double* x, double* r, const double* z, const double* q, bool assign,
                      hipStream_t s);
void launch_zero(int n, double* x, hipStream_t s);
void launch_mask_bc(int n, double* b, const int8_t* bc, hipStream_t s);
// CG fused passes (src/cg.hpp:160-211); alpha = rnorm / *d_py and beta = (*d_new - *d_sub) / rnorm are
// formed on the device from the reduced scalars, so no host round trip sits between the kernels
void launch_cg_update(int n, double* x, double* r, double* y, const double* p, const double* dinv,
                      double rnorm, const double* d_py, hipStream_t s);
void launch_cg_update2(int n, double* x, double* r, const double* p, const double* y, double rnorm,
                       const double* d_py, hipStream_t s);
void launch_cg_direction(int n, double* p, const double* y, double rnorm, const double* d_new,
                         const double* d_sub, hipStream_t s);
} // namespace pmg
