// Native communicator: the halo exchange and the scalar reductions of the distributed
// vector issued by the library itself on RCCL (xGMI), with no host callback on the hot path.
//
// Replaces what the reference delegates to dolfinx's common::Scatterer over GPU-aware MPI
// (src/vector.hpp:186-238: pack -> hipDeviceSynchronize -> MPI_Isend/Irecv per neighbour ->
// MPI_Waitall -> unpack) and MPI_Allreduce (src/vector.hpp:350,384):
//
//   begin : pack kernel on the compute stream; event; the communicator's stream waits for it and
//           issues ONE group of ncclSend/ncclRecv, one pair per neighbour rank (a 2x2x2 brick
//           split has 7 neighbours = the 7 xGMI links of a GPU, every pair directly connected);
//           event "arrived" behind the group.
//   (the interior-cell kernels run on the compute stream meanwhile)
//   end   : the compute stream waits for "arrived"; unpack kernel.
//
// Nothing synchronises the host.  Reductions: the block partials are folded on the device, the
// scalars are summed over the ranks with ncclAllReduce on the communicator's stream, and the
// host reads them from pinned memory only where the algorithm branches on the value.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the copy already loaded in the process --
// e.g. PyTorch's -- if there is one): a single-GPU run needs no RCCL at all, and the library
// does not pin the process to one of two copies of librccl with the same soname.
#include "common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>

using namespace pmg;

namespace
{
struct RcclApi
{
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int load_rccl()
{
  if (g_rccl.handle)
    return PMG_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) // a copy that is already part of the process first
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)))
      break;
  for (int i = 0; !h && i < 3; ++i)
    h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h)
    return fail(PMG_ERR_INVALID, "RCCL is not available: %s", dlerror());
#define PMG_SYM(field, name)                                                                       \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                         \
  if (!g_rccl.field)                                                                               \
    return fail(PMG_ERR_INVALID, "RCCL symbol %s not found", name);
  PMG_SYM(GetUniqueId, "ncclGetUniqueId")
  PMG_SYM(CommInitRank, "ncclCommInitRank")
  PMG_SYM(CommDestroy, "ncclCommDestroy")
  PMG_SYM(GroupStart, "ncclGroupStart")
  PMG_SYM(GroupEnd, "ncclGroupEnd")
  PMG_SYM(Send, "ncclSend")
  PMG_SYM(Recv, "ncclRecv")
  PMG_SYM(AllReduce, "ncclAllReduce")
  PMG_SYM(AllGather, "ncclAllGather")
  PMG_SYM(GetErrorString, "ncclGetErrorString")
#undef PMG_SYM
  g_rccl.handle = h;
  return PMG_OK;
}

#define PMG_NCCL(call)                                                                             \
  do                                                                                               \
  {                                                                                                \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess)                                                                         \
      return pmg::fail(PMG_ERR_HIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_),     \
                       __FILE__, __LINE__);                                                        \
  } while (0)
} // namespace

static_assert(sizeof(ncclUniqueId) == PMG_COMM_ID_BYTES, "pmg_comm id size");

extern "C" int pmg_comm_unique_id(char* id)
{
  PMG_REQUIRE(id, "pmg_comm_unique_id: NULL argument");
  PMG_TRY(load_rccl());
  ncclUniqueId u;
  PMG_NCCL(g_rccl.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return PMG_OK;
}

extern "C" int pmg_comm_create(pmg_comm* out, int rank, int nranks, const char* id)
{
  PMG_REQUIRE(out && id && nranks >= 1 && rank >= 0 && rank < nranks, "pmg_comm_create: bad argument");
  PMG_TRY(load_rccl());
  auto* c = new pmg_comm_s;
  HandleGuard<pmg_comm> guard(c, pmg_comm_destroy);
  c->rank = rank;
  c->nranks = nranks;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  PMG_NCCL(g_rccl.CommInitRank(&c->comm, nranks, u, rank)); // on the calling thread's current device
  PMG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  PMG_HIP(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
  PMG_HIP(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
  *out = guard.release();
  return PMG_OK;
}

extern "C" int pmg_comm_destroy(pmg_comm c)
{
  if (!c)
    return PMG_OK;
  if (c->stream)
    (void)hipStreamSynchronize(c->stream);
  wcomm_destroy(c);
  if (c->comm && g_rccl.CommDestroy)
    (void)g_rccl.CommDestroy(c->comm);
  if (c->ev_in)
    (void)hipEventDestroy(c->ev_in);
  if (c->ev_out)
    (void)hipEventDestroy(c->ev_out);
  if (c->stream)
    (void)hipStreamDestroy(c->stream);
  delete c;
  return PMG_OK;
}

// Set-up helper (the reference's set-up uses MPI for this): `bytes` bytes of host memory from every rank, in rank
// order, into recv[nranks * bytes] on every rank.  Blocking; not for the hot path.
extern "C" int pmg_comm_allgather(pmg_comm c, const void* send, size_t bytes, void* recv)
{
  PMG_REQUIRE(c && send && recv && bytes > 0, "pmg_comm_allgather: bad argument");
  if (c->wcomm)
    return wcomm_allgather(c, send, bytes, recv);
  char *d_in = nullptr, *d_out = nullptr;
  PMG_HIP(hipMalloc(&d_in, bytes));
  hipError_t e = hipMalloc(&d_out, bytes * (size_t)c->nranks);
  if (e == hipSuccess)
    e = hipMemcpy(d_in, send, bytes, hipMemcpyHostToDevice);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess)
  {
    r = g_rccl.AllGather(d_in, d_out, bytes, ncclInt8, c->comm, c->stream);
    if (r == ncclSuccess)
      e = hipStreamSynchronize(c->stream);
    if (r == ncclSuccess && e == hipSuccess)
      e = hipMemcpy(recv, d_out, bytes * (size_t)c->nranks, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  PMG_NCCL(r);
  PMG_HIP(e);
  return PMG_OK;
}

// values[0 .. n) (device) summed over the ranks in place, stream-ordered on `stream`
extern "C" int pmg_comm_allreduce_sum(pmg_comm c, double* values, int n, pmg_stream stream)
{
  PMG_REQUIRE(c && values && n >= 0, "pmg_comm_allreduce_sum: bad argument");
  if (n == 0)
    return PMG_OK;
  return comm_allreduce(c, values, n, false, S(stream));
}

extern "C" int pmg_comm_rank(pmg_comm c) { return c ? c->rank : -1; }
extern "C" int pmg_comm_size(pmg_comm c) { return c ? c->nranks : -1; }

extern "C" int pmg_layout_set_comm(pmg_layout l, pmg_comm comm, int32_t n_neighbors,
                                   const int32_t* neighbor_ranks, const int32_t* send_counts,
                                   const int32_t* recv_counts)
{
  PMG_REQUIRE(l && comm, "pmg_layout_set_comm: NULL argument");
  PMG_REQUIRE(n_neighbors >= 0 && (n_neighbors == 0 || (neighbor_ranks && send_counts && recv_counts)),
              "pmg_layout_set_comm: neighbour arrays missing");
  long long ns = 0, nr = 0;
  for (int i = 0; i < n_neighbors; ++i)
  {
    PMG_REQUIRE(neighbor_ranks[i] >= 0 && neighbor_ranks[i] < comm->nranks,
                "pmg_layout_set_comm: neighbour rank %d out of range", neighbor_ranks[i]);
    PMG_REQUIRE(send_counts[i] >= 0 && recv_counts[i] >= 0, "pmg_layout_set_comm: negative count");
    ns += send_counts[i];
    nr += recv_counts[i];
  }
  PMG_REQUIRE(ns == l->n_send && nr == l->n_recv,
              "pmg_layout_set_comm: per-neighbour counts (%lld, %lld) do not add up to the layout's "
              "n_send, n_recv (%d, %d)", ns, nr, l->n_send, l->n_recv);
  l->comm = comm;
  l->exchanged_eagerly = false;
  l->nb_rank.assign(neighbor_ranks, neighbor_ranks + n_neighbors);
  l->nb_send.assign(send_counts, send_counts + n_neighbors);
  l->nb_recv.assign(recv_counts, recv_counts + n_neighbors);
  if (!l->ev_packed)
    PMG_HIP(hipEventCreateWithFlags(&l->ev_packed, hipEventDisableTiming));
  if (!l->ev_arrived)
    PMG_HIP(hipEventCreateWithFlags(&l->ev_arrived, hipEventDisableTiming));
  // padded staging buffers: one 256-byte aligned segment per neighbour
  constexpr size_t ALIGN = 32; // doubles
  auto layout_side = [&](const std::vector<int32_t>& counts, std::vector<size_t>& off, double** buf,
                         int32_t** pos) -> int {
    off.assign(counts.size(), 0);
    std::vector<int32_t> h;
    size_t at = 0;
    for (size_t k = 0; k < counts.size(); ++k)
    {
      off[k] = at;
      for (int32_t j = 0; j < counts[k]; ++j)
        h.push_back((int32_t)(at + (size_t)j));
      at = (at + (size_t)counts[k] + ALIGN - 1) / ALIGN * ALIGN;
    }
    (void)hipFree(*buf);
    (void)hipFree(*pos);
    *buf = nullptr;
    *pos = nullptr;
    PMG_HIP(hipMalloc(buf, sizeof(double) * (at ? at : 1)));
    PMG_HIP(hipMemset(*buf, 0, sizeof(double) * (at ? at : 1)));
    PMG_HIP(hipStreamSynchronize(nullptr)); // (null-stream fill: not ordered against the non-blocking streams)
    PMG_HIP(hipMalloc(pos, sizeof(int32_t) * (h.empty() ? 1 : h.size())));
    if (!h.empty())
      PMG_HIP(hipMemcpy(*pos, h.data(), sizeof(int32_t) * h.size(), hipMemcpyHostToDevice));
    return PMG_OK;
  };
  PMG_TRY(layout_side(l->nb_send, l->send_off, &l->c_send, &l->send_pos));
  PMG_TRY(layout_side(l->nb_recv, l->recv_off, &l->c_recv, &l->recv_pos));
  return PMG_OK;
}

namespace
{
// May the communicator's stream be forked into a stream capture?  The HIP 7.0 runtime (the copy PyTorch 2.10+rocm7.0
// bundles and a Python process therefore runs on) recurses without end in hip::Stream::EndCapture when the capture
// holds a forked stream on which RCCL enqueued (RCCL forks its own stream from it again): a stack overflow at
// hipStreamEndCapture, 174 000 frames deep (gpurun_out / profiles/rccl_capture_probe_r03.md).  The 7.2 runtime ends the
// same capture correctly.  PMG_CAPTURE_FORK=0/1 overrides.
bool capture_fork_supported()
{
  static int cached = -1;
  if (cached < 0)
  {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess)
      v = 0;
    cached = v >= 70200000 ? 1 : 0; // HIP_VERSION = major * 10^7 + minor * 10^5 + patch
    if (const char* e = std::getenv("PMG_CAPTURE_FORK"))
      cached = (e[0] == '1') ? 1 : 0;
  }
  return cached == 1;
}
} // namespace

extern "C" int pmg_comm_capture_overlaps(void) { return capture_fork_supported() ? 1 : 0; }

namespace pmg
{
// One grouped neighbour exchange on the communicator's stream, ordered behind everything enqueued
// on `s` so far (the pack kernel); `reverse` swaps the roles of the two staging buffers
// (ghost -> owner, src/vector.hpp:249-267).  The arrival is signalled by l->ev_arrived.
int comm_exchange_begin(pmg_layout l, bool reverse, hipStream_t s)
{
  pmg_comm c = l->comm;
  if (l->nb_rank.empty()) // no partner: nothing to post (point-to-point groups are not collectives)
    return PMG_OK;
  PMG_REQUIRE(c->comm, "this communicator is made of windows and moves no halo by itself: attach halo windows to the "
                       "layout (pmg_layout_set_windows)");
  // While `s` is being captured into a graph the replayed cycle costs the host one hipGraphLaunch instead of
  // ~115 us per exchange.  On a runtime whose capture can take it (capture_fork_supported) the communicator's
  // stream is forked into the capture exactly as in the eager path -- event on `s`, wait on the communicator's stream,
  // group, event, and the join in comm_exchange_end -- so the exchange is a parallel branch of the graph and still
  // overlaps the interior cells.  Otherwise the group is issued on `s` itself: correct, no overlap.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  PMG_HIP(hipStreamIsCapturing(s, &cap));
  const bool inline_group = cap == hipStreamCaptureStatusActive && !capture_fork_supported();
  hipStream_t cs = inline_group ? s : c->stream;
  l->exchange_inline = inline_group;
  if (!inline_group)
  {
    PMG_HIP(hipEventRecord(l->ev_packed, s));
    PMG_HIP(hipStreamWaitEvent(c->stream, l->ev_packed, 0));
  }
  const double* out = reverse ? l->c_recv : l->c_send;
  double* in = reverse ? l->c_send : l->c_recv;
  const std::vector<int32_t>& nout = reverse ? l->nb_recv : l->nb_send;
  const std::vector<int32_t>& nin = reverse ? l->nb_send : l->nb_recv;
  const std::vector<size_t>& oout = reverse ? l->recv_off : l->send_off;
  const std::vector<size_t>& oin = reverse ? l->send_off : l->recv_off;
  PMG_NCCL(g_rccl.GroupStart());
  ncclResult_t r = ncclSuccess;
  for (size_t i = 0; i < l->nb_rank.size() && r == ncclSuccess; ++i)
  {
    if (nout[i] > 0)
      r = g_rccl.Send(out + oout[i], (size_t)nout[i], ncclDouble, l->nb_rank[i], c->comm, cs);
    if (nin[i] > 0 && r == ncclSuccess)
      r = g_rccl.Recv(in + oin[i], (size_t)nin[i], ncclDouble, l->nb_rank[i], c->comm, cs);
  }
  ncclResult_t re = g_rccl.GroupEnd(); // always close the group
  PMG_NCCL(r);
  PMG_NCCL(re);
  if (!inline_group)
  {
    PMG_HIP(hipEventRecord(l->ev_arrived, c->stream));
    if (cap != hipStreamCaptureStatusActive)
      l->exchanged_eagerly = true;
  }
  return PMG_OK;
}

// May a cycle over this layout be captured into a graph now?  RCCL sets up the connection to a peer (and a
// collective's channels) the first time it is used; that must not happen inside a stream capture.  So the first
// exchange of every layout and the first all-reduce of a communicator are always issued eagerly -- a cycle is
// captured from its second application on.  Every rank reaches the same decision: the flags follow the call
// sequence, which is the same on all ranks of a partitioned problem.
bool comm_capture_ready(pmg_layout l, bool with_allreduce)
{
  if (!l->comm || l->comm->wcomm) // (a communicator of windows: plain kernels, nothing to warm up)
    return true;
  if (!l->win && !l->nb_rank.empty() && !l->exchanged_eagerly) // halo windows are plain kernels: nothing to warm up
    return false;
  return !with_allreduce || l->comm->reduced_eagerly;
}

int comm_exchange_end(pmg_layout l, hipStream_t s)
{
  if (l->nb_rank.empty() || l->exchange_inline)
    return PMG_OK;
  PMG_HIP(hipStreamWaitEvent(s, l->ev_arrived, 0));
  return PMG_OK;
}

// values[0..n) (device) summed / maximised over the ranks in place, stream-ordered on `s`
int comm_allreduce(pmg_layout l, double* d_values, int n, bool max, hipStream_t s)
{
  return comm_allreduce(l->comm, d_values, n, max, s);
}
int comm_allreduce(pmg_comm c, double* d_values, int n, bool max, hipStream_t s)
{
  if (c->wcomm)
    return wcomm_allreduce(c, d_values, n, max, s);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  PMG_HIP(hipStreamIsCapturing(s, &cap));
  if (cap == hipStreamCaptureStatusActive) // inside a graph capture: on `s` itself (see comm_exchange_begin)
  {
    PMG_NCCL(g_rccl.AllReduce(d_values, d_values, (size_t)n, ncclDouble, max ? ncclMax : ncclSum, c->comm, s));
    return PMG_OK;
  }
  PMG_HIP(hipEventRecord(c->ev_in, s));
  PMG_HIP(hipStreamWaitEvent(c->stream, c->ev_in, 0));
  PMG_NCCL(g_rccl.AllReduce(d_values, d_values, (size_t)n, ncclDouble, max ? ncclMax : ncclSum, c->comm,
                            c->stream));
  PMG_HIP(hipEventRecord(c->ev_out, c->stream));
  PMG_HIP(hipStreamWaitEvent(s, c->ev_out, 0));
  c->reduced_eagerly = true;
  return PMG_OK;
}
} // namespace pmg
