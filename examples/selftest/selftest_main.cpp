// Self-test of the C++ adapter (include/pmg_amd.hpp) below the drivers: vector
// algebra and the halo scatter with an exchange callback supplied from C++ (a rank
// that is its own periodic neighbour, so the callback is a device-to-device copy on
// the library's stream), compute_boundary_cells, and the error behaviour the
// reference's classes have (std::runtime_error, src/vector.hpp:343, src/laplacian.hpp:346).
// Exit code 0 = all checks passed.  Run by tests/test_gpu_drivers.py.
#include "pmg_amd.hpp"

#include <cmath>
#include <cstdio>
#include <numeric>
#include <random>

using namespace pmg_amd;
using T = double;
using DeviceVector = acc::Vector<T, acc::Device::HIP>;

static int failures = 0;
#define CHECK(cond)                                                                                \
  do                                                                                               \
  {                                                                                                \
    if (!(cond))                                                                                   \
    {                                                                                              \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                                \
      ++failures;                                                                                  \
    }                                                                                              \
  } while (0)

struct SelfHalo
{
  IndexMap* map = nullptr;
  int calls[4] = {0, 0, 0, 0};
  static int exchange(void* user, int phase, pmg_stream stream)
  {
    auto* h = static_cast<SelfHalo*>(user);
    ++h->calls[phase];
    auto send = h->map->send_buffer();
    auto recv = h->map->recv_buffer();
    hipError_t e = hipSuccess;
    if (phase == 0) // owners' packed values -> ghosts
      e = hipMemcpyAsync(recv.data(), send.data(), sizeof(double) * recv.size(), hipMemcpyDeviceToDevice,
                         (hipStream_t)stream);
    else if (phase == 2) // ghost values -> owners
      e = hipMemcpyAsync(send.data(), recv.data(), sizeof(double) * send.size(), hipMemcpyDeviceToDevice,
                         (hipStream_t)stream);
    return e == hipSuccess ? 0 : 1;
  }
  static int allreduce(void*, double*, int) { return 0; } // one rank: the sum / maximum is the value
};

template <typename F>
static bool throws(F&& f, const char* needle)
{
  try
  {
    f();
  }
  catch (const std::runtime_error& e)
  {
    return std::string(e.what()).find(needle) != std::string::npos;
  }
  return false;
}

int main()
{
  try
  {
    const int n = 100003, m = 4099; // odd sizes: the 16-byte paths have tails
    std::mt19937_64 rng(7);
    std::normal_distribution<double> dist;
    std::vector<std::int32_t> send(m), recv(m);
    std::vector<std::int32_t> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int j = 0; j < m; ++j)
    {
      send[j] = perm[j];
      recv[j] = j;
    }
    SelfHalo halo;
    auto nomax = std::make_shared<IndexMap>(n, m, send, recv, &SelfHalo::exchange, &SelfHalo::allreduce, &halo);
    auto map = std::make_shared<IndexMap>(n, m, send, recv, &SelfHalo::exchange, &SelfHalo::allreduce, &halo,
                                          &SelfHalo::allreduce);
    halo.map = map.get();

    std::vector<double> a(n + m), b(n + m);
    for (auto& v : a)
      v = dist(rng);
    for (auto& v : b)
      v = dist(rng);
    DeviceVector x(map, 1), y(map, 1), r(map, 1);
    x.copy_from_host(a);
    y.copy_from_host(b);

    // forward scatter: ghosts <- owners
    x.scatter_fwd_begin();
    x.scatter_fwd_end();
    std::vector<double> got = x.data_copy();
    bool ok = true;
    for (int j = 0; j < m; ++j)
      ok = ok && got[n + j] == a[send[j]];
    for (int i = 0; i < n; ++i)
      ok = ok && got[i] == a[i];
    CHECK(ok);
    CHECK(halo.calls[0] == 1 && halo.calls[1] == 1);
    // reverse scatter: owners += ghosts
    x.scatter_rev_begin();
    x.scatter_rev_end();
    got = x.data_copy();
    std::vector<double> ref(a.begin(), a.begin() + n);
    for (int j = 0; j < m; ++j)
      ref[send[j]] += a[send[j]];
    double err = 0;
    for (int i = 0; i < n; ++i)
      err = std::max(err, std::abs(got[i] - ref[i]));
    CHECK(err < 1e-14);
    CHECK(halo.calls[2] == 1 && halo.calls[3] == 1);

    // BLAS-1 against the host (owned entries)
    x.copy_from_host(a);
    long double dot = 0, nn = 0, amax = 0;
    for (int i = 0; i < n; ++i)
    {
      dot += (long double)a[i] * b[i];
      nn += (long double)a[i] * a[i];
      amax = std::max(amax, (long double)std::abs(a[i]));
    }
    CHECK(std::abs(acc::inner_product(x, y) - (double)dot) < 1e-10 * n);
    CHECK(std::abs(acc::squared_norm(x) - (double)nn) < 1e-12 * (double)nn);
    CHECK(std::abs(acc::norm(x) - std::sqrt((double)nn)) < 1e-12 * std::sqrt((double)nn));
    CHECK(acc::norm(x, acc::Norm::linf) == (double)amax);
    acc::axpy(r, -0.75, x, y);
    got = r.data_copy();
    err = 0;
    for (int i = 0; i < n; ++i)
      err = std::max(err, std::abs(got[i] - (-0.75 * a[i] + b[i])));
    CHECK(err < 1e-15);
    acc::pointwise_mult(r, x, y);
    got = r.data_copy();
    ok = true;
    for (int i = 0; i < n; ++i)
      ok = ok && got[i] == a[i] * b[i];
    CHECK(ok);
    acc::copy(r, x);
    acc::scale(r, 3.0);
    got = r.data_copy();
    ok = true;
    for (int i = 0; i < n; ++i)
      ok = ok && got[i] == 3.0 * a[i];
    CHECK(ok);
    r.set(1.5);
    got = r.data_copy();
    CHECK(got.front() == 1.5 && got.back() == 1.5);

    {
      DeviceVector w(nomax, 1); // several ranks but no max-reduction callback: linf must refuse
      CHECK(throws([&] { acc::norm(w, acc::Norm::linf); }, "allreduce_max"));
    }

    // The library's RCCL communicator with this rank as its own neighbour: a forward scatter captured into a hipGraph
    // (on a HIP >= 7.2 runtime the communicator's stream is forked into the capture, so the exchange stays a parallel
    // branch: pmg_comm_capture_overlaps) must reproduce the eager one, replay after replay.
    {
      auto comm = std::make_shared<const Communicator>(0, 1, Communicator::unique_id());
      const std::int32_t nb[1] = {0}, cnt[1] = {m};
      auto cmap = std::make_shared<IndexMap>(n, m, send, recv, comm, nb, cnt, cnt);
      DeviceVector xe(cmap, 1), xg(cmap, 1), w(cmap, 1);
      xe.copy_from_host(a);
      xg.copy_from_host(a);
      w.copy_from_host(b);
      xe.scatter_fwd_begin(); // eager first: peer connections are set up outside any capture
      xe.scatter_fwd_end();
      hipStream_t cs;
      bool hip_ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
      hip_ok = hip_ok && hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) == hipSuccess;
      bool lib_ok = pmg_scatter_fwd_begin(cmap->layout(), xg.mutable_array().data(), (pmg_stream)cs) == PMG_OK;
      lib_ok = lib_ok && pmg_vec_scale(cmap->layout(), w.mutable_array().data(), 2.0, (pmg_stream)cs) == PMG_OK; // "interior work"
      lib_ok = lib_ok && pmg_scatter_fwd_end(cmap->layout(), xg.mutable_array().data(), (pmg_stream)cs) == PMG_OK;
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      hip_ok = hip_ok && hipStreamEndCapture(cs, &graph) == hipSuccess && graph != nullptr;
      hip_ok = hip_ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      CHECK(hip_ok && lib_ok);
      if (hip_ok && lib_ok)
      {
        for (int rep = 0; rep < 3; ++rep)
        {
          std::vector<double> mod(a);
          for (int j = 0; j < m; ++j)
            mod[send[j]] = a[send[j]] + rep; // new owner values every replay
          xg.copy_from_host(mod);
          CHECK(hipDeviceSynchronize() == hipSuccess);
          CHECK(hipGraphLaunch(exec, cs) == hipSuccess);
          CHECK(hipStreamSynchronize(cs) == hipSuccess);
          std::vector<double> gg = xg.data_copy();
          bool same = true;
          for (int j = 0; j < m; ++j)
            same = same && gg[n + j] == mod[send[j]];
          CHECK(same);
        }
        std::vector<double> ge = xe.data_copy();
        bool eager_ok = true;
        for (int j = 0; j < m; ++j)
          eager_ok = eager_ok && ge[n + j] == a[send[j]];
        CHECK(eager_ok);
        std::printf("captured exchange: %s\n", pmg_comm_capture_overlaps() ? "communicator stream forked into the capture"
                                                                            : "issued on the capturing stream");
      }
      if (exec)
        (void)hipGraphExecDestroy(exec);
      if (graph)
        (void)hipGraphDestroy(graph);
      (void)hipStreamDestroy(cs);
    }

    // Halo windows with this rank as its own neighbour (pmg_layout_set_windows through the adapter): forward scatters,
    // eager and replayed from a hipGraph, and the reverse scatter.
    {
      auto comm = std::make_shared<const Communicator>(0, 1, Communicator::unique_id());
      const std::int32_t nb[1] = {0}, cnt[1] = {m};
      auto wmap = std::make_shared<IndexMap>(n, m, send, recv, comm, nb, cnt, cnt, pmg_amd::Halo::windows);
      DeviceVector xw(wmap, 1);
      bool ok_w = true;
      for (int rep = 0; rep < 5; ++rep) // more exchanges than window slots
      {
        std::vector<double> mod(a);
        for (int j = 0; j < m; ++j)
          mod[send[j]] = a[send[j]] + rep;
        xw.copy_from_host(mod);
        xw.scatter_fwd_begin();
        xw.scatter_fwd_end();
        std::vector<double> gw = xw.data_copy();
        for (int j = 0; j < m; ++j)
          ok_w = ok_w && gw[n + j] == mod[send[j]];
      }
      CHECK(ok_w);
      hipStream_t cs;
      bool hip_ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
      CHECK(hipDeviceSynchronize() == hipSuccess);
      hip_ok = hip_ok && hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed) == hipSuccess;
      bool lib_ok = pmg_scatter_fwd_begin(wmap->layout(), xw.mutable_array().data(), (pmg_stream)cs) == PMG_OK;
      lib_ok = lib_ok && pmg_scatter_fwd_end(wmap->layout(), xw.mutable_array().data(), (pmg_stream)cs) == PMG_OK;
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      hip_ok = hip_ok && hipStreamEndCapture(cs, &graph) == hipSuccess && graph != nullptr;
      hip_ok = hip_ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      CHECK(hip_ok && lib_ok);
      if (hip_ok && lib_ok)
        for (int rep = 0; rep < 3; ++rep) // the exchange number lives on the device: every replay is a new exchange
        {
          std::vector<double> mod(a);
          for (int j = 0; j < m; ++j)
            mod[send[j]] = a[send[j]] - rep;
          xw.copy_from_host(mod);
          CHECK(hipDeviceSynchronize() == hipSuccess);
          CHECK(hipGraphLaunch(exec, cs) == hipSuccess);
          CHECK(hipStreamSynchronize(cs) == hipSuccess);
          std::vector<double> gw = xw.data_copy();
          bool same = true;
          for (int j = 0; j < m; ++j)
            same = same && gw[n + j] == mod[send[j]];
          CHECK(same);
        }
      if (exec)
        (void)hipGraphExecDestroy(exec);
      if (graph)
        (void)hipGraphDestroy(graph);
      (void)hipStreamDestroy(cs);
    }

    // compute_boundary_cells: 4 cells of 2 dofs, 3 owned cells, 5 owned dofs
    std::vector<std::int32_t> dm = {0, 1, 2, 5, 3, 4, 0, 1};
    auto [lc, bc] = compute_boundary_cells(dm, 3, 4, 2, 5);
    CHECK((lc == std::vector<int>{0, 2}) && (bc == std::vector<int>{1, 3}));

    // error behaviour
    auto small = std::make_shared<IndexMap>(10, 0);
    DeviceVector s(small, 1);
    CHECK(throws([&] { acc::inner_product(x, s); }, "Incompatible vector sizes"));
    CHECK(throws([&] { DeviceVector bad(small, 2); }, "block size"));
    device_array<double> kap(std::vector<double>(1, 1.0));
    device_array<std::int32_t> dmd(std::vector<std::int32_t>(1000, 0)), gd(std::vector<std::int32_t>(8, 0));
    device_array<double> xg(std::vector<double>(24, 0.0));
    device_array<std::int8_t> bcm(std::vector<std::int8_t>(10, 0));
    CHECK(throws(
        [&] {
          acc::MatFreeLaplacian<T> op(9, kap.span(), dmd.span(), xg.span(), gd.span(), {}, {}, {0}, {}, bcm.span());
        },
        "Unsupported degree"));
    CHECK(throws([&] { Interpolator<T> ip(2, 2, dmd.span(), dmd.span(), {}, {}); }, "degree"));
    CHECK(throws([&] { Interpolator<T> ip(LagrangeElement{3}, LagrangeElement{2}, dmd.span(), dmd.span(), {}, {}); },
                 "degree"));
    // a vector of another map with the same size_local is still another map (different halo plan)
    auto twin = std::make_shared<IndexMap>(n, 0);
    DeviceVector tw(twin, 1);
    CHECK(throws([&] { acc::inner_product(x, tw); }, "Incompatible vector sizes"));

    // acc::transform (src/vector.hpp:449-454) with a device functor, thrust_vector(), copy constructor
    x.copy_from_host(a);
    acc::transform(x, [] __host__ __device__(const T& v) { return 2.0 * v + 1.0; });
    got = x.data_copy();
    ok = true;
    for (int i = 0; i < n; ++i)
      ok = ok && got[i] == 2.0 * a[i] + 1.0;
    CHECK(ok);
    CHECK(x.thrust_vector().size() == (std::size_t)(n + m));
    DeviceVector xc(x);
    CHECK(xc.data_copy() == x.data_copy() && xc.map() == x.map());
  }
  catch (const std::exception& e)
  {
    std::printf("unexpected exception: %s\n", e.what());
    return 2;
  }
  if (failures)
    return 1;
  std::printf("adapter selftest passed\n");
  return 0;
}
