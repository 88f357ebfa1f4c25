// pmg_amd.hpp -- C++ adapter over the C ABI of pmg_amd.h.
//
// Re-exports the duck-typed concepts the reference's drivers are written against, with
// the reference's template spellings (Wells-Group/pmg-dolfinx @ 2024_08_07):
//
//   acc::Device, acc::Vector<T, Device>            src/vector.hpp:61-66, 74-325
//   acc::inner_product ... acc::transform          src/vector.hpp:333-454
//   acc::MatFreeLaplacian<T>                       src/laplacian.hpp:284-526
//   acc::Chebyshev<Vector>                         src/chebyshev.hpp:19-106
//   acc::CGSolver<Vector>                          src/cg.hpp:93-250
//   Interpolator<T>                                src/interpolate.hpp:93-329
//   acc::MultigridPreconditioner<Vector, Operator, Solver, CoarseSolver, Interpolator>
//                                                  src/pmg.hpp:13-184
//
// same member names, argument meaning and error behaviour (std::runtime_error), so that
// examples/pmg/main.cpp:306-365 and examples/mat_free/main.cpp:236-288 compile on top of it
// as they stand (examples/pmg/pmg_main.cpp in this repository carries those lines).  T is
// double and Device is HIP: the library is FP64 on gfx950, like the reference's
// `using T = double` (examples/pmg/main.cpp:30); anything else is a compile-time error.
//
// dolfinx is not a dependency.  The one dolfinx type on the path, common::IndexMap (+ its
// Scatterer), is replaced by pmg_amd::IndexMap, which carries the same information flattened
// to arrays (INTEGRATION.md shows how a maintainer builds it from a dolfinx IndexMap), and the
// basix elements the Interpolator is constructed from by anything with a degree() member
// (basix::FiniteElement has one; pmg_amd::LagrangeElement is the stand-in).  Define
// PMG_AMD_DOLFINX_NAMESPACE before including this header to get the names under the
// reference's own namespaces (dolfinx::acc::..., dolfinx::common::IndexMap, ::Interpolator).
//
// Header-only, C++20 (std::span, like the reference); compile with hipcc (the vector storage is
// a thrust::device_vector, exactly as in the reference, so thrust_vector() and acc::transform
// exist); link with libpmg_amd.so and amdhip64.
#pragma once

#include "pmg_amd.h"

#include <hip/hip_runtime.h>
#include <thrust/copy.h>
#include <thrust/device_vector.h>
#include <thrust/execution_policy.h>
#include <thrust/transform.h>

#include <unistd.h>

#include <array>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <random>
#include <span>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace pmg_amd
{
/// Identity of this process among the ranks that exchange window handles: a random token (pid in the low bits), not
/// the bare pid -- ranks in different PID namespaces can share a pid.
inline std::uint64_t process_token()
{
  static const std::uint64_t token = [] {
    std::random_device rd;
    return ((std::uint64_t)rd() << 32) ^ ((std::uint64_t)rd() << 11) ^ (std::uint64_t)getpid();
  }();
  return token;
}

inline void check(int rc)
{
  if (rc != PMG_OK)
    throw std::runtime_error(pmg_last_error());
}
inline void hip_check(hipError_t e, const char* what)
{
  if (e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

/// Owning device array for the drivers' mesh data (dofmaps, coordinates, markers).
template <typename T>
class device_array
{
public:
  device_array() = default;
  explicit device_array(std::size_t n) { resize(n); }
  explicit device_array(std::span<const T> host) { assign(host); }
  device_array(const device_array&) = delete;
  device_array& operator=(const device_array&) = delete;
  device_array(device_array&& o) noexcept : _p(std::exchange(o._p, nullptr)), _n(std::exchange(o._n, 0)) {}
  device_array& operator=(device_array&& o) noexcept
  {
    std::swap(_p, o._p);
    std::swap(_n, o._n);
    return *this;
  }
  ~device_array()
  {
    if (_p)
      (void)hipFree(_p);
  }
  void resize(std::size_t n)
  {
    if (_p)
      (void)hipFree(_p);
    _p = nullptr;
    _n = n;
    if (n)
    {
      hip_check(hipMalloc((void**)&_p, n * sizeof(T)), "hipMalloc");
      hip_check(hipMemset(_p, 0, n * sizeof(T)), "hipMemset");
    }
  }
  void assign(std::span<const T> host)
  {
    resize(host.size());
    if (_n)
      hip_check(hipMemcpy(_p, host.data(), _n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
  }
  std::vector<T> to_host() const
  {
    std::vector<T> h(_n);
    if (_n)
      hip_check(hipMemcpy(h.data(), _p, _n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H");
    return h;
  }
  T* data() { return _p; }
  const T* data() const { return _p; }
  std::size_t size() const { return _n; }
  std::span<T> span() { return {_p, _n}; }
  std::span<const T> span() const { return {_p, _n}; }

private:
  T* _p = nullptr;
  std::size_t _n = 0;
};

/// One process's handle on the native RCCL communicator (pmg_comm, see pmg_amd.h): the role MPI
/// plays under dolfinx's Scatterer.  Rank 0 creates the id, every rank constructs from it.
class Communicator
{
public:
  static std::array<char, PMG_COMM_ID_BYTES> unique_id()
  {
    std::array<char, PMG_COMM_ID_BYTES> id{};
    check(pmg_comm_unique_id(id.data()));
    return id;
  }
  Communicator(int rank, int size, const std::array<char, PMG_COMM_ID_BYTES>& id)
  {
    check(pmg_comm_create(&_c, rank, size, id.data()));
  }
  /// What a rank publishes about its window for the communicator made of windows (pmg_comm_create_windows).
  struct WindowHandle
  {
    char handle[PMG_WINDOW_HANDLE_BYTES] = {};
    std::uint64_t pid = 0;   // process_token() of the owner
    void* pointer = nullptr; // meaningful inside that process only
  };
  /// A communicator without a transport library: `allgather` hands this rank's WindowHandle to every rank and returns
  /// all of them in rank order -- once, at set-up, by whatever means the caller has (files, MPI_Allgather, a socket).
  Communicator(int rank, int size, const std::function<std::vector<WindowHandle>(const WindowHandle&)>& allgather)
  {
    WindowHandle mine;
    check(pmg_window_alloc(PMG_COMM_WINDOW_BYTES, &_window, mine.handle));
    mine.pid = process_token();
    mine.pointer = _window;
    const std::vector<WindowHandle> all = allgather(mine);
    if ((int)all.size() != size)
      throw std::runtime_error("Communicator: the window handles of all ranks are needed");
    std::vector<void*> windows((std::size_t)size, nullptr);
    for (int r = 0; r < size; ++r)
    {
      if (r == rank || all[(std::size_t)r].pid == mine.pid)
        windows[(std::size_t)r] = all[(std::size_t)r].pointer;
      else
      {
        check(pmg_window_open(all[(std::size_t)r].handle, &windows[(std::size_t)r]));
        _opened.push_back(windows[(std::size_t)r]);
      }
    }
    check(pmg_comm_create_windows(&_c, rank, size, windows.data()));
    _windows = true;
  }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  ~Communicator()
  {
    pmg_comm_destroy(_c);
    for (void* p : _opened)
      pmg_window_close(p);
    pmg_window_free(_window);
  }
  /// true: made of windows -- its IndexMaps need Halo::windows
  bool windows() const { return _windows; }
  int rank() const { return pmg_comm_rank(_c); }
  int size() const { return pmg_comm_size(_c); }
  pmg_comm handle() const { return _c; }
  /// Set-up time: one trivially copyable record from every rank, in rank order (MPI_Allgather's role).
  template <typename R>
  std::vector<R> allgather(const R& mine) const
  {
    static_assert(std::is_trivially_copyable_v<R>);
    std::vector<R> all((std::size_t)size());
    check(pmg_comm_allgather(_c, &mine, sizeof(R), all.data()));
    return all;
  }

private:
  pmg_comm _c = nullptr;
  void* _window = nullptr;
  std::vector<void*> _opened;
  bool _windows = false;
};

/// How the halo of an IndexMap with a communicator travels: grouped ncclSend / ncclRecv, or direct stores into the
/// neighbours' windows (pmg_layout_set_windows; RCCL then serves the reductions only).
enum class Halo
{
  exchange,
  windows
};

/// Stand-in for dolfinx::common::IndexMap + common::Scatterer on this path:
/// owned/ghost sizes, the packed index lists of the halo
/// (Scatterer::local_indices() / remote_indices(), src/vector.hpp:91-92) and who moves the
/// packed buffers: the library's RCCL communicator (second constructor) or the caller's
/// callbacks (first constructor; MPI).  Owns the device copies and the
/// staging buffers; every Vector / operator / solver on this map shares them.
class IndexMap
{
public:
  IndexMap(std::int32_t size_local, std::int32_t num_ghosts, std::span<const std::int32_t> send_indices = {},
           std::span<const std::int32_t> recv_indices = {}, pmg_exchange_fn exchange = nullptr,
           pmg_allreduce_fn allreduce_sum = nullptr, void* user = nullptr, pmg_allreduce_fn allreduce_max = nullptr)
      : _size_local(size_local), _num_ghosts(num_ghosts), _send_idx(send_indices), _recv_idx(recv_indices),
        _send(send_indices.size()), _recv(recv_indices.size())
  {
    check(pmg_layout_create(&_layout, size_local, num_ghosts, (std::int32_t)send_indices.size(), _send_idx.data(),
                            _send.data(), (std::int32_t)recv_indices.size(), _recv_idx.data(), _recv.data(),
                            exchange, allreduce_sum, user));
    if (allreduce_max)
      check(pmg_layout_set_allreduce_max(_layout, allreduce_max));
  }
  /// Halo over the native communicator: index lists grouped by neighbour, in the order of
  /// `neighbors`, with the per-neighbour counts.
  IndexMap(std::int32_t size_local, std::int32_t num_ghosts, std::span<const std::int32_t> send_indices,
           std::span<const std::int32_t> recv_indices, std::shared_ptr<const Communicator> comm,
           std::span<const std::int32_t> neighbors, std::span<const std::int32_t> send_counts,
           std::span<const std::int32_t> recv_counts, Halo halo = Halo::exchange)
      : IndexMap(size_local, num_ghosts, send_indices, recv_indices)
  {
    if (neighbors.size() != send_counts.size() || neighbors.size() != recv_counts.size())
      throw std::runtime_error("IndexMap: neighbour arrays of different lengths");
    _comm = std::move(comm);
    check(pmg_layout_set_comm(_layout, _comm->handle(), (std::int32_t)neighbors.size(), neighbors.data(),
                              send_counts.data(), recv_counts.data()));
    if (halo == Halo::windows || _comm->windows())
      attach_windows(neighbors, send_counts, recv_counts);
  }
  IndexMap(const IndexMap&) = delete;
  IndexMap& operator=(const IndexMap&) = delete;
  ~IndexMap()
  {
    pmg_layout_destroy(_layout);
    for (void* p : _opened)
      pmg_window_close(p);
    pmg_window_free(_window);
    pmg_window_free(_flags);
  }

  std::int32_t size_local() const { return _size_local; }
  std::int32_t num_ghosts() const { return _num_ghosts; }
  pmg_layout layout() const { return _layout; }
  /// The packed send / receive staging buffers (device), for an exchange callback.
  std::span<double> send_buffer() { return _send.span(); }
  std::span<double> recv_buffer() { return _recv.span(); }

private:
  // What a rank tells the others about its window (collective over the communicator: every rank constructs its
  // IndexMaps in the same order).
  struct WindowRecord
  {
    std::int32_t n = 0;
    std::uint64_t pid = 0; // process_token() of the owner
    std::int32_t neighbors[PMG_WINDOW_MAX_NEIGHBORS] = {};
    std::int64_t doubles = 0, fwd[PMG_WINDOW_MAX_NEIGHBORS] = {}, rev[PMG_WINDOW_MAX_NEIGHBORS] = {};
    char window[PMG_WINDOW_HANDLE_BYTES] = {}, flags[PMG_WINDOW_HANDLE_BYTES] = {};
    void *window_ptr = nullptr, *flags_ptr = nullptr; // meaningful inside that process only
  };
  void attach_windows(std::span<const std::int32_t> neighbors, std::span<const std::int32_t> send_counts,
                      std::span<const std::int32_t> recv_counts)
  {
    const int n = (int)neighbors.size();
    auto mine = std::make_unique<WindowRecord>();
    check(pmg_layout_window_describe(n, send_counts.data(), recv_counts.data(), &mine->doubles, mine->fwd, mine->rev));
    check(pmg_window_alloc(sizeof(double) * (std::size_t)mine->doubles, &_window, mine->window));
    check(pmg_window_alloc(sizeof(std::uint64_t) * PMG_WINDOW_FLAG_WORDS, &_flags, mine->flags));
    mine->n = n;
    mine->pid = process_token();
    mine->window_ptr = _window;
    mine->flags_ptr = _flags;
    for (int k = 0; k < n; ++k)
      mine->neighbors[k] = neighbors[k];
    const std::vector<WindowRecord> all = _comm->allgather(*mine);
    const int me = _comm->rank();
    std::vector<double*> nb_window((std::size_t)n);
    std::vector<std::uint64_t*> nb_flags((std::size_t)n);
    std::vector<std::int64_t> nb_doubles((std::size_t)n), nb_fwd((std::size_t)n), nb_rev((std::size_t)n);
    std::vector<std::int32_t> nb_slot((std::size_t)n);
    std::map<int, std::pair<void*, void*>> mapped;
    std::map<int, int> seen;
    for (int k = 0; k < n; ++k)
    {
      const int r = neighbors[k];
      const WindowRecord& info = all.at((std::size_t)r);
      // the j-th time rank r appears in my list pairs with the j-th time I appear in r's list
      int j = seen[r]++, slot = -1;
      for (int i = 0; i < info.n; ++i)
        if (info.neighbors[i] == me && j-- == 0)
        {
          slot = i;
          break;
        }
      if (slot < 0)
        throw std::runtime_error("IndexMap: a neighbour does not list this rank as its neighbour");
      if (!mapped.count(r))
      {
        if (info.pid == mine->pid) // my own window (a rank that is its own periodic neighbour)
          mapped[r] = {info.window_ptr, info.flags_ptr};
        else
        {
          void *w = nullptr, *f = nullptr;
          check(pmg_window_open(info.window, &w));
          _opened.push_back(w);
          check(pmg_window_open(info.flags, &f));
          _opened.push_back(f);
          mapped[r] = {w, f};
        }
      }
      nb_window[(std::size_t)k] = static_cast<double*>(mapped[r].first);
      nb_flags[(std::size_t)k] = static_cast<std::uint64_t*>(mapped[r].second);
      nb_doubles[(std::size_t)k] = info.doubles;
      nb_fwd[(std::size_t)k] = info.fwd[slot];
      nb_rev[(std::size_t)k] = info.rev[slot];
      nb_slot[(std::size_t)k] = slot;
    }
    check(pmg_layout_set_windows(_layout, n, send_counts.data(), recv_counts.data(), static_cast<double*>(_window),
                                 static_cast<std::uint64_t*>(_flags), nb_window.data(), nb_flags.data(),
                                 nb_doubles.data(), nb_fwd.data(), nb_rev.data(), nb_slot.data()));
    (void)_comm->allgather(me); // every rank has mapped its neighbours before anybody stores
  }

  std::int32_t _size_local, _num_ghosts;
  device_array<std::int32_t> _send_idx, _recv_idx;
  device_array<double> _send, _recv;
  std::shared_ptr<const Communicator> _comm;
  pmg_layout _layout = nullptr;
  void *_window = nullptr, *_flags = nullptr;
  std::vector<void*> _opened;
};

/// compute_boundary_cells (src/mesh.hpp:105-143) on flattened inputs: cells that touch
/// no ghost dof ("local": they can run while the halo is in flight) and cells that do,
/// plus every ghost cell ("boundary").  `dofmap` is a host array [ncells][N]; the first
/// ncells_owned cells are owned, the rest are ghost cells.
inline std::pair<std::vector<int>, std::vector<int>>
compute_boundary_cells(std::span<const std::int32_t> dofmap, std::int32_t ncells_owned, std::int32_t ncells, int N,
                       std::int32_t size_local)
{
  if ((std::size_t)ncells * N != dofmap.size() || ncells_owned > ncells)
    throw std::runtime_error("compute_boundary_cells: dofmap size does not match the cell counts");
  std::vector<int> local_cells, boundary_cells;
  for (std::int32_t c = 0; c < ncells; ++c)
  {
    bool mark = c >= ncells_owned;
    for (int k = 0; !mark && k < N; ++k)
      mark = dofmap[(std::size_t)c * N + k] >= size_local;
    (mark ? boundary_cells : local_cells).push_back(c);
  }
  return {std::move(local_cells), std::move(boundary_cells)};
}

/// Cell-local node order of the dofmaps (and of dphi_geometry / G_weights) a caller hands over -- pmg_amd.h,
/// "cell-local node order".  `ascending`: nodes numbered by coordinate along every direction;
/// `endpoints_first`: the order of a basix tensor-product element, vertex 0, vertex 1, interior left to right
/// (what dolfinx's create_functionspace gives the reference, examples/pmg/main.cpp:83-87).  The constructors
/// default to default_node_order: endpoints_first under PMG_AMD_DOLFINX_NAMESPACE (a translation unit written
/// against dolfinx), ascending otherwise (the self-contained drivers of examples/ generate ascending dofmaps);
/// define PMG_AMD_DEFAULT_NODE_ORDER (0 / 1) to choose explicitly.
enum class NodeOrder : int
{
  ascending = PMG_NODES_ASCENDING,
  endpoints_first = PMG_NODES_ENDPOINTS_FIRST
};
#if defined(PMG_AMD_DEFAULT_NODE_ORDER)
inline constexpr NodeOrder default_node_order = static_cast<NodeOrder>(PMG_AMD_DEFAULT_NODE_ORDER);
#elif defined(PMG_AMD_DOLFINX_NAMESPACE)
inline constexpr NodeOrder default_node_order = NodeOrder::endpoints_first;
#else
inline constexpr NodeOrder default_node_order = NodeOrder::ascending;
#endif
/// The caller's array [ncells][nd^3] (ascending) as a caller in basix order would hold it, and back -- for
/// drivers and tests that generate their own meshes.
inline std::vector<std::int32_t> dofmap_in_node_order(std::span<const std::int32_t> dofmap, int degree,
                                                      NodeOrder order)
{
  const int nd = degree + 1, N = nd * nd * nd;
  std::vector<std::int32_t> p1(nd), out(dofmap.size());
  check(pmg_node_permutation(static_cast<int>(order), degree, nullptr, p1.data()));
  for (std::size_t c = 0; c < dofmap.size() / N; ++c)
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nd; ++b)
        for (int k = 0; k < nd; ++k)
          out[c * N + (a * nd + b) * nd + k] = dofmap[c * N + (p1[a] * nd + p1[b]) * nd + p1[k]];
  return out;
}

/// What the Interpolator needs to know about a basix GLL Lagrange element: its degree.
struct LagrangeElement
{
  int p;
  int degree() const { return p; }
};

namespace acc
{
/// src/vector.hpp:61-66.  Only HIP exists here.
enum class Device
{
  CUDA,
  HIP,
  CPP
};

enum class Norm
{
  l2,
  linf
};

/// acc::Vector<T, Device> (src/vector.hpp:74-325), block size 1.
template <typename T, Device D>
class Vector
{
  static_assert(std::is_same_v<T, double>, "pmg_amd: the library computes in double (examples/pmg/main.cpp:30)");
  static_assert(D == Device::HIP, "pmg_amd: gfx950 only -- use acc::Device::HIP");

public:
  using value_type = T;
  constexpr static Device device = D;

  Vector(std::shared_ptr<const IndexMap> map, int bs) : _map(std::move(map))
  {
    if (bs != 1)
      throw std::runtime_error("Vector: block size must be 1");
    _x = thrust::device_vector<T>((std::size_t)_map->size_local() + _map->num_ghosts(), T(0)); // :88
  }
  Vector(const Vector& x) = default; // :99 (the halo plan lives in the map, so nothing is lost)
  Vector(Vector&&) = default;
  Vector& operator=(const Vector& x) = delete; // :102
  Vector& operator=(Vector&& x) = default;     // :105

  void set(T v) { check(pmg_vec_set(_map->layout(), data(), v, nullptr)); } // :109-115
  /// Owned part from a host container with array() (la::Vector) or a contiguous range (:117-122).
  template <typename OtherVector>
  void copy_from_host(const OtherVector& other)
  {
    const T* src;
    std::size_t n;
    if constexpr (requires { other.array(); })
    {
      src = other.array().data();
      n = other.array().size();
    }
    else
    {
      src = std::data(other);
      n = std::size(other);
    }
    if (n < (std::size_t)_map->size_local())
      throw std::runtime_error("copy_from_host: source shorter than size_local");
    hip_check(hipMemcpy(data(), src, sizeof(T) * _map->size_local(), hipMemcpyHostToDevice), "hipMemcpy H2D");
  }
  template <typename OtherVector>
  void copy(OtherVector& other) // :124-129
  {
    _x.resize(other.array().size());
    thrust::copy(other.array().begin(), other.array().end(), _x.begin());
  }
  std::shared_ptr<const IndexMap> map() const { return _map; }                    // :132
  constexpr int bs() const { return 1; }                                          // :135
  thrust::device_vector<T>& thrust_vector() { return _x; }                        // :138
  std::span<const T> array() const { return {thrust::raw_pointer_cast(_x.data()), _x.size()}; } // :141-150
  std::span<T> mutable_array() { return {thrust::raw_pointer_cast(_x.data()), _x.size()}; }     // :153-162
  /// Host copy of owned + ghost entries (data_copy(), :296-302).
  std::vector<T> data_copy() const
  {
    std::vector<T> h(_x.size());
    thrust::copy(_x.begin(), _x.end(), h.begin());
    return h;
  }

  void scatter_fwd_begin() { check(pmg_scatter_fwd_begin(_map->layout(), data(), nullptr)); } // :186-207
  void scatter_fwd_end() { check(pmg_scatter_fwd_end(_map->layout(), data(), nullptr)); }     // :213-238
  void scatter_fwd()                                                                          // :242-246
  {
    scatter_fwd_begin();
    scatter_fwd_end();
  }
  void scatter_rev_begin() { check(pmg_scatter_rev_begin(_map->layout(), data(), nullptr)); } // :249-263
  void scatter_rev_end() { check(pmg_scatter_rev_end(_map->layout(), data(), nullptr)); }     // :271-286

private:
  T* data() { return thrust::raw_pointer_cast(_x.data()); }
  std::shared_ptr<const IndexMap> _map;
  thrust::device_vector<T> _x;
};

/// Two vectors are compatible when they share the halo plan, i.e. the same pmg_layout: equal
/// size_local alone would let an operator run one map's pack / unpack lists on another's storage.
template <typename V1, typename V2>
inline void require_same_map(const V1& a, const V2& b)
{
  if (a.map()->layout() != b.map()->layout())
    throw std::runtime_error("Incompatible vector sizes"); // src/vector.hpp:343
}

// Free functions of src/vector.hpp:333-454.
template <typename Vector>
auto inner_product(const Vector& a, const Vector& b)
{
  require_same_map(a, b);
  double r = 0;
  check(pmg_vec_inner_product(a.map()->layout(), a.array().data(), b.array().data(), &r, nullptr));
  return r;
}
template <typename Vector>
auto squared_norm(const Vector& a)
{
  double r = 0;
  check(pmg_vec_squared_norm(a.map()->layout(), a.array().data(), &r, nullptr));
  return r;
}
template <typename Vector>
auto norm(const Vector& a, Norm type = Norm::l2)
{
  double r = 0;
  check(pmg_vec_norm(a.map()->layout(), a.array().data(), type == Norm::l2 ? 0 : 1, &r, nullptr));
  return r;
}
/// r = alpha * x + y
template <typename Vector, typename S>
void axpy(Vector& r, S alpha, const Vector& x, const Vector& y)
{
  require_same_map(x, y);
  require_same_map(r, x);
  check(pmg_vec_axpy(r.map()->layout(), r.mutable_array().data(), (double)alpha, x.array().data(), y.array().data(),
                     nullptr));
}
template <typename Vector, typename S>
void scale(Vector& r, S alpha)
{
  check(pmg_vec_scale(r.map()->layout(), r.mutable_array().data(), (double)alpha, nullptr));
}
/// a = b
template <typename Vector>
void copy(Vector& a, const Vector& b)
{
  require_same_map(a, b);
  check(pmg_vec_copy(a.map()->layout(), a.mutable_array().data(), b.array().data(), nullptr));
}
/// w = x .* y
template <typename Vector>
void pointwise_mult(Vector& w, const Vector& x, const Vector& y)
{
  require_same_map(x, y);
  require_same_map(w, x);
  check(pmg_vec_pointwise_mult(w.map()->layout(), w.mutable_array().data(), x.array().data(), y.array().data(),
                               nullptr));
}
/// x_i = op(x_i) on the owned entries, any device functor (:449-454).
template <typename Vector, typename UnaryFunction>
void transform(Vector& x, UnaryFunction op)
{
  auto& v = x.thrust_vector();
  thrust::transform(thrust::device, v.begin(), v.begin() + x.map()->size_local(), v.begin(), op);
}

/// acc::MatFreeLaplacian<T> (src/laplacian.hpp:284-526).  The spans are
/// device memory owned by the caller and must outlive the operator (:500-509).
template <typename T>
class MatFreeLaplacian
{
  static_assert(std::is_same_v<T, double>, "pmg_amd: the library computes in double");

public:
  using value_type = T;

  /// Argument list of the reference (:289-297).  dphi_geometry and G_weights may be
  /// empty: both follow from the degree.  The handle is created with the first
  /// vector (or index map) the operator sees, because the reference's operator
  /// borrows the halo of its input vector (:378,425).
  MatFreeLaplacian(int degree, std::span<const T> coefficients, std::span<const std::int32_t> dofmap,
                   std::span<const T> xgeom, std::span<const std::int32_t> geometry_dofmap,
                   std::span<const T> dphi_geometry, std::span<const T> G_weights, const std::vector<int>& lcells,
                   const std::vector<int>& bcells, std::span<const std::int8_t> bc_marker, std::size_t batch_size = 0,
                   NodeOrder node_order = default_node_order)
      : _degree(degree), _kappa(coefficients), _dofmap(dofmap), _xgeom(xgeom), _geom_dofmap(geometry_dofmap),
        _dphi(dphi_geometry), _gw(G_weights), _lcells(lcells.begin(), lcells.end()),
        _bcells(bcells.begin(), bcells.end()), _bc(bc_marker), _batch_size(batch_size), _node_order(node_order)
  {
    if (degree < 1 || degree > PMG_MAX_DEGREE)
      throw std::runtime_error("Unsupported degree [mat-free operator]"); // :346
    const std::size_t N = (std::size_t)(degree + 1) * (degree + 1) * (degree + 1);
    if (dofmap.size() % N != 0 || dofmap.size() / N != coefficients.size()
        || geometry_dofmap.size() != 8 * coefficients.size())
      throw std::runtime_error("MatFreeLaplacian: array sizes do not match the cell count");
  }
  MatFreeLaplacian(const MatFreeLaplacian&) = delete;
  MatFreeLaplacian& operator=(const MatFreeLaplacian&) = delete;
  ~MatFreeLaplacian()
  {
    if (_op)
      pmg_laplacian_destroy(_op);
  }

  /// out = A in (:462-482): zeroes out, updates the ghosts of in.
  template <typename Vector>
  void operator()(Vector& in, Vector& out)
  {
    require_same_map(in, out);
    check(pmg_laplacian_apply(handle(in.map()), in.mutable_array().data(), out.mutable_array().data(), nullptr));
  }
  template <typename Vector>
  void get_diag_inverse(Vector& diag_inv) // :484-489
  {
    check(pmg_laplacian_get_diag_inverse(handle(diag_inv.map()), diag_inv.mutable_array().data(), nullptr));
  }
  template <typename Vector>
  void set_diag_inverse(const Vector& diag_inv) // :491-495
  {
    check(pmg_laplacian_set_diag_inverse(handle(diag_inv.map()), diag_inv.array().data(), nullptr));
  }
  /// Matrix-free inverse diagonal, replaces examples/pmg/main.cpp:274-279 (no CSR).
  void compute_diag_inverse(const std::shared_ptr<const IndexMap>& map)
  {
    check(pmg_laplacian_compute_diag_inverse(handle(map), nullptr));
  }
  /// b = GLL-collocated load vector of the nodal source f, BC rows zeroed
  /// (assemble_vector + set_bc, examples/pmg/main.cpp:289-300).
  template <typename Vector>
  void assemble_rhs(const Vector& f, Vector& b)
  {
    require_same_map(f, b);
    check(pmg_laplacian_assemble_rhs(handle(f.map()), f.array().data(), b.mutable_array().data(), nullptr));
  }
  int degree() const { return _degree; }

  pmg_laplacian handle(const std::shared_ptr<const IndexMap>& map)
  {
    if (!_op)
    {
      if ((std::size_t)map->size_local() + map->num_ghosts() != _bc.size())
        throw std::runtime_error("MatFreeLaplacian: vector size does not match the bc marker"); // cf. :479
      // dofmap, dphi_geometry and G_weights are indexed in the caller's cell-local node order (the library keeps
      // an ascending copy of the dofmap); without the two tables the library builds its own
      const bool tables = !_dphi.empty() && !_gw.empty();
      check(pmg_laplacian_create_ordered(
          &_op, map->layout(), _degree, (std::int32_t)_kappa.size(), _kappa.data(), _dofmap.data(), _xgeom.data(),
          (std::int32_t)(_xgeom.size() / 3), _geom_dofmap.data(), tables ? _dphi.data() : nullptr,
          tables ? _gw.data() : nullptr, _lcells.data(), (std::int32_t)_lcells.size(), _bcells.data(),
          (std::int32_t)_bcells.size(), _bc.data(), static_cast<int>(_node_order), nullptr, nullptr));
      if (_batch_size != 0) // :383-396: the geometry tensor is not kept, it is recomputed batch by batch
        check(pmg_laplacian_set_geometry_batch(_op, (long long)_batch_size));
      _map = map;
    }
    else if (map->layout() != _map->layout())
      throw std::runtime_error("MatFreeLaplacian: vector lives on a different index map");
    return _op;
  }

private:
  int _degree;
  std::span<const T> _kappa;
  std::span<const std::int32_t> _dofmap;
  std::span<const T> _xgeom;
  std::span<const std::int32_t> _geom_dofmap;
  std::span<const T> _dphi, _gw;
  std::vector<std::int32_t> _lcells, _bcells;
  std::span<const std::int8_t> _bc;
  std::size_t _batch_size;
  NodeOrder _node_order;
  std::shared_ptr<const IndexMap> _map;
  pmg_laplacian _op = nullptr;
};

/// acc::Chebyshev<Vector> (src/chebyshev.hpp:19-106).
template <typename Vector>
class Chebyshev
{
  using T = typename Vector::value_type;

public:
  Chebyshev(std::shared_ptr<const IndexMap> map, int /*bs*/, std::array<T, 2> eig_range) : _map(std::move(map))
  {
    check(pmg_chebyshev_create(&_s, _map->layout(), eig_range[0], eig_range[1]));
  }
  Chebyshev(const Chebyshev&) = delete;
  Chebyshev& operator=(const Chebyshev&) = delete;
  ~Chebyshev() { pmg_chebyshev_destroy(_s); }
  void set_max_iterations(int n) { check(pmg_chebyshev_set_max_iterations(_s, n)); } // :40
  template <typename Operator>
  void solve(Operator& A, Vector& x, const Vector& b, bool /*verbose*/ = false) // :46-91
  {
    check(pmg_chebyshev_solve(_s, A.handle(x.map()), x.mutable_array().data(), b.array().data(), nullptr));
  }
  pmg_chebyshev handle() const { return _s; }

private:
  std::shared_ptr<const IndexMap> _map;
  pmg_chebyshev _s = nullptr;
};

/// acc::CGSolver<Vector> (src/cg.hpp:93-250).
template <typename Vector>
class CGSolver
{
  using T = typename Vector::value_type;

public:
  CGSolver(std::shared_ptr<const IndexMap> map, int /*bs*/) : _map(std::move(map))
  {
    check(pmg_cg_create(&_s, _map->layout()));
  }
  CGSolver(const CGSolver&) = delete;
  CGSolver& operator=(const CGSolver&) = delete;
  ~CGSolver() { pmg_cg_destroy(_s); }
  void set_max_iterations(int n) { check(pmg_cg_set_max_iterations(_s, n)); } // :110
  void set_tolerance(double rtol) { check(pmg_cg_set_tolerance(_s, rtol)); }  // :113
  void store_coefficients(bool flag) { check(pmg_cg_store_coefficients(_s, flag ? 1 : 0)); } // :116
  /// Polak-Ribiere beta for a V-cycle preconditioner with a Krylov coarse solver (not in the reference).
  void set_flexible(bool flag) { check(pmg_cg_set_flexible(_s, flag ? 1 : 0)); }
  /// Jacobi-preconditioned CG (:147-222); returns the iteration count.
  template <typename Operator>
  int solve(Operator& A, Vector& x, const Vector& b, bool /*verbose*/ = false)
  {
    int its = 0;
    check(pmg_cg_solve(_s, A.handle(x.map()), x.mutable_array().data(), b.array().data(), nullptr, &its, nullptr));
    return its;
  }
  /// CG preconditioned by one V-cycle per iteration (BASELINE config 2; not in the reference).
  template <typename Operator, typename MG>
    requires requires(MG& m) { m.handle(); }
  int solve(Operator& A, Vector& x, const Vector& b, MG& precond, bool /*verbose*/ = false)
  {
    int its = 0;
    check(pmg_cg_solve(_s, A.handle(x.map()), x.mutable_array().data(), b.array().data(), precond.handle(), &its,
                       nullptr));
    return its;
  }
  std::vector<T> alphas() const { return coefficients().first; } // :118
  std::vector<T> betas() const { return coefficients().second; } // :119
  /// Lanczos tridiagonal from the stored coefficients + QL implicit (:121-142), ascending.
  std::vector<T> compute_eigenvalues() const
  {
    std::vector<T> e(4096);
    const int n = pmg_cg_compute_eigenvalues(_s, e.data(), (int)e.size());
    if (n < 0)
      throw std::runtime_error(pmg_last_error()); // :125,138
    e.resize(n);
    return e;
  }
  T residual() const // :144
  {
    double r = 0;
    check(pmg_cg_residual(_s, &r));
    return r;
  }
  pmg_cg handle() const { return _s; }

private:
  std::pair<std::vector<T>, std::vector<T>> coefficients() const
  {
    std::vector<T> a(4096), b(4096);
    const int n = pmg_cg_coefficients(_s, a.data(), b.data(), (int)a.size());
    if (n < 0)
      throw std::runtime_error(pmg_last_error());
    a.resize(n);
    b.resize(n);
    return {a, b};
  }
  std::shared_ptr<const IndexMap> _map;
  pmg_cg _s = nullptr;
};
/// The library's algebraic multigrid in the slot of the reference's CoarseSolverType<T>
/// (src/amg.hpp:9-118: PETSc KSPCG + hypre BoomerAMG on the degree-1 level): `solve(x, y)` solves
/// A x = y from a zero initial guess.  Constructed from the degree-1 operator.
template <typename Vector>
class AmgSolver
{
public:
  template <typename Operator>
  AmgSolver(Operator& coarse_operator, const std::shared_ptr<const IndexMap>& map)
  {
    check(pmg_amg_create(&_a, coarse_operator.handle(map), nullptr));
  }
  /// Several ranks (`local_to_global` as dolfinx's IndexMap gives it: owned dofs, then ghosts).  By default the first
  /// coarsening is done per rank and only level 1 is gathered (pmg_amg_create_distributed: no rank holds the global
  /// degree-1 matrix); gather_level0 = true gathers the global matrix on every rank instead (the single-rank
  /// hierarchy on every rank, pmg_amg_create_replicated).  Either way one all-reduce of a level-1 vector per cycle.
  template <typename Operator>
  AmgSolver(Operator& coarse_operator, const std::shared_ptr<const IndexMap>& map,
            std::span<const std::int64_t> local_to_global, std::int64_t size_global, bool gather_level0 = false)
  {
    if ((std::int64_t)local_to_global.size() != (std::int64_t)map->size_local() + map->num_ghosts())
      throw std::runtime_error("AmgSolver: local_to_global must cover owned and ghost dofs");
    if (gather_level0)
      check(pmg_amg_create_replicated(&_a, coarse_operator.handle(map), local_to_global.data(), size_global, nullptr));
    else
      check(pmg_amg_create_distributed(&_a, coarse_operator.handle(map), local_to_global.data(), size_global, nullptr));
  }
  AmgSolver(const AmgSolver&) = delete;
  AmgSolver& operator=(const AmgSolver&) = delete;
  ~AmgSolver() { pmg_amg_destroy(_a); }
  /// CG preconditioned by one AMG cycle (the default: 60 iterations, rtol 1e-5, src/amg.hpp:36-40).
  void set_krylov(int max_iter, double rtol) { check(pmg_amg_set_krylov(_a, max_iter, rtol)); }
  /// n AMG cycles from a zero initial guess: a fixed linear operator, no host synchronisation.
  void set_cycles(int n) { check(pmg_amg_set_cycles(_a, n)); }
  void set_smoother_iterations(int k) { check(pmg_amg_set_smoother_iterations(_a, k)); }
  void solve(Vector& x, Vector& y) // src/amg.hpp:67-68
  {
    check(pmg_amg_solve(_a, x.mutable_array().data(), y.array().data(), &_its, nullptr));
  }
  int iterations() const { return _its; }
  int num_levels() const { return pmg_amg_num_levels(_a); }
  pmg_amg handle() const { return _a; }

private:
  pmg_amg _a = nullptr;
  int _its = 0;
};
} // namespace acc

/// Interpolator<T> (src/interpolate.hpp:93-329).  The reference passes the two basix
/// elements (:104-107); on this path they are GLL tensor-product Lagrange elements, so
/// their degree() says everything -- basix::FiniteElement<T> has that member,
/// pmg_amd::LagrangeElement stands in for it without basix.  Dofmap spans are device memory
/// (caller-owned), the cell lists host memory.
template <typename T>
class Interpolator
{
  static_assert(std::is_same_v<T, double>, "pmg_amd: the library computes in double");

public:
  template <typename Element>
    requires requires(const Element& e) { e.degree(); }
  Interpolator(const Element& Q1_element, const Element& Q2_element, std::span<const std::int32_t> Q1_dofmap,
               std::span<const std::int32_t> Q2_dofmap, std::span<const std::int32_t> l_cells,
               std::span<const std::int32_t> b_cells, NodeOrder node_order = default_node_order)
      : Interpolator((int)Q1_element.degree(), (int)Q2_element.degree(), Q1_dofmap, Q2_dofmap, l_cells, b_cells,
                     node_order)
  {
  }
  /// Both dofmaps in `node_order` (the reference's come from two basix tensor-product spaces, :104-107; its
  /// operator from basix::compute_interpolation_operator in the same order, :118).
  Interpolator(int degree_coarse, int degree_fine, std::span<const std::int32_t> dofmap_coarse,
               std::span<const std::int32_t> dofmap_fine, std::span<const std::int32_t> lcells,
               std::span<const std::int32_t> bcells, NodeOrder node_order = default_node_order)
      : _pc(degree_coarse), _pf(degree_fine), _dmc(dofmap_coarse), _dmf(dofmap_fine),
        _lcells(lcells.begin(), lcells.end()), _bcells(bcells.begin(), bcells.end()), _node_order(node_order)
  {
    if (degree_coarse < 1 || degree_fine <= degree_coarse || degree_fine > PMG_MAX_DEGREE)
      throw std::runtime_error("Interpolator: need 1 <= coarse degree < fine degree <= 8");
  }
  Interpolator(const Interpolator&) = delete;
  Interpolator& operator=(const Interpolator&) = delete;
  ~Interpolator()
  {
    if (_ip)
      pmg_interpolator_destroy(_ip);
  }
  /// Prolongation (:186-239).
  template <typename Vector>
  void interpolate(Vector& Q1_vector, Vector& Q2_vector)
  {
    check(pmg_interpolator_interpolate(handle(Q1_vector.map(), Q2_vector.map(), nullptr),
                                       Q1_vector.mutable_array().data(), Q2_vector.mutable_array().data(), nullptr));
  }
  /// Restriction (:246-303).
  template <typename Vector>
  void reverse_interpolate(Vector& Q2_vector, Vector& Q1_vector)
  {
    check(pmg_interpolator_reverse_interpolate(handle(Q1_vector.map(), Q2_vector.map(), nullptr),
                                               Q2_vector.mutable_array().data(), Q1_vector.mutable_array().data(),
                                               nullptr));
  }
  /// Created on first use; if the fine-level operator is known by then (the V-cycle
  /// passes it) the transfers share its cell patches.
  pmg_interpolator handle(const std::shared_ptr<const IndexMap>& coarse, const std::shared_ptr<const IndexMap>& fine,
                          pmg_laplacian fine_operator)
  {
    if (!_ip)
    {
      const std::size_t Nf = (std::size_t)(_pf + 1) * (_pf + 1) * (_pf + 1);
      check(pmg_interpolator_create_ordered(
          &_ip, coarse->layout(), fine->layout(), _pc, _pf, (std::int32_t)(_dmf.size() / Nf), _dmc.data(), _dmf.data(),
          _lcells.data(), (std::int32_t)_lcells.size(), _bcells.data(), (std::int32_t)_bcells.size(), fine_operator,
          static_cast<int>(_node_order), nullptr, nullptr, nullptr));
    }
    return _ip;
  }

private:
  int _pc, _pf;
  std::span<const std::int32_t> _dmc, _dmf;
  std::vector<std::int32_t> _lcells, _bcells;
  NodeOrder _node_order;
  pmg_interpolator _ip = nullptr;
};

namespace acc
{
/// acc::MultigridPreconditioner<Vector, Operator, Solver, CoarseSolver, Interpolator>
/// (src/pmg.hpp:13-184), template parameters in the reference's order.  Levels coarse -> fine.
/// CoarseSolver is any type with `solve(Vector& x, Vector& b)` (the reference's
/// CoarseSolverType<T>, src/amg.hpp:67): it is reached through the C ABI's coarse-solver
/// callback; the library's own solvers (CGSolver<Vector>, AmgSolver<Vector>) are wired natively.
template <typename Vector, typename Operator, typename Solver, typename CoarseSolver, typename Interpolator>
class MultigridPreconditioner
{
  using T = typename Vector::value_type;

public:
  MultigridPreconditioner(std::vector<std::shared_ptr<const IndexMap>> maps, int /*bs*/,
                          std::span<const std::int8_t> bc_marker)
      : _maps(std::move(maps))
  {
    std::vector<pmg_layout> layouts;
    for (auto& m : _maps)
      layouts.push_back(m->layout());
    check(pmg_multigrid_create(&_mg, (int)layouts.size(), layouts.data(), bc_marker.data()));
  }
  MultigridPreconditioner(const MultigridPreconditioner&) = delete;
  MultigridPreconditioner& operator=(const MultigridPreconditioner&) = delete;
  ~MultigridPreconditioner() { pmg_multigrid_destroy(_mg); }

  void set_solvers(std::vector<std::shared_ptr<Solver>>& solvers) // :44
  {
    _solvers = solvers;
    _wired = false;
  }
  /// :46; nullptr = the level-0 smoother solves the coarsest level (:106-109).
  void set_coarse_solver(std::shared_ptr<CoarseSolver> solver)
  {
    _coarse_solver = std::move(solver);
    if (!_coarse_solver)
    {
      check(pmg_multigrid_set_coarse_callback(_mg, nullptr, nullptr));
      check(pmg_multigrid_set_coarse_solver(_mg, nullptr));
      return;
    }
    if constexpr (requires(CoarseSolver& c) { { c.handle() } -> std::same_as<pmg_cg>; })
      check(pmg_multigrid_set_coarse_solver(_mg, _coarse_solver->handle()));
    else if constexpr (requires(CoarseSolver& c) { { c.handle() } -> std::same_as<pmg_amg>; })
      check(pmg_multigrid_set_coarse_amg(_mg, _coarse_solver->handle()));
    else
    {
      // any solve(Vector&, Vector&): the cycle hands over its coarsest-level arrays, the bridge
      // moves them through two vectors on the coarsest map
      _cx = std::make_unique<Vector>(_maps.front(), 1);
      _cb = std::make_unique<Vector>(_maps.front(), 1);
      check(pmg_multigrid_set_coarse_callback(_mg, &MultigridPreconditioner::coarse_bridge, this));
    }
  }
  void set_operators(std::vector<std::shared_ptr<Operator>>& operators) // :48
  {
    _operators = operators;
    _wired = false;
  }
  void set_interpolators(std::vector<std::shared_ptr<Interpolator>>& interpolators) // :50-53
  {
    _matfree_interpolation = interpolators;
    _wired = false;
  }
  /// x = rhs, y = initial guess in / result out (:56-155).  With verbose the final
  /// residual norm is computed and returned (the reference prints it, :147-150); else 0.
  T apply(const Vector& x, Vector& y, bool verbose = false)
  {
    wire();
    double rnorm = 0;
    check(pmg_multigrid_apply(_mg, x.array().data(), y.mutable_array().data(), verbose ? &rnorm : nullptr, nullptr));
    return rnorm;
  }
  pmg_multigrid handle()
  {
    wire();
    return _mg;
  }

private:
  static int coarse_bridge(void* user, double* x, double* b, pmg_stream stream)
  {
    auto* self = static_cast<MultigridPreconditioner*>(user);
    try
    {
      const std::size_t n = self->_cx->array().size();
      hipStream_t s = (hipStream_t)stream;
      hip_check(hipMemcpyAsync(self->_cb->mutable_array().data(), b, sizeof(T) * n, hipMemcpyDeviceToDevice, s), "D2D");
      hip_check(hipMemcpyAsync(self->_cx->mutable_array().data(), x, sizeof(T) * n, hipMemcpyDeviceToDevice, s), "D2D");
      hip_check(hipStreamSynchronize(s), "sync"); // the caller's solver may work on any stream
      self->_coarse_solver->solve(*self->_cx, *self->_cb);
      hip_check(hipDeviceSynchronize(), "sync");
      hip_check(hipMemcpyAsync(x, self->_cx->array().data(), sizeof(T) * n, hipMemcpyDeviceToDevice, s), "D2D");
      return 0;
    }
    catch (const std::exception& e)
    {
      self->_coarse_error = e.what();
      return 1;
    }
  }
  void wire()
  {
    if (_wired)
      return;
    const std::size_t L = _maps.size();
    if (_operators.size() != L || _solvers.size() != L || _matfree_interpolation.size() + 1 != L)
      throw std::runtime_error("MultigridPreconditioner: need one operator and solver per level and one "
                               "interpolator per pair of levels");
    std::vector<pmg_laplacian> ops;
    std::vector<pmg_chebyshev> sm;
    std::vector<pmg_interpolator> ip;
    for (std::size_t i = 0; i < L; ++i)
    {
      ops.push_back(_operators[i]->handle(_maps[i]));
      sm.push_back(_solvers[i]->handle());
    }
    for (std::size_t i = 0; i + 1 < L; ++i)
      ip.push_back(_matfree_interpolation[i]->handle(_maps[i], _maps[i + 1], ops[i + 1]));
    check(pmg_multigrid_set_operators(_mg, ops.data()));
    check(pmg_multigrid_set_solvers(_mg, sm.data()));
    check(pmg_multigrid_set_interpolators(_mg, ip.data()));
    _wired = true;
  }
  std::vector<std::shared_ptr<const IndexMap>> _maps;
  std::vector<std::shared_ptr<Operator>> _operators;
  std::vector<std::shared_ptr<Solver>> _solvers;
  std::vector<std::shared_ptr<Interpolator>> _matfree_interpolation;
  std::shared_ptr<CoarseSolver> _coarse_solver;
  std::unique_ptr<Vector> _cx, _cb;
  std::string _coarse_error;
  pmg_multigrid _mg = nullptr;
  bool _wired = false;
};
} // namespace acc
} // namespace pmg_amd

#ifdef PMG_AMD_DOLFINX_NAMESPACE
// The reference's own namespaces, for translation units written against its headers
// (`using namespace dolfinx;`, examples/pmg/main.cpp:29).
namespace dolfinx
{
namespace common
{
using IndexMap = pmg_amd::IndexMap;
}
namespace acc
{
using namespace pmg_amd::acc;
}
namespace la
{
using Norm = pmg_amd::acc::Norm; // dolfinx::la::Norm::l2 / linf (src/vector.hpp:360-389)
}
} // namespace dolfinx
using pmg_amd::Interpolator;
#endif
