// pmg_amd.hpp -- C++ adapter over the C ABI of pmg_amd.h.
//
// Re-exports the duck-typed concepts the reference's drivers are written
// against (Wells-Group/pmg-dolfinx @ 2024_08_07):
//
//   acc::Vector                    src/vector.hpp:74-325 (+ free functions :333-454)
//   acc::MatFreeLaplacian          src/laplacian.hpp:284-526
//   acc::Chebyshev<Vector>         src/chebyshev.hpp:19-106
//   acc::CGSolver<Vector>          src/cg.hpp:93-250
//   Interpolator                   src/interpolate.hpp:93-329
//   acc::MultigridPreconditioner   src/pmg.hpp:16-184
//
// with the same member names, argument meaning and error behaviour
// (std::runtime_error), so that examples/pmg/main.cpp:270-365 and
// examples/mat_free/main.cpp:236-288 read the same on top of it (see
// examples/ in this repository).  dolfinx is not a dependency: the one dolfinx
// type on the path, common::IndexMap (+ its Scatterer), is replaced by
// pmg_amd::IndexMap below, which carries the same information flattened to
// arrays; INTEGRATION.md shows the constructor a maintainer adds to build it
// from a dolfinx IndexMap.  Header-only, C++20 (std::span, like the reference),
// needs the HIP runtime headers; link with libpmg_amd.so and amdhip64.
#pragma once

#include "pmg_amd.h"

#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>
#include <memory>
#include <span>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace pmg_amd
{
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

/// Owning device array (the reference uses thrust::device_vector for this).
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

/// Stand-in for dolfinx::common::IndexMap + common::Scatterer on this path:
/// owned/ghost sizes, the packed index lists of the halo
/// (Scatterer::local_indices() / remote_indices(), src/vector.hpp:91-92) and the
/// exchange itself as callbacks (see pmg_amd.h).  Owns the device copies and the
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
  IndexMap(const IndexMap&) = delete;
  IndexMap& operator=(const IndexMap&) = delete;
  ~IndexMap() { pmg_layout_destroy(_layout); }

  std::int32_t size_local() const { return _size_local; }
  std::int32_t num_ghosts() const { return _num_ghosts; }
  pmg_layout layout() const { return _layout; }
  /// The packed send / receive staging buffers (device), for an exchange callback.
  std::span<double> send_buffer() { return _send.span(); }
  std::span<double> recv_buffer() { return _recv.span(); }

private:
  std::int32_t _size_local, _num_ghosts;
  device_array<std::int32_t> _send_idx, _recv_idx;
  device_array<double> _send, _recv;
  pmg_layout _layout = nullptr;
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

namespace acc
{
enum class Norm
{
  l2,
  linf
};

/// acc::Vector<double, Device::HIP> (src/vector.hpp:74-325), block size 1.
class Vector
{
public:
  using value_type = double;

  Vector(std::shared_ptr<const IndexMap> map, int bs) : _map(std::move(map))
  {
    if (bs != 1)
      throw std::runtime_error("Vector: block size must be 1");
    _x.resize((std::size_t)_map->size_local() + _map->num_ghosts()); // zero-initialised, :88
  }
  Vector(Vector&&) = default;
  Vector& operator=(Vector&&) = default;

  void set(double v) { check(pmg_vec_set(_map->layout(), _x.data(), v, nullptr)); } // :109-115
  /// Owned part from a host array (:120-128).
  void copy_from_host(std::span<const double> host)
  {
    if ((std::int32_t)host.size() < _map->size_local())
      throw std::runtime_error("copy_from_host: source shorter than size_local");
    hip_check(hipMemcpy(_x.data(), host.data(), sizeof(double) * _map->size_local(), hipMemcpyHostToDevice),
              "hipMemcpy H2D");
  }
  std::shared_ptr<const IndexMap> map() const { return _map; }
  constexpr int bs() const { return 1; }
  std::span<const double> array() const { return _x.span(); }
  std::span<double> mutable_array() { return _x.span(); }
  /// Host copy of owned + ghost entries (data_copy(), :296-302).
  std::vector<double> data_copy() const { return _x.to_host(); }

  void scatter_fwd_begin() { check(pmg_scatter_fwd_begin(_map->layout(), _x.data(), nullptr)); } // :186-207
  void scatter_fwd_end() { check(pmg_scatter_fwd_end(_map->layout(), _x.data(), nullptr)); }     // :213-238
  void scatter_fwd()
  {
    scatter_fwd_begin();
    scatter_fwd_end();
  }
  void scatter_rev_begin() { check(pmg_scatter_rev_begin(_map->layout(), _x.data(), nullptr)); } // :249-263
  void scatter_rev_end() { check(pmg_scatter_rev_end(_map->layout(), _x.data(), nullptr)); }     // :271-286

private:
  std::shared_ptr<const IndexMap> _map;
  device_array<double> _x;
};

inline void require_same_map(const Vector& a, const Vector& b)
{
  if (a.map()->size_local() != b.map()->size_local())
    throw std::runtime_error("Incompatible vector sizes"); // src/vector.hpp:343
}

// Free functions of src/vector.hpp:333-454.
inline double inner_product(const Vector& a, const Vector& b)
{
  require_same_map(a, b);
  double r = 0;
  check(pmg_vec_inner_product(a.map()->layout(), a.array().data(), b.array().data(), &r, nullptr));
  return r;
}
inline double squared_norm(const Vector& a)
{
  double r = 0;
  check(pmg_vec_squared_norm(a.map()->layout(), a.array().data(), &r, nullptr));
  return r;
}
inline double norm(const Vector& a, Norm type = Norm::l2)
{
  double r = 0;
  check(pmg_vec_norm(a.map()->layout(), a.array().data(), type == Norm::l2 ? 0 : 1, &r, nullptr));
  return r;
}
/// r = alpha * x + y
inline void axpy(Vector& r, double alpha, const Vector& x, const Vector& y)
{
  require_same_map(x, y);
  check(pmg_vec_axpy(r.map()->layout(), r.mutable_array().data(), alpha, x.array().data(), y.array().data(),
                     nullptr));
}
inline void scale(Vector& r, double alpha)
{
  check(pmg_vec_scale(r.map()->layout(), r.mutable_array().data(), alpha, nullptr));
}
/// b = a
inline void copy(Vector& b, const Vector& a)
{
  require_same_map(a, b);
  check(pmg_vec_copy(b.map()->layout(), b.mutable_array().data(), a.array().data(), nullptr));
}
/// w = x .* y
inline void pointwise_mult(Vector& w, const Vector& x, const Vector& y)
{
  require_same_map(x, y);
  check(pmg_vec_pointwise_mult(w.map()->layout(), w.mutable_array().data(), x.array().data(), y.array().data(),
                               nullptr));
}

/// acc::MatFreeLaplacian<double> (src/laplacian.hpp:284-526).  The spans are
/// device memory owned by the caller and must outlive the operator (:500-509).
class MatFreeLaplacian
{
public:
  using value_type = double;

  /// Argument list of the reference (:289-297).  dphi_geometry and G_weights may be
  /// empty: both follow from the degree.  The handle is created with the first
  /// vector (or index map) the operator sees, because the reference's operator
  /// borrows the halo of its input vector (:378,425).
  MatFreeLaplacian(int degree, std::span<const double> coefficients, std::span<const std::int32_t> dofmap,
                   std::span<const double> xgeom, std::span<const std::int32_t> geometry_dofmap,
                   std::span<const double> dphi_geometry, std::span<const double> G_weights,
                   const std::vector<int>& lcells, const std::vector<int>& bcells,
                   std::span<const std::int8_t> bc_marker, std::size_t batch_size = 0)
      : _degree(degree), _kappa(coefficients), _dofmap(dofmap), _xgeom(xgeom), _geom_dofmap(geometry_dofmap),
        _dphi(dphi_geometry), _gw(G_weights), _lcells(lcells.begin(), lcells.end()),
        _bcells(bcells.begin(), bcells.end()), _bc(bc_marker)
  {
    if (degree < 1 || degree > PMG_MAX_DEGREE)
      throw std::runtime_error("Unsupported degree [mat-free operator]"); // :346
    if (batch_size != 0)
      throw std::runtime_error("MatFreeLaplacian: geometry batching is not supported (G stays resident)");
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
  void operator()(Vector& in, Vector& out)
  {
    check(pmg_laplacian_apply(handle(in.map()), in.mutable_array().data(), out.mutable_array().data(), nullptr));
  }
  void get_diag_inverse(Vector& diag_inv) // :484-489
  {
    check(pmg_laplacian_get_diag_inverse(handle(diag_inv.map()), diag_inv.mutable_array().data(), nullptr));
  }
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
  void assemble_rhs(const Vector& f, Vector& b)
  {
    check(pmg_laplacian_assemble_rhs(handle(f.map()), f.array().data(), b.mutable_array().data(), nullptr));
  }
  int degree() const { return _degree; }

  pmg_laplacian handle(const std::shared_ptr<const IndexMap>& map)
  {
    if (!_op)
    {
      if ((std::size_t)map->size_local() + map->num_ghosts() != _bc.size())
        throw std::runtime_error("MatFreeLaplacian: vector size does not match the bc marker"); // cf. :479
      const bool tables = !_dphi.empty() && !_gw.empty();
      if (tables)
        check(pmg_laplacian_create_with_tables(
            &_op, map->layout(), _degree, (std::int32_t)_kappa.size(), _kappa.data(), _dofmap.data(), _xgeom.data(),
            (std::int32_t)(_xgeom.size() / 3), _geom_dofmap.data(), _dphi.data(), _gw.data(), _lcells.data(),
            (std::int32_t)_lcells.size(), _bcells.data(), (std::int32_t)_bcells.size(), _bc.data(), nullptr));
      else
        check(pmg_laplacian_create(&_op, map->layout(), _degree, (std::int32_t)_kappa.size(), _kappa.data(),
                                   _dofmap.data(), _xgeom.data(), (std::int32_t)(_xgeom.size() / 3),
                                   _geom_dofmap.data(), _lcells.data(), (std::int32_t)_lcells.size(),
                                   _bcells.data(), (std::int32_t)_bcells.size(), _bc.data(), nullptr));
      _map = map;
    }
    else if (map.get() != _map.get() && map->size_local() != _map->size_local())
      throw std::runtime_error("MatFreeLaplacian: vector lives on a different index map");
    return _op;
  }

private:
  int _degree;
  std::span<const double> _kappa;
  std::span<const std::int32_t> _dofmap;
  std::span<const double> _xgeom;
  std::span<const std::int32_t> _geom_dofmap;
  std::span<const double> _dphi, _gw;
  std::vector<std::int32_t> _lcells, _bcells;
  std::span<const std::int8_t> _bc;
  std::shared_ptr<const IndexMap> _map;
  pmg_laplacian _op = nullptr;
};

/// acc::Chebyshev<Vector> (src/chebyshev.hpp:19-106).
template <typename V = Vector>
class Chebyshev
{
public:
  Chebyshev(std::shared_ptr<const IndexMap> map, int /*bs*/, std::array<double, 2> eig_range) : _map(std::move(map))
  {
    check(pmg_chebyshev_create(&_s, _map->layout(), eig_range[0], eig_range[1]));
  }
  Chebyshev(const Chebyshev&) = delete;
  Chebyshev& operator=(const Chebyshev&) = delete;
  ~Chebyshev() { pmg_chebyshev_destroy(_s); }
  void set_max_iterations(int n) { check(pmg_chebyshev_set_max_iterations(_s, n)); } // :40
  template <typename Operator>
  void solve(Operator& A, V& x, const V& b, bool /*verbose*/ = false) // :46-91
  {
    check(pmg_chebyshev_solve(_s, A.handle(x.map()), x.mutable_array().data(), b.array().data(), nullptr));
  }
  pmg_chebyshev handle() const { return _s; }

private:
  std::shared_ptr<const IndexMap> _map;
  pmg_chebyshev _s = nullptr;
};

template <typename V, typename Operator, typename Interp, typename Solver>
class MultigridPreconditioner;

/// acc::CGSolver<Vector> (src/cg.hpp:93-250).
template <typename V = Vector>
class CGSolver
{
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
  int solve(Operator& A, V& x, const V& b, bool /*verbose*/ = false)
  {
    int its = 0;
    check(pmg_cg_solve(_s, A.handle(x.map()), x.mutable_array().data(), b.array().data(), nullptr, &its, nullptr));
    return its;
  }
  /// CG preconditioned by one V-cycle per iteration (BASELINE config 2; not in the reference).
  template <typename Operator, typename MG>
  int solve(Operator& A, V& x, const V& b, MG& precond, bool /*verbose*/ = false)
  {
    int its = 0;
    check(pmg_cg_solve(_s, A.handle(x.map()), x.mutable_array().data(), b.array().data(), precond.handle(), &its,
                       nullptr));
    return its;
  }
  std::vector<double> alphas() const { return coefficients().first; } // :118
  std::vector<double> betas() const { return coefficients().second; } // :119
  /// Lanczos tridiagonal from the stored coefficients + QL implicit (:121-142), ascending.
  std::vector<double> compute_eigenvalues() const
  {
    std::vector<double> e(4096);
    const int n = pmg_cg_compute_eigenvalues(_s, e.data(), (int)e.size());
    if (n < 0)
      throw std::runtime_error(pmg_last_error()); // :125,138
    e.resize(n);
    return e;
  }
  double residual() const
  {
    double r = 0;
    check(pmg_cg_residual(_s, &r));
    return r;
  }
  pmg_cg handle() const { return _s; }

private:
  std::pair<std::vector<double>, std::vector<double>> coefficients() const
  {
    std::vector<double> a(4096), b(4096);
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
} // namespace acc

/// Interpolator<double> (src/interpolate.hpp:93-329).  The reference passes the two
/// basix elements; on this path they are GLL tensor-product Lagrange elements, so
/// the degrees say everything.  Dofmap spans are device memory (caller-owned),
/// the cell lists host memory.
class Interpolator
{
public:
  Interpolator(int degree_coarse, int degree_fine, std::span<const std::int32_t> dofmap_coarse,
               std::span<const std::int32_t> dofmap_fine, std::span<const std::int32_t> lcells,
               std::span<const std::int32_t> bcells)
      : _pc(degree_coarse), _pf(degree_fine), _dmc(dofmap_coarse), _dmf(dofmap_fine),
        _lcells(lcells.begin(), lcells.end()), _bcells(bcells.begin(), bcells.end())
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
  void interpolate(acc::Vector& coarse, acc::Vector& fine)
  {
    check(pmg_interpolator_interpolate(handle(coarse.map(), fine.map(), nullptr), coarse.mutable_array().data(),
                                       fine.mutable_array().data(), nullptr));
  }
  /// Restriction (:246-303).
  void reverse_interpolate(acc::Vector& fine, acc::Vector& coarse)
  {
    check(pmg_interpolator_reverse_interpolate(handle(coarse.map(), fine.map(), nullptr),
                                               fine.mutable_array().data(), coarse.mutable_array().data(), nullptr));
  }
  /// Created on first use; if the fine-level operator is known by then (the V-cycle
  /// passes it) the transfers share its cell patches.
  pmg_interpolator handle(const std::shared_ptr<const IndexMap>& coarse, const std::shared_ptr<const IndexMap>& fine,
                          pmg_laplacian fine_operator)
  {
    if (!_ip)
    {
      const std::size_t Nf = (std::size_t)(_pf + 1) * (_pf + 1) * (_pf + 1);
      check(pmg_interpolator_create_with_operator(
          &_ip, coarse->layout(), fine->layout(), _pc, _pf, (std::int32_t)(_dmf.size() / Nf), _dmc.data(), _dmf.data(),
          _lcells.data(), (std::int32_t)_lcells.size(), _bcells.data(), (std::int32_t)_bcells.size(), fine_operator,
          nullptr));
    }
    return _ip;
  }

private:
  int _pc, _pf;
  std::span<const std::int32_t> _dmc, _dmf;
  std::vector<std::int32_t> _lcells, _bcells;
  pmg_interpolator _ip = nullptr;
};

namespace acc
{
/// acc::MultigridPreconditioner (src/pmg.hpp:16-184).  Levels coarse -> fine.
template <typename V = Vector, typename Operator = MatFreeLaplacian, typename Interp = pmg_amd::Interpolator,
          typename Solver = Chebyshev<V>>
class MultigridPreconditioner
{
public:
  MultigridPreconditioner(std::vector<std::shared_ptr<const IndexMap>> maps, int /*bs*/,
                          std::span<const std::int8_t> bc_marker_coarsest)
      : _maps(std::move(maps))
  {
    std::vector<pmg_layout> layouts;
    for (auto& m : _maps)
      layouts.push_back(m->layout());
    check(pmg_multigrid_create(&_mg, (int)layouts.size(), layouts.data(), bc_marker_coarsest.data()));
  }
  MultigridPreconditioner(const MultigridPreconditioner&) = delete;
  MultigridPreconditioner& operator=(const MultigridPreconditioner&) = delete;
  ~MultigridPreconditioner() { pmg_multigrid_destroy(_mg); }

  void set_solvers(std::vector<std::shared_ptr<Solver>>& solvers) // :44
  {
    _solvers = solvers;
    _wired = false;
  }
  /// A CGSolver on the coarsest map (its iteration cap and tolerance apply; zero initial guess,
  /// like the reference's KSP solve, src/amg.hpp:36-44) or nullptr for the smoother
  /// (src/pmg.hpp:106-109).  The hypre BoomerAMG preconditioner of the reference's coarse
  /// solver is third-party and out of scope: this CG is Jacobi-preconditioned.
  void set_coarse_solver(std::shared_ptr<CGSolver<V>> solver) // :46
  {
    _coarse = std::move(solver);
    check(pmg_multigrid_set_coarse_solver(_mg, _coarse ? _coarse->handle() : nullptr));
  }
  void set_operators(std::vector<std::shared_ptr<Operator>>& operators) // :48
  {
    _operators = operators;
    _wired = false;
  }
  void set_interpolators(std::vector<std::shared_ptr<Interp>>& interpolators) // :50-53
  {
    _interpolators = interpolators;
    _wired = false;
  }
  /// x = rhs, y = initial guess in / result out (:56-155).  With verbose the final
  /// residual norm is computed and returned (the reference prints it, :147-150); else 0.
  double apply(const V& x, V& y, bool verbose = false)
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
  void wire()
  {
    if (_wired)
      return;
    const std::size_t L = _maps.size();
    if (_operators.size() != L || _solvers.size() != L || _interpolators.size() + 1 != L)
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
      ip.push_back(_interpolators[i]->handle(_maps[i], _maps[i + 1], ops[i + 1]));
    check(pmg_multigrid_set_operators(_mg, ops.data()));
    check(pmg_multigrid_set_solvers(_mg, sm.data()));
    check(pmg_multigrid_set_interpolators(_mg, ip.data()));
    _wired = true;
  }
  std::vector<std::shared_ptr<const IndexMap>> _maps;
  std::vector<std::shared_ptr<Operator>> _operators;
  std::vector<std::shared_ptr<Solver>> _solvers;
  std::vector<std::shared_ptr<Interp>> _interpolators;
  std::shared_ptr<CGSolver<V>> _coarse;
  pmg_multigrid _mg = nullptr;
  bool _wired = false;
};
} // namespace acc
} // namespace pmg_amd
