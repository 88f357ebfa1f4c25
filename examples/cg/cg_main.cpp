// Matrix-free CG then Chebyshev driver: the MI355X counterpart of the reference's
// examples/cg/main.cpp (:41-296) over include/pmg_amd.hpp.  Degree 3 (:88), kappa = 2 (:127),
// source f = 1000 exp(-((x - 1/2)^2 + (y - 1/2)^2) / 0.02) (:135-148), Dirichlet value 1.3 on the
// whole boundary (:156-158).
//   1. Jacobi-preconditioned CG on b = 1, x0 = 0: 20 iterations, rtol 1e-6, coefficients stored
//      (:238-254); Lanczos eigenvalues, smoothing range {0.1, 1.1} * lambda_max (:256-258);
//   2. 30 Chebyshev iterations (:269-270) on the assembled load vector (GLL rule: lumped mass times
//      f, lifting b -= A g for the boundary value g, b[bc] = 1.3; :231-236) from the non-zero guess
//      x = 1, x[bc] = 1.3 (:275-280), residual norm printed per iteration (src/chebyshev.hpp:62-89).
// What stands above replaces dolfinx (mesh, function space, index map); the inverse diagonal comes
// from the matrix-free kernel instead of the assembled CSR operator of :224-229.
//   --ranks px,py,pz   one process per GPU and brick on the library's RCCL communicator
//                      (examples/pmg/run_ranks.sh launches the processes)
#define PMG_AMD_DOLFINX_NAMESPACE
// this driver generates its own mesh (examples/common/box_mesh.hpp: cell-local nodes by ascending coordinate); with
// dolfinx-generated dofmaps the macro above alone is right: the adapter then defaults to basix's order
#define PMG_AMD_DEFAULT_NODE_ORDER 0
#include "../common/box_mesh.hpp"
#include "../common/brick_partition.hpp"
#include "../common/rank_launch.hpp"
#include "pmg_amd.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace dolfinx;
using T = double; // examples/cg/main.cpp:30
using DeviceVector = dolfinx::acc::Vector<T, acc::Device::HIP>; // :161
using pmg_amd::check;
using pmg_amd::device_array;
using pmg_amd::hip_check;

namespace
{
struct Options : examples::RankOptions
{
  int n = 16, order = 3, cg_its = 20, cheb_its = 30;
};

void upload(DeviceVector& v, const std::vector<T>& h)
{
  hip_check(hipMemcpy(v.mutable_array().data(), h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice), "H2D");
}

void solve(const Options& o)
{
  const bool root = o.rank == 0;
  const int order = o.order, nd = order + 1;
  const T kappa = 2.0, g_bc = 1.3;
  std::shared_ptr<const pmg_amd::Communicator> comm = examples::bootstrap(o);

  examples::BrickPartition mesh(o.n, o.ranks, o.rank);
  std::vector<double> gll(nd), w(nd);
  check(pmg_gll_table(nd, gll.data(), w.data()));
  const examples::PartitionLevel lv = mesh.level(order, gll);
  std::shared_ptr<const common::IndexMap> map
      = comm ? std::make_shared<const common::IndexMap>(lv.size_local, lv.num_ghosts, lv.send_indices, lv.recv_indices,
                                                         comm, lv.neighbors, lv.send_counts, lv.recv_counts, o.halo())
             : std::make_shared<const common::IndexMap>(lv.size_local, lv.num_ghosts);
  if (root)
  {
    std::cout << "-----------------------------------\n";
    std::cout << "Number of ranks : " << o.size() << "\n";
    std::cout << "Number of cells-global : " << (long long)o.n * o.n * o.n << "\n";
    std::cout << "Number of dofs-global : " << (long long)mesh.global_ndofs(order) << "\n";
    std::cout << "-----------------------------------\n";
  }

  device_array<T> constants_d(std::vector<T>(mesh.ncells, kappa)), xgeom_d(mesh.xgeom);
  device_array<std::int32_t> dofmap_d(lv.dofmap), xdofmap_d(mesh.geom_dofmap);
  device_array<std::int8_t> bc_marker_d(lv.bc_marker), no_marker_d(std::vector<std::int8_t>(lv.ndofs(), 0));
  auto [lcells, bcells]
      = pmg_amd::compute_boundary_cells(lv.dofmap, mesh.ncells_owned, mesh.ncells, nd * nd * nd, lv.size_local);

  // Create operators (:221-229); the second one carries no Dirichlet rows and serves the lifting
  acc::MatFreeLaplacian<T> op(order, constants_d.span(), dofmap_d.span(), xgeom_d.span(), xdofmap_d.span(), {}, {},
                              lcells, bcells, bc_marker_d.span());
  acc::MatFreeLaplacian<T> op_free(order, constants_d.span(), dofmap_d.span(), xgeom_d.span(), xdofmap_d.span(), {},
                                   {}, lcells, bcells, no_marker_d.span());
  op.compute_diag_inverse(map);

  // Assemble RHS (:231-236)
  std::vector<T> fh(lv.ndofs()), gh(lv.ndofs());
  for (std::int32_t d = 0; d < lv.ndofs(); ++d)
  {
    const T dx = (lv.x[3 * d] - 0.5) * (lv.x[3 * d] - 0.5), dy = (lv.x[3 * d + 1] - 0.5) * (lv.x[3 * d + 1] - 0.5);
    fh[d] = 1000 * std::exp(-(dx + dy) / 0.02);
    gh[d] = lv.bc_marker[d] ? g_bc : 0.0;
  }
  DeviceVector f(map, 1), g(map, 1), Ag(map, 1), b(map, 1);
  for (T& v : fh) // assemble_rhs weights by the cell's kappa (the pmg form, examples/pmg/poisson.py:35-40); L here has none
    v /= kappa;
  upload(f, fh);
  upload(g, gh);
  op_free.assemble_rhs(f, b);  // L = inner(f, v) * dx with the GLL rule
  op_free(g, Ag);              // apply_lifting
  acc::axpy(b, -1.0, Ag, b);   // b -= A g
  {
    std::vector<T> bh = b.data_copy(); // set_bc
    for (std::int32_t d = 0; d < lv.ndofs(); ++d)
      if (lv.bc_marker[d])
        bh[d] = g_bc;
    upload(b, bh);
  }

  DeviceVector b_d(map, 1);
  b_d.set(T{1.0});
  b_d.scatter_fwd();

  DeviceVector x(map, 1);
  x.set(T{0.0});

  // Create distributed CG solver (:245-254)
  dolfinx::acc::CGSolver<DeviceVector> cg(map, 1);
  cg.set_max_iterations(o.cg_its);
  cg.set_tolerance(1e-6);
  cg.store_coefficients(true);

  auto t0 = std::chrono::steady_clock::now();
  int its = cg.solve(op, x, b_d, false);
  hip_check(hipDeviceSynchronize(), "sync");
  const double t_cg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  std::vector<T> eign = cg.compute_eigenvalues();
  std::sort(eign.begin(), eign.end());
  std::array<T, 2> eig_range = {0.1 * eign.back(), 1.1 * eign.back()};

  if (root)
  {
    std::printf("Number of iterations %d\n", its);
    std::printf("Computed eigs = (%.15e, %.15e)\n", eign.front(), eign.back());
    std::printf("Using eig range:%.15e - %.15e\n", eig_range[0], eig_range[1]);
  }

  dolfinx::acc::Chebyshev<DeviceVector> cheb(map, 1, eig_range);

  // Try non-zero initial guess to make sure that works OK (:275-280)
  std::vector<T> sol(lv.ndofs(), 1.0);
  for (std::int32_t d = 0; d < lv.ndofs(); ++d)
    if (lv.bc_marker[d])
      sol[d] = g_bc;
  upload(x, sol);

  // the residual history the reference prints from inside the solver (src/chebyshev.hpp:62-89, verbose):
  // one more iteration per line, from the same guess
  DeviceVector r(map, 1), Ax(map, 1), x0(map, 1);
  acc::copy(x0, x);
  auto residual = [&](DeviceVector& xx) {
    op(xx, Ax);
    acc::axpy(r, -1.0, Ax, b);
    return acc::norm(r);
  };
  {
    const T r0 = residual(x), bn = acc::norm(b);
    if (root)
      std::printf("Norm of b = %.15e\nChebyshev iteration 0: residual norm = %.15e\n", bn, r0);
  }
  for (int k : {1, 2, 5, 10, 20, o.cheb_its})
  {
    if (k > o.cheb_its)
      continue;
    acc::copy(x, x0);
    cheb.set_max_iterations(k);
    cheb.solve(op, x, b, false);
    const T rn = residual(x);
    if (root)
      std::printf("Chebyshev iteration %d: residual norm = %.15e\n", k, rn);
  }
  t0 = std::chrono::steady_clock::now();
  acc::copy(x, x0);
  cheb.set_max_iterations(o.cheb_its);
  cheb.solve(op, x, b, false);
  hip_check(hipDeviceSynchronize(), "sync");
  const double t_cheb = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const T xn = acc::norm(x);
  if (root)
  {
    std::printf("Norm of x = %.15e\n", xn);
    std::printf("ZZZ CG %.6f s\nZZZ Chebyshev %.6f s\n", t_cg, t_cheb); // the reference's timer names (:252,268)
  }
}
} // namespace

int main(int argc, char** argv)
{
  Options o;
  std::size_t ndofs = 0;
  o.rank = examples::default_rank();
  try
  {
    for (int i = 1; i < argc; ++i)
    {
      auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
      if (!std::strcmp(argv[i], "--n"))
        o.n = std::atoi(next());
      else if (!std::strcmp(argv[i], "--ndofs")) // dofs per rank, like the reference (:47-49)
        ndofs = std::strtoull(next(), nullptr, 10);
      else if (!std::strcmp(argv[i], "--degree"))
        o.order = std::atoi(next());
      else if (!std::strcmp(argv[i], "--cg-its"))
        o.cg_its = std::atoi(next());
      else if (!std::strcmp(argv[i], "--cheb-its"))
        o.cheb_its = std::atoi(next());
      else if (!std::strcmp(argv[i], "--ranks"))
        o.ranks = examples::parse3(next());
      else if (!std::strcmp(argv[i], "--rank"))
        o.rank = std::atoi(next());
      else if (!std::strcmp(argv[i], "--native-comm"))
        o.native_comm = true;
      else if (!std::strcmp(argv[i], "--halo")) // exchange | windows
        o.windows = std::string(next()) == "windows";
      else if (!std::strcmp(argv[i], "--comm")) // rccl | windows
        o.window_comm = std::string(next()) == "windows";
      else if (!std::strcmp(argv[i], "--id-file"))
        o.id_file = next();
      else
      {
        std::cout << "usage: cg [--n cells_per_direction | --ndofs N_per_rank] [--degree P] [--cg-its N] [--cheb-its N]\n"
                     "          [--ranks px,py,pz [--rank r] [--id-file F]] [--native-comm] [--halo exchange|windows] [--comm rccl|windows]\n";
        return !std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h") ? 0 : 2;
      }
    }
    if (o.order < 1 || o.order > PMG_MAX_DEGREE)
      throw std::runtime_error("Unsupported degree");
    if (ndofs)
      o.n = examples::cells_for_ndofs(ndofs * (std::size_t)o.size(), o.order);
    examples::select_device(o);
    solve(o);
  }
  catch (const std::exception& ex)
  {
    std::cerr << "error: " << ex.what() << "\n";
    return 1;
  }
  return 0;
}
