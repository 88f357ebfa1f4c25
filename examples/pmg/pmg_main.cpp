// p-multigrid driver: the MI355X counterpart of the reference's examples/pmg/main.cpp:solve
// (:41-380), written against the reference's own type spellings through include/pmg_amd.hpp
// (PMG_AMD_DOLFINX_NAMESPACE): the smoother set-up, interpolators, V-cycle wiring and the cycle loop
// below are the reference's lines :306-365; what stands above them replaces dolfinx (mesh, function
// spaces, index maps, load vector).  Defaults are BASELINE config 2: 64^3 hexes, degrees 1,2,4,
// Chebyshev(3).  Per level: operator (:270-272), matrix-free inverse diagonal (replaces the CSR
// assembly of :274-279), load vector of f = -div(kappa grad(sin 2 pi x sin 3 pi y sin 4 pi z)) / kappa
// sampled at the GLL nodes (examples/pmg/poisson.py:6-8,30,35-40; :289-300).
//
//   --amg            coarse solver on the degree-1 level (:331-335): CG <= 60 iterations, rtol 1e-5,
//                    preconditioned by the library's algebraic multigrid (hypre's role)
//   --amg-cycles N   the same hierarchy as N stationary AMG cycles instead of the Krylov solve
//   --coarse-cg      Jacobi-preconditioned CG (60 iterations) as the coarse solver
//   --pcg            additionally solve with CG preconditioned by the cycle (random right-hand side
//                    with --random-rhs)
//   --ranks px,py,pz one process per GPU and brick (the reference runs under mpirun -n 8,
//                    examples/pmg/submit.sh:29): halo exchange and reductions on the library's RCCL
//                    communicator.  Rank and world size come from --rank / RANK (OMPI_COMM_WORLD_RANK,
//                    PMI_RANK); the GPU from LOCAL_RANK; rank 0 publishes the communicator id in
//                    --id-file.  examples/pmg/run_ranks.sh launches the processes.
//   --graph          time the cycles as hipGraph replays (one launch per cycle; with --ranks the grouped send/recv of
//                    every halo exchange are captured on the compute stream: no host work per exchange)
//   --check-partition px,py,pz   host-only consistency check of the brick partition (no GPU)
//   --node-order basix   the dofmaps handed over in basix's cell-local node order (endpoints first), as dolfinx
//                    gives them to the reference (examples/pmg/main.cpp:83-87); same numbers as without
//   --output FILE    write the solution as a legacy VTK file of the fine-level GLL points (:369-379)
#define PMG_AMD_DOLFINX_NAMESPACE
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
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <tuple>

using namespace dolfinx; // examples/pmg/main.cpp:29
using T = double;        // :30
using DeviceVector = dolfinx::acc::Vector<T, acc::Device::HIP>; // :33
using pmg_amd::check;
using pmg_amd::device_array;
using pmg_amd::hip_check;

/// The coarse solver of the reference is CoarseSolverType<T> (src/amg.hpp): anything with
/// solve(DeviceVector& x, DeviceVector& y).  Here: the library's AMG (Krylov or stationary mode).
template <typename U>
using CoarseSolverType = pmg_amd::acc::AmgSolver<acc::Vector<U, acc::Device::HIP>>;

namespace
{
/// What `V[i]->element()->basix_element()` of the reference reaches: the degree.
struct Element
{
  pmg_amd::LagrangeElement e;
  const pmg_amd::LagrangeElement& basix_element() const { return e; }
};
struct Space
{
  std::shared_ptr<Element> el;
  examples::PartitionLevel lv;
  std::shared_ptr<Element> element() const { return el; }
};

struct Options : examples::RankOptions
{
  int n = 64, cheb_its = 3, cycles = 10, amg_cycles = 0;
  std::vector<int> orders = {1, 2, 4};
  bool pcg = false, coarse_cg = false, use_amg = false, random_rhs = false, graph = false;
  std::string output;
  pmg_amd::NodeOrder node_order = pmg_amd::NodeOrder::ascending; // of the dofmaps handed to the library
  bool amg_gather = false; // --amg-setup gathered
};
using examples::parse3;

template <typename FineOperator>
void solve(const Options& o)
{
  const int size = o.ranks[0] * o.ranks[1] * o.ranks[2];
  const bool root = o.rank == 0;
  const std::vector<int>& order = o.orders;
  const T kappa = 2.0; // :190-193

  std::shared_ptr<const pmg_amd::Communicator> comm = examples::bootstrap(o);

  examples::BrickPartition mesh(o.n, o.ranks, o.rank);
  device_array<T> constants(std::vector<T>(mesh.ncells, kappa));
  device_array<T> geom_x_d(mesh.xgeom);
  device_array<std::int32_t> geom_x_dofmap_d(mesh.geom_dofmap);
  std::span<const T> device_constants = constants.span(), geom_x = geom_x_d.span();
  std::span<const std::int32_t> geom_x_dofmap = geom_x_dofmap_d.span();

  // function spaces, index maps, device copies of the dofmaps and Dirichlet markers (:83-257)
  std::vector<std::shared_ptr<Space>> V(order.size());
  std::vector<std::shared_ptr<const common::IndexMap>> maps(V.size());
  std::vector<device_array<std::int32_t>> dofmaps_d(V.size());
  std::vector<device_array<std::int8_t>> bc_markers_d(V.size());
  std::vector<std::span<const std::int32_t>> device_dofmaps(V.size());
  std::vector<std::span<const std::int8_t>> bc_marker_d_span(V.size());
  std::vector<std::span<const T>> geometry_dphi_d_span(V.size()), Gweights_d_span(V.size()); // from the degree
  std::vector<int> lcells, bcells; // src/mesh.hpp:105-143; the same split on every level
  for (std::size_t i = 0; i < V.size(); i++)
  {
    const int P = order[i], nd = P + 1;
    std::vector<double> gll(nd), w(nd);
    check(pmg_gll_table(nd, gll.data(), w.data()));
    V[i] = std::make_shared<Space>(Space{std::make_shared<Element>(Element{{P}}), mesh.level(P, gll)});
    const examples::PartitionLevel& lv = V[i]->lv;
    if (comm)
      maps[i] = std::make_shared<const common::IndexMap>(lv.size_local, lv.num_ghosts, lv.send_indices,
                                                         lv.recv_indices, comm, lv.neighbors, lv.send_counts,
                                                         lv.recv_counts, o.halo());
    else
      maps[i] = std::make_shared<const common::IndexMap>(lv.size_local, lv.num_ghosts);
    if (i == V.size() - 1) // :97 (finest space)
      std::tie(lcells, bcells) = pmg_amd::compute_boundary_cells(lv.dofmap, mesh.ncells_owned, mesh.ncells,
                                                                 nd * nd * nd, lv.size_local);
    // --node-order basix: hand the library the dofmaps as dolfinx holds them (endpoints first per direction,
    // examples/pmg/main.cpp:83-87); the brick partition generates them ascending
    if (o.node_order == pmg_amd::NodeOrder::ascending)
      dofmaps_d[i].assign(lv.dofmap);
    else
      dofmaps_d[i].assign(pmg_amd::dofmap_in_node_order(lv.dofmap, P, o.node_order));
    bc_markers_d[i].assign(lv.bc_marker);
    device_dofmaps[i] = dofmaps_d[i].span();
    bc_marker_d_span[i] = bc_markers_d[i].span();
    if (root)
      std::printf("Level %zu: degree %d, %lld dofs (rank 0: %d owned + %d ghosts, %zu neighbours)\n", i, P,
                  (long long)mesh.global_ndofs(P), lv.size_local, lv.num_ghosts, lv.neighbors.size());
  }

  std::vector<std::shared_ptr<FineOperator>> operators(V.size());
  std::vector<std::shared_ptr<DeviceVector>> bs(V.size());
  for (std::size_t i = 0; i < V.size(); i++)
  {
    operators[i] = std::make_shared<acc::MatFreeLaplacian<T>>(
        order[i], device_constants, device_dofmaps[i], geom_x, geom_x_dofmap, geometry_dphi_d_span[i],
        Gweights_d_span[i], lcells, bcells, bc_marker_d_span[i], 0, o.node_order); // :270-272
    operators[i]->compute_diag_inverse(maps[i]);                  // replaces :274-279 (no CSR)

    const examples::PartitionLevel& lv = V[i]->lv;
    std::vector<T> fh(lv.ndofs());
    const double pi = M_PI, c2 = (4.0 + 9.0 + 16.0) * pi * pi;
    for (std::int32_t d = 0; d < lv.ndofs(); ++d)
      fh[d] = c2 * std::sin(2 * pi * lv.x[3 * d]) * std::sin(3 * pi * lv.x[3 * d + 1]) * std::sin(4 * pi * lv.x[3 * d + 2]);
    DeviceVector f(maps[i], 1);
    hip_check(hipMemcpy(f.mutable_array().data(), fh.data(), sizeof(T) * fh.size(), hipMemcpyHostToDevice), "H2D");
    bs[i] = std::make_shared<DeviceVector>(maps[i], 1);
    operators[i]->assemble_rhs(f, *bs[i]); // :289-300
  }

  // ---------------------------------------------------------------------------------------------
  // From here to the end of the cycle loop: examples/pmg/main.cpp:303-367 (logging calls dropped,
  // Chebyshev degree and cycle count from the command line).

  // Create chebyshev smoother for each level
  std::vector<std::shared_ptr<acc::Chebyshev<DeviceVector>>> smoothers(V.size());
  for (std::size_t i = 0; i < V.size(); i++)
  {
    dolfinx::acc::CGSolver<DeviceVector> cg(maps[i], 1);
    cg.set_max_iterations(20);
    cg.set_tolerance(1e-6);
    cg.store_coefficients(true);

    DeviceVector x(maps[i], 1);

    x.set(T{0.0});
    DeviceVector y(maps[i], 1);
    y.set(T{1.0});

    [[maybe_unused]] int its = cg.solve(*operators[i], x, y, false);
    std::vector<T> eign = cg.compute_eigenvalues();
    std::sort(eign.begin(), eign.end());
    if (root)
      std::printf("Eigenvalues level %zu: %.17g - %.17g\n", i, eign.front(), eign.back());
    std::array<T, 2> eig_range = {0.1 * eign.back(), 1.1 * eign.back()};
    smoothers[i] = std::make_shared<acc::Chebyshev<DeviceVector>>(maps[i], 1, eig_range);
    smoothers[i]->set_max_iterations(o.cheb_its);
  }

  // Create Matrix-Free Interpolators
  std::vector<std::shared_ptr<Interpolator<T>>> matfree_interpolators(V.size() - 1);

  for (int i = 0; i < (int)V.size() - 1; ++i)
  {
    matfree_interpolators[i] = std::make_shared<Interpolator<T>>(
        V[i]->element()->basix_element(), V[i + 1]->element()->basix_element(), device_dofmaps[i],
        device_dofmaps[i + 1], lcells, bcells, o.node_order);
  }

  std::shared_ptr<CoarseSolverType<T>> coarse_solver; // :331-335
  if (o.use_amg)
  {
    auto t0 = std::chrono::steady_clock::now();
    if (size > 1) // first coarsening per rank, level 1 gathered on every rank (--amg-setup gathered: the global
                  // degree-1 matrix on every rank instead); one all-reduce of a level-1 vector per cycle
      coarse_solver = std::make_shared<CoarseSolverType<T>>(*operators[0], maps[0], V[0]->lv.local_to_global,
                                                           mesh.global_ndofs(order[0]), o.amg_gather);
    else
      coarse_solver = std::make_shared<CoarseSolverType<T>>(*operators[0], maps[0]);
    if (o.amg_cycles > 0)
      coarse_solver->set_cycles(o.amg_cycles);
    if (root)
      std::printf("AMG coarse solver: %d levels, set-up %.2f s, %s\n", coarse_solver->num_levels(),
                  std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(),
                  o.amg_cycles > 0 ? "stationary cycles" : "CG <= 60 iterations, rtol 1e-5");
  }

  using SolverType = acc::Chebyshev<DeviceVector>;
  using PMG = acc::MultigridPreconditioner<DeviceVector, FineOperator, SolverType, CoarseSolverType<T>,
                                           Interpolator<T>>;

  PMG pmg(maps, 1, bc_marker_d_span[0]);
  pmg.set_solvers(smoothers);
  pmg.set_operators(operators);
  pmg.set_coarse_solver(coarse_solver);

  // Sets matrix-free kernels to do interpolation
  pmg.set_interpolators(matfree_interpolators);

  std::shared_ptr<dolfinx::acc::CGSolver<DeviceVector>> coarse_cg;
  if (o.coarse_cg && !o.use_amg) // a second coarse-solver type through the C ABI's native CG hook
  {
    coarse_cg = std::make_shared<dolfinx::acc::CGSolver<DeviceVector>>(maps[0], 1);
    coarse_cg->set_max_iterations(60);
    coarse_cg->set_tolerance(1e-5);
    check(pmg_multigrid_set_coarse_solver(pmg.handle(), coarse_cg->handle()));
  }

  // Create solution vector
  DeviceVector x(maps.back(), 1);
  x.set(T{0.0});

  if (root)
    std::printf("Norm of b = %.15e\n", acc::norm(*bs.back()));
  else
    (void)acc::norm(*bs.back()); // reductions are collective
  int niter = o.cycles;
  for (int i = 0; i < niter; i++)
  {
    const T rnorm = pmg.apply(*bs.back(), x, true);
    if (root)
      std::printf("Cycle %d: residual norm = %.15e\n", i + 1, rnorm);
  }
  // ---------------------------------------------------------------------------------------------
  {
    const T xn = acc::norm(x);
    if (root)
      std::printf("Norm of x = %.15e\n", xn);
  }

  // --graph: the timed cycles replayed as one hipGraph each (with --ranks: the halo exchange captured with them)
  if (o.graph)
    check(pmg_multigrid_set_graph(pmg.handle(), 1));
  // timing of the cycle alone (no residual evaluation)
  hipEvent_t e0, e1;
  hip_check(hipEventCreate(&e0), "event");
  hip_check(hipEventCreate(&e1), "event");
  const int reps = 10;
  pmg.apply(*bs.back(), x, false);
  hip_check(hipEventRecord(e0, nullptr), "record");
  for (int i = 0; i < reps; ++i)
    pmg.apply(*bs.back(), x, false);
  hip_check(hipEventRecord(e1, nullptr), "record");
  hip_check(hipEventSynchronize(e1), "sync");
  float ms = 0;
  hip_check(hipEventElapsedTime(&ms, e0, e1), "elapsed");
  if (root)
    std::printf("V-cycle: %.3f ms, %.3f GDoF/s%s\n", ms / reps,
                (double)mesh.global_ndofs(order.back()) / (ms * 1e-3 / reps) * 1e-9,
                o.graph ? (pmg_multigrid_graph_replays(pmg.handle()) > 0 ? " (hipGraph replays)" : " (not capturable: eager)")
                        : "");
  if (o.graph)
    check(pmg_multigrid_set_graph(pmg.handle(), 0));

  if (o.pcg)
  {
    dolfinx::acc::CGSolver<DeviceVector> cg(maps.back(), 1);
    cg.set_max_iterations(100);
    cg.set_tolerance(1e-8);
    // with a Krylov coarse solve inside, the cycle is not a fixed linear operator
    cg.set_flexible(o.coarse_cg || (o.use_amg && o.amg_cycles == 0)); // (stationary AMG cycles: a fixed operator)
    DeviceVector rhs(maps.back(), 1);
    if (o.random_rhs)
    {
      const examples::PartitionLevel& lv = V.back()->lv;
      std::vector<T> g(lv.ndofs());
      for (std::int32_t d = 0; d < lv.ndofs(); ++d) // a function of the GLOBAL dof: the same vector on any partition
      {
        std::mt19937_64 gen(0x9E3779B97F4A7C15ull ^ (std::uint64_t)lv.local_to_global[d]);
        g[d] = lv.bc_marker[d] ? 0.0 : std::normal_distribution<T>()(gen);
      }
      hip_check(hipMemcpy(rhs.mutable_array().data(), g.data(), sizeof(T) * g.size(), hipMemcpyHostToDevice), "H2D");
    }
    else
      acc::copy(rhs, *bs.back());
    x.set(0.0);
    const int its = cg.solve(*operators.back(), x, rhs, pmg, false);
    DeviceVector Ax(maps.back(), 1), r(maps.back(), 1);
    (*operators.back())(x, Ax);
    acc::axpy(r, -1.0, Ax, rhs); // r = b - A x
    const T rn = acc::norm(r), bn = acc::norm(rhs), xn = acc::norm(x);
    if (root)
      std::printf("PCG with V-cycle preconditioner: %d iterations, |b - A x| / |b| = %.3e, Norm of x = %.15e\n", its,
                  rn / bn, xn);
  }

  if (!o.output.empty()) // :369-379 (VTX there; here the fine-level points of this rank as legacy VTK)
  {
    const examples::PartitionLevel& lv = V.back()->lv;
    auto xv = x.thrust_vector(); // :373
    std::vector<T> u(xv.size());
    thrust::copy(xv.begin(), xv.end(), u.begin()); // :374
    const std::string name = size > 1 ? o.output + "." + std::to_string(o.rank) : o.output;
    std::ofstream f(name);
    f << "# vtk DataFile Version 3.0\npmg_amd solution, degree " << order.back() << "\nASCII\nDATASET POLYDATA\nPOINTS "
      << lv.size_local << " double\n";
    f.precision(17);
    for (std::int32_t d = 0; d < lv.size_local; ++d)
      f << lv.x[3 * d] << " " << lv.x[3 * d + 1] << " " << lv.x[3 * d + 2] << "\n";
    f << "POINT_DATA " << lv.size_local << "\nSCALARS u double 1\nLOOKUP_TABLE default\n";
    for (std::int32_t d = 0; d < lv.size_local; ++d)
      f << u[d] << "\n";
    if (root)
      std::printf("Solution written to %s\n", name.c_str());
  }
}
} // namespace

int main(int argc, char** argv)
{
  Options o;
  std::size_t ndofs = 0;
  std::string check_dims;
  o.rank = examples::default_rank();
  try
  {
    for (int i = 1; i < argc; ++i)
    {
      auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
      if (!std::strcmp(argv[i], "--n"))
        o.n = std::atoi(next());
      else if (!std::strcmp(argv[i], "--ndofs")) // dofs per rank, like the reference (:412-435)
        ndofs = std::strtoull(next(), nullptr, 10);
      else if (!std::strcmp(argv[i], "--orders"))
      {
        o.orders.clear();
        std::stringstream ss(next());
        for (std::string tok; std::getline(ss, tok, ',');)
          o.orders.push_back(std::atoi(tok.c_str()));
      }
      else if (!std::strcmp(argv[i], "--smoother-its"))
        o.cheb_its = std::atoi(next());
      else if (!std::strcmp(argv[i], "--cycles"))
        o.cycles = std::atoi(next());
      else if (!std::strcmp(argv[i], "--pcg"))
        o.pcg = true;
      else if (!std::strcmp(argv[i], "--random-rhs"))
        o.random_rhs = true;
      else if (!std::strcmp(argv[i], "--graph"))
        o.graph = true;
      else if (!std::strcmp(argv[i], "--coarse-cg"))
        o.coarse_cg = true;
      else if (!std::strcmp(argv[i], "--amg"))
        o.use_amg = true;
      else if (!std::strcmp(argv[i], "--amg-cycles"))
      {
        o.use_amg = true;
        o.amg_cycles = std::atoi(next());
      }
      else if (!std::strcmp(argv[i], "--ranks"))
        o.ranks = parse3(next());
      else if (!std::strcmp(argv[i], "--rank"))
        o.rank = std::atoi(next());
      else if (!std::strcmp(argv[i], "--native-comm")) // one rank through the RCCL communicator anyway
        o.native_comm = true;
      else if (!std::strcmp(argv[i], "--comm"))
      {
        const std::string how = next();
        if (how != "windows" && how != "rccl")
          throw std::runtime_error("--comm rccl | windows");
        o.window_comm = how == "windows";
      }
      else if (!std::strcmp(argv[i], "--halo"))
      {
        const std::string how = next();
        if (how != "windows" && how != "exchange")
          throw std::runtime_error("--halo exchange | windows");
        o.windows = how == "windows";
      }
      else if (!std::strcmp(argv[i], "--id-file"))
        o.id_file = next();
      else if (!std::strcmp(argv[i], "--amg-setup"))
      {
        const std::string v = next();
        if (v != "gathered" && v != "distributed")
          throw std::runtime_error("--amg-setup distributed | gathered");
        o.amg_gather = v == "gathered";
      }
      else if (!std::strcmp(argv[i], "--node-order"))
      {
        const std::string v = next();
        if (v == "basix" || v == "endpoints_first")
          o.node_order = pmg_amd::NodeOrder::endpoints_first;
        else if (v == "ascending")
          o.node_order = pmg_amd::NodeOrder::ascending;
        else
          throw std::runtime_error("--node-order ascending | basix");
      }
      else if (!std::strcmp(argv[i], "--output"))
        o.output = next();
      else if (!std::strcmp(argv[i], "--check-partition"))
        check_dims = next();
      else
      {
        std::cout << "usage: pmg [--n cells_per_direction | --ndofs N_per_rank] [--orders 1,2,4] [--smoother-its K]\n"
                     "           [--cycles C] [--pcg [--random-rhs]] [--amg | --amg-cycles N | --coarse-cg] [--graph]\n"
                     "           [--ranks px,py,pz [--rank r] [--id-file F]] [--native-comm] [--halo exchange|windows]\n"
                     "           [--comm rccl|windows]\n"
                     "           [--node-order ascending|basix] [--amg-setup distributed|gathered]\n"
                     "           [--output FILE]\n"
                     "           [--check-partition px,py,pz]\n";
        return !std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h") ? 0 : 2;
      }
    }
    if (o.orders.empty() || !std::is_sorted(o.orders.begin(), o.orders.end())
        || std::adjacent_find(o.orders.begin(), o.orders.end()) != o.orders.end())
      throw std::runtime_error("--orders must be strictly ascending (coarse to fine)");
    const int size = o.ranks[0] * o.ranks[1] * o.ranks[2];
    if (ndofs) // cells per direction of the whole mesh so that a rank holds about ndofs fine dofs
      o.n = examples::cells_for_ndofs(ndofs * (std::size_t)size, o.orders.back());
    if (!check_dims.empty())
    {
      const auto dims = parse3(check_dims.c_str());
      for (int P : o.orders)
      {
        std::vector<double> gll(P + 1), w(P + 1);
        check(pmg_gll_table(P + 1, gll.data(), w.data()));
        const std::string err = examples::check_partition(o.n, dims, P, gll);
        if (!err.empty())
          throw std::runtime_error("partition check failed at degree " + std::to_string(P) + ": " + err);
      }
      std::printf("partition %dx%dx%d of %d^3 cells consistent for every degree\n", dims[0], dims[1], dims[2], o.n);
      return 0;
    }
    examples::select_device(o);
    solve<acc::MatFreeLaplacian<T>>(o);
  }
  catch (const std::exception& ex)
  {
    std::cerr << "error: " << ex.what() << "\n";
    return 1;
  }
  return 0;
}
