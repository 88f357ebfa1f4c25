// p-multigrid driver: the MI355X counterpart of the reference's
// examples/pmg/main.cpp:solve (:41-380), written against the same concept names
// (include/pmg_amd.hpp).  Defaults are BASELINE config 2: 64^3 hexes, degrees
// 1,2,4, Chebyshev(3).  Per level: operator (:270-272), matrix-free inverse
// diagonal (replaces the CSR assembly of :274-279), load vector of
// f = -div(kappa grad(sin 2 pi x sin 3 pi y sin 4 pi z)) / kappa sampled at the
// GLL nodes (examples/pmg/poisson.py:6-8,30,35-40; :289-300), eigenvalue estimate
// by 20 iterations of Jacobi-CG on b = 1 (:306-327); then the interpolators
// (:336-341), the V-cycle (:348-355) and `--cycles` applications from x = 0 with
// the residual norm printed after each (:362-367, verbose).  `--pcg` additionally
// solves with CG preconditioned by the cycle.  Single rank.
#include "../common/box_mesh.hpp"
#include "pmg_amd.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <tuple>

using namespace pmg_amd;
using DeviceVector = acc::Vector;

int main(int argc, char** argv)
{
  int n = 64, cheb_its = 3, cycles = 10;
  std::size_t ndofs = 0;
  std::vector<int> orders = {1, 2, 4};
  bool pcg = false, coarse_cg = false;
  for (int i = 1; i < argc; ++i)
  {
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
    if (!std::strcmp(argv[i], "--n"))
      n = std::atoi(next());
    else if (!std::strcmp(argv[i], "--ndofs"))
      ndofs = std::strtoull(next(), nullptr, 10);
    else if (!std::strcmp(argv[i], "--orders"))
    {
      orders.clear();
      std::stringstream ss(next());
      for (std::string tok; std::getline(ss, tok, ',');)
        orders.push_back(std::atoi(tok.c_str()));
    }
    else if (!std::strcmp(argv[i], "--smoother-its"))
      cheb_its = std::atoi(next());
    else if (!std::strcmp(argv[i], "--cycles"))
      cycles = std::atoi(next());
    else if (!std::strcmp(argv[i], "--pcg"))
      pcg = true;
    else if (!std::strcmp(argv[i], "--coarse-cg"))
      coarse_cg = true;
    else
    {
      std::cout << "usage: pmg [--n cells_per_direction | --ndofs N] [--orders 1,2,4] [--smoother-its K] "
                   "[--cycles C] [--pcg] [--coarse-cg]\n";
      return !std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h") ? 0 : 2;
    }
  }
  try
  {
    if (orders.empty() || !std::is_sorted(orders.begin(), orders.end())
        || std::adjacent_find(orders.begin(), orders.end()) != orders.end())
      throw std::runtime_error("--orders must be strictly ascending (coarse to fine)");
    if (ndofs)
      n = examples::cells_for_ndofs(ndofs, orders.back());
    const std::size_t L = orders.size();
    const double kappa = 2.0; // :190-193

    examples::BoxMesh mesh(n);
    device_array<double> kappa_d(std::vector<double>(mesh.ncells(), kappa));
    device_array<double> xgeom(mesh.xgeom);
    device_array<std::int32_t> xdofmap(mesh.geom_dofmap);
    std::vector<int> lcells, bcells; // src/mesh.hpp:105-143; the same split on every level

    std::vector<std::shared_ptr<const IndexMap>> maps(L);
    std::vector<device_array<std::int32_t>> dofmaps(L);
    std::vector<device_array<std::int8_t>> bc_markers(L);
    std::vector<std::shared_ptr<acc::MatFreeLaplacian>> operators(L);
    std::vector<std::shared_ptr<DeviceVector>> bs(L);
    for (std::size_t i = 0; i < L; ++i)
    {
      const int P = orders[i], nd = P + 1;
      std::vector<double> gll(nd), w(nd);
      check(pmg_gll_table(nd, gll.data(), w.data()));
      examples::FunctionSpace V(mesh, P, gll);
      std::cout << "Level " << i << ": degree " << P << ", " << V.ndofs << " dofs\n";
      if (i == 0)
        std::tie(lcells, bcells) = compute_boundary_cells(V.dofmap, mesh.ncells(), mesh.ncells(), nd * nd * nd, V.ndofs);
      maps[i] = std::make_shared<const IndexMap>(V.ndofs, 0);
      dofmaps[i].assign(V.dofmap);
      bc_markers[i].assign(V.bc_marker);
      operators[i] = std::make_shared<acc::MatFreeLaplacian>(P, kappa_d.span(), dofmaps[i].span(), xgeom.span(),
                                                             xdofmap.span(), std::span<const double>{},
                                                             std::span<const double>{}, lcells, bcells,
                                                             bc_markers[i].span());
      operators[i]->compute_diag_inverse(maps[i]);

      std::vector<double> fh(V.ndofs);
      const double pi = M_PI, c2 = (4.0 + 9.0 + 16.0) * pi * pi;
      for (std::int32_t d = 0; d < V.ndofs; ++d)
        fh[d] = c2 * std::sin(2 * pi * V.x[3 * d]) * std::sin(3 * pi * V.x[3 * d + 1])
                * std::sin(4 * pi * V.x[3 * d + 2]);
      DeviceVector f(maps[i], 1);
      f.copy_from_host(fh);
      bs[i] = std::make_shared<DeviceVector>(maps[i], 1);
      operators[i]->assemble_rhs(f, *bs[i]);
    }

    // Chebyshev smoother for each level, :306-330
    std::vector<std::shared_ptr<acc::Chebyshev<DeviceVector>>> smoothers(L);
    for (std::size_t i = 0; i < L; ++i)
    {
      acc::CGSolver<DeviceVector> cg(maps[i], 1);
      cg.set_max_iterations(20);
      cg.set_tolerance(1e-6);
      cg.store_coefficients(true);
      DeviceVector x(maps[i], 1), y(maps[i], 1);
      x.set(0.0);
      y.set(1.0);
      [[maybe_unused]] int its = cg.solve(*operators[i], x, y, false);
      std::vector<double> eign = cg.compute_eigenvalues();
      std::sort(eign.begin(), eign.end());
      std::printf("Eigenvalues level %zu: %.17g - %.17g\n", i, eign.front(), eign.back());
      std::array<double, 2> eig_range = {0.1 * eign.back(), 1.1 * eign.back()};
      smoothers[i] = std::make_shared<acc::Chebyshev<DeviceVector>>(maps[i], 1, eig_range);
      smoothers[i]->set_max_iterations(cheb_its);
    }

    std::vector<std::int32_t> lcells32(lcells.begin(), lcells.end()), bcells32(bcells.begin(), bcells.end());
    std::vector<std::shared_ptr<Interpolator>> interpolators(L - 1);
    for (std::size_t i = 0; i + 1 < L; ++i)
      interpolators[i] = std::make_shared<Interpolator>(orders[i], orders[i + 1], dofmaps[i].span(),
                                                        dofmaps[i + 1].span(), lcells32, bcells32);

    using PMG = acc::MultigridPreconditioner<DeviceVector, acc::MatFreeLaplacian, Interpolator,
                                             acc::Chebyshev<DeviceVector>>;
    PMG pmg(maps, 1, bc_markers[0].span());
    pmg.set_solvers(smoothers);
    pmg.set_operators(operators);
    pmg.set_interpolators(interpolators);
    if (coarse_cg) // the reference's --amg role (:331-335: KSPCG, 60 iterations; here Jacobi-preconditioned)
    {
      auto coarse = std::make_shared<acc::CGSolver<DeviceVector>>(maps[0], 1);
      coarse->set_max_iterations(60);
      coarse->set_tolerance(1e-5);
      pmg.set_coarse_solver(coarse);
    }

    DeviceVector x(maps.back(), 1);
    x.set(0.0);
    std::printf("Norm of b = %.15e\n", acc::norm(*bs.back()));
    for (int i = 0; i < cycles; ++i)
    {
      const double rnorm = pmg.apply(*bs.back(), x, true);
      std::printf("Cycle %d: residual norm = %.15e\n", i + 1, rnorm);
    }
    std::printf("Norm of x = %.15e\n", acc::norm(x));

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
    std::printf("V-cycle: %.3f ms, %.3f GDoF/s\n", ms / reps, maps.back()->size_local() / (ms * 1e-3 / reps) * 1e-9);

    if (pcg)
    {
      acc::CGSolver<DeviceVector> cg(maps.back(), 1);
      cg.set_max_iterations(100);
      cg.set_tolerance(1e-8);
      cg.set_flexible(coarse_cg); // the cycle is not a fixed linear operator with a Krylov coarse solve
      x.set(0.0);
      const int its = cg.solve(*operators.back(), x, *bs.back(), pmg, false);
      DeviceVector Ax(maps.back(), 1), r(maps.back(), 1);
      (*operators.back())(x, Ax);
      acc::axpy(r, -1.0, Ax, *bs.back()); // r = b - A x
      std::printf("PCG with V-cycle preconditioner: %d iterations, |b - A x| / |b| = %.3e, Norm of x = %.15e\n", its,
                  acc::norm(r) / acc::norm(*bs.back()), acc::norm(x));
    }
  }
  catch (const std::exception& ex)
  {
    std::cerr << "error: " << ex.what() << "\n";
    return 1;
  }
  return 0;
}
