// Matrix-free Laplacian apply driver: the MI355X counterpart of the reference's
// examples/mat_free/main.cpp (BASELINE config 1 with the defaults --n 16 --degree 1).
// Unit cube, n^3 hexes, degree P, kappa = 2 (:132), Dirichlet marker on all exterior
// dofs (:163-165,236-240), u = 1 (:250-256), nreps applies timed, ||u||, ||y|| printed
// (:267-268).  --mat_comp (degree 1 only) compares y with the 7-point stencil the
// collocated P=1 operator reduces to, evaluated on the host (the reference compares
// with an assembled CSR operator there, :270-288).  Single rank.
#include "../common/box_mesh.hpp"
#include "pmg_amd.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace pmg_amd;
using T = double;                                          // examples/mat_free/main.cpp:31
using DeviceVector = acc::Vector<T, acc::Device::HIP>;

int main(int argc, char** argv)
{
  int n = 16, degree = 1, nreps = 1000;
  std::size_t ndofs = 0;
  bool mat_comp = false;
  std::size_t batch_size = 0; // :38,46-50: cells whose geometry tensor is held at a time (0 = all, resident)
  for (int i = 1; i < argc; ++i)
  {
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
    if (!std::strcmp(argv[i], "--n"))
      n = std::atoi(next());
    else if (!std::strcmp(argv[i], "--ndofs"))
      ndofs = std::strtoull(next(), nullptr, 10);
    else if (!std::strcmp(argv[i], "--degree"))
      degree = std::atoi(next());
    else if (!std::strcmp(argv[i], "--nreps"))
      nreps = std::atoi(next());
    else if (!std::strcmp(argv[i], "--mat_comp"))
      mat_comp = true;
    else if (!std::strcmp(argv[i], "--batch_size"))
      batch_size = std::strtoull(next(), nullptr, 10);
    else
    {
      std::cout << "usage: mat_free [--n cells_per_direction | --ndofs N] [--degree P] [--nreps R] [--mat_comp] "
                   "[--batch_size cells]\n";
      return !std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h") ? 0 : 2;
    }
  }
  try
  {
    if (degree < 1 || degree > PMG_MAX_DEGREE)
      throw std::runtime_error("Unsupported degree");
    if (ndofs)
      n = examples::cells_for_ndofs(ndofs, degree);
    const int nd = degree + 1;
    std::vector<double> gll(nd), w(nd);
    check(pmg_gll_table(nd, gll.data(), w.data()));

    examples::BoxMesh mesh(n);
    examples::FunctionSpace V(mesh, degree, gll);
    std::cout << "Mesh " << n << "^3 hexes, degree " << degree << ", " << V.ndofs << " dofs\n";

    auto map = std::make_shared<const IndexMap>(V.ndofs, 0);
    device_array<double> kappa(std::vector<double>(mesh.ncells(), 2.0)); // :132
    device_array<std::int32_t> dofmap(V.dofmap), xdofmap(mesh.geom_dofmap);
    device_array<double> xgeom(mesh.xgeom);
    device_array<std::int8_t> bc(V.bc_marker);
    // one rank: no ghost dofs, so every cell comes out local (src/mesh.hpp:105-143)
    auto [lcells, bcells] = compute_boundary_cells(V.dofmap, mesh.ncells(), mesh.ncells(), nd * nd * nd, V.ndofs);

    auto t0 = std::chrono::steady_clock::now();
    acc::MatFreeLaplacian<T> op(degree, kappa.span(), dofmap.span(), xgeom.span(), xdofmap.span(), {}, {}, lcells,
                                bcells, bc.span(), batch_size);
    DeviceVector u(map, 1), y(map, 1);
    u.set(1.0);
    op(u, y); // creates the handle (geometry, patches) and warms up
    hip_check(hipDeviceSynchronize(), "sync");
    auto t1 = std::chrono::steady_clock::now();
    std::cout << "Create matfree operator: " << std::chrono::duration<double>(t1 - t0).count() << " s\n";
    std::printf("Geometry tensor held: %.3f MB%s\n", pmg_laplacian_geometry_bytes(op.handle(map)) * 1e-6,
                batch_size ? " (recomputed batch by batch in every apply)" : " (resident)");

    hipEvent_t e0, e1;
    hip_check(hipEventCreate(&e0), "event");
    hip_check(hipEventCreate(&e1), "event");
    hip_check(hipEventRecord(e0, nullptr), "record");
    for (int i = 0; i < nreps; ++i)
      op(u, y);
    hip_check(hipEventRecord(e1, nullptr), "record");
    hip_check(hipEventSynchronize(e1), "sync");
    float ms = 0;
    hip_check(hipEventElapsedTime(&ms, e0, e1), "elapsed");
    const double per = ms * 1e-3 / nreps;
    const double N = (double)nd * nd * nd, U = (double)degree * degree * degree;
    const double bytes = (48 * N + 4 * N + 8 + 17 * U) * mesh.ncells(); // SURVEY 8d, model storedG
    std::printf("Mat-free Matvec: %d reps, %.3f us per apply, %.3f GDoF/s, %.1f GB/s algorithmic\n", nreps, per * 1e6,
                V.ndofs / per * 1e-9, bytes / per * 1e-9);
    std::printf("Norm of u = %.15e\n", acc::norm(u));
    std::printf("Norm of y = %.15e\n", acc::norm(y));

    if (mat_comp)
    {
      if (degree != 1)
        throw std::runtime_error("--mat_comp: the host comparison operator is the P=1 stencil only");
      // rows of the collocated P=1 operator on a uniform grid: kappa*h*(6 u_c - sum of the 6 neighbours),
      // Dirichlet columns masked, Dirichlet rows y = u
      const int m = n + 1;
      const double h = 1.0 / n, kap = 2.0;
      std::vector<double> uh = u.data_copy(), z(V.ndofs);
      auto at = [&](int i, int j, int k) {
        const std::size_t d = ((std::size_t)i * m + j) * m + k;
        return V.bc_marker[d] ? 0.0 : uh[d];
      };
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j)
          for (int k = 0; k < m; ++k)
          {
            const std::size_t d = ((std::size_t)i * m + j) * m + k;
            z[d] = V.bc_marker[d] ? uh[d]
                                  : kap * h
                                        * (6 * at(i, j, k) - at(i - 1, j, k) - at(i + 1, j, k) - at(i, j - 1, k)
                                           - at(i, j + 1, k) - at(i, j, k - 1) - at(i, j, k + 1));
          }
      DeviceVector zd(map, 1), e(map, 1);
      zd.copy_from_host(z);
      acc::axpy(e, -1.0, y, zd);
      std::printf("Norm of z = %.15e\n", acc::norm(zd));
      std::printf("Norm of error = %.3e\n", acc::norm(e));
    }
  }
  catch (const std::exception& ex)
  {
    std::cerr << "error: " << ex.what() << "\n";
    return 1;
  }
  return 0;
}
