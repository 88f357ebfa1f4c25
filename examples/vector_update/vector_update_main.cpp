// Halo / vector-update overlap micro-benchmark: the MI355X counterpart of the reference's
// examples/vector-update/main.cpp (:20-124) over include/pmg_amd.hpp.  Degree-2 space with about
// 50 000 dofs per rank (:24,35), x = rank, y = 1 (:103-107); 100 times
//     x.scatter_fwd_begin(); norm(x); axpy(x, 1.0, x, y); x.scatter_fwd_end();      (:110-119)
// and the norm printed per iteration on rank 0.  The exchange travels on the communicator's stream
// while the reduction and the update run on the compute stream; the time per iteration is printed
// next to that of the same loop without the exchange.
//   --ranks px,py,pz   one process per GPU and brick on the library's RCCL communicator
//                      (examples/pmg/run_ranks.sh launches the processes); --native-comm runs a
//                      single rank through the communicator as well
#define PMG_AMD_DOLFINX_NAMESPACE
// this driver generates its own mesh (examples/common/box_mesh.hpp: cell-local nodes by ascending coordinate); with
// dolfinx-generated dofmaps the macro above alone is right: the adapter then defaults to basix's order
#define PMG_AMD_DEFAULT_NODE_ORDER 0
#include "../common/box_mesh.hpp"
#include "../common/brick_partition.hpp"
#include "../common/rank_launch.hpp"
#include "pmg_amd.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace dolfinx;
using T = double; // examples/vector-update/main.cpp:19
using DeviceVector = dolfinx::acc::Vector<T, acc::Device::HIP>; // :98
using pmg_amd::check;
using pmg_amd::hip_check;

namespace
{
struct Options : examples::RankOptions
{
  int n = 0, order = 2, iterations = 100;
  std::size_t ndofs = 50000;
  bool quiet = false;
};

void run(const Options& o)
{
  const bool root = o.rank == 0;
  const int nd = o.order + 1;
  std::shared_ptr<const pmg_amd::Communicator> comm = examples::bootstrap(o);
  const int n = o.n > 0 ? o.n : examples::cells_for_ndofs(o.ndofs * (std::size_t)o.size(), o.order);
  examples::BrickPartition mesh(n, o.ranks, o.rank);
  std::vector<double> gll(nd), w(nd);
  check(pmg_gll_table(nd, gll.data(), w.data()));
  const examples::PartitionLevel lv = mesh.level(o.order, gll);
  std::shared_ptr<const common::IndexMap> map
      = comm ? std::make_shared<const common::IndexMap>(lv.size_local, lv.num_ghosts, lv.send_indices, lv.recv_indices,
                                                         comm, lv.neighbors, lv.send_counts, lv.recv_counts, o.halo())
             : std::make_shared<const common::IndexMap>(lv.size_local, lv.num_ghosts);
  if (root)
  {
    std::cout << "-----------------------------------\n";
    std::cout << "Number of ranks : " << o.size() << "\n";
    std::cout << "Number of cells-global : " << (long long)n * n * n << "\n";
    std::cout << "Number of dofs-global : " << (long long)mesh.global_ndofs(o.order) << "\n";
    std::cout << "Number of dofs-rank 0 : " << lv.size_local << " owned + " << lv.num_ghosts << " ghosts, "
              << lv.neighbors.size() << " neighbours\n";
    std::cout << "-----------------------------------\n";
  }

  DeviceVector x(map, 1);
  x.set(T(o.rank));

  DeviceVector y(map, 1);
  y.set(T{1});

  hipEvent_t e0, e1;
  hip_check(hipEventCreate(&e0), "event");
  hip_check(hipEventCreate(&e1), "event");
  auto loop = [&](bool exchange, bool print) {
    x.set(T(o.rank));
    hip_check(hipEventRecord(e0, nullptr), "record");
    for (int i = 0; i < o.iterations; i++)
    {
      if (exchange)
        x.scatter_fwd_begin();
      auto value = acc::norm(x, dolfinx::la::Norm::l2);
      acc::axpy(x, 1.0, x, y);
      if (exchange)
        x.scatter_fwd_end();

      if (root && print)
        std::printf("Dot value: %.15e\n", value);
    }
    hip_check(hipEventRecord(e1, nullptr), "record");
    hip_check(hipEventSynchronize(e1), "sync");
    float ms = 0;
    hip_check(hipEventElapsedTime(&ms, e0, e1), "elapsed");
    return ms / o.iterations * 1e3;
  };
  loop(true, false); // warm-up (first RCCL exchange sets up the channels)
  const double us_with = loop(true, !o.quiet);
  // the ghosts must hold the owners' values after the last exchange: owner value = rank of the owner + iterations
  {
    std::vector<T> xh = x.data_copy();
    double worst = 0.0;
    std::size_t idx = 0;
    for (std::size_t k = 0; k < lv.neighbors.size(); ++k)
      for (std::int32_t j = 0; j < lv.recv_counts[k]; ++j, ++idx)
      {
        // what the owner packed in the last iteration: its value after iterations - 1 updates
        const double expect = (double)lv.neighbors[k] + (o.iterations - 1);
        worst = std::fmax(worst, std::fabs(xh[lv.size_local + lv.recv_indices[idx]] - expect));
      }
    if (lv.num_ghosts > 0 && worst != 0.0)
      throw std::runtime_error("ghost values differ from the owners' (max deviation " + std::to_string(worst) + ")");
    if (root)
      std::printf("Ghost check: %d ghost values equal the owners' values\n", lv.num_ghosts);
  }
  const double us_without = loop(false, false);
  const T xn = acc::norm(x);
  if (root)
  {
    std::printf("Norm of x = %.15e\n", xn);
    std::printf("Iteration with exchange: %.2f us, without: %.2f us (halo %.1f kB out per rank 0)\n", us_with,
                us_without, 8e-3 * (double)lv.send_indices.size());
  }
}
} // namespace

int main(int argc, char** argv)
{
  Options o;
  o.rank = examples::default_rank();
  try
  {
    for (int i = 1; i < argc; ++i)
    {
      auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
      if (!std::strcmp(argv[i], "--n"))
        o.n = std::atoi(next());
      else if (!std::strcmp(argv[i], "--ndofs"))
        o.ndofs = std::strtoull(next(), nullptr, 10);
      else if (!std::strcmp(argv[i], "--degree"))
        o.order = std::atoi(next());
      else if (!std::strcmp(argv[i], "--iterations"))
        o.iterations = std::atoi(next());
      else if (!std::strcmp(argv[i], "--quiet"))
        o.quiet = true;
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
        std::cout << "usage: vector_update [--n cells_per_direction | --ndofs N_per_rank] [--degree P] [--iterations N]\n"
                     "                     [--quiet] [--ranks px,py,pz [--rank r] [--id-file F]] [--native-comm] [--halo exchange|windows] [--comm rccl|windows]\n";
        return !std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h") ? 0 : 2;
      }
    }
    if (o.order < 1 || o.order > PMG_MAX_DEGREE)
      throw std::runtime_error("Unsupported degree");
    examples::select_device(o);
    run(o);
  }
  catch (const std::exception& ex)
  {
    std::cerr << "error: " << ex.what() << "\n";
    return 1;
  }
  return 0;
}
