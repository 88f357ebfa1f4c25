// One process per GPU and brick: what the reference's drivers get from MPI_Init / mpirun
// (examples/pmg/submit.sh:29, examples/pmg/select_gpu.sh).  Rank from --rank or RANK /
// OMPI_COMM_WORLD_RANK / PMI_RANK, the GPU from LOCAL_RANK, the communicator id from rank 0
// through a file (--id-file); examples/pmg/run_ranks.sh launches the processes.
#pragma once
#include "pmg_amd.hpp"

#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <initializer_list>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace examples
{
struct RankOptions
{
  int rank = 0;
  std::array<int, 3> ranks = {1, 1, 1};
  bool native_comm = false; // one rank through the RCCL communicator anyway
  bool windows = false;     // --halo windows: the halo as stores into the neighbours' windows, RCCL for the reductions
  bool window_comm = false; // --comm windows: no RCCL at all -- reductions and halo through windows
  pmg_amd::Halo halo() const { return windows ? pmg_amd::Halo::windows : pmg_amd::Halo::exchange; }
  std::string id_file = "/tmp/pmg_amd_comm_id";
  int size() const { return ranks[0] * ranks[1] * ranks[2]; }
};

inline int env_int(std::initializer_list<const char*> names, int fallback)
{
  for (const char* nm : names)
    if (const char* v = std::getenv(nm))
      return std::atoi(v);
  return fallback;
}

inline std::array<int, 3> parse3(const char* s)
{
  std::array<int, 3> r = {1, 1, 1};
  if (std::sscanf(s, "%d,%d,%d", &r[0], &r[1], &r[2]) != 3 || r[0] < 1 || r[1] < 1 || r[2] < 1)
    throw std::runtime_error("expected px,py,pz");
  return r;
}

inline int default_rank() { return env_int({"RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK"}, 0); }

/// Select this rank's GPU (before anything else touches the device).
inline void select_device(const RankOptions& o)
{
  if (o.rank < 0 || o.rank >= o.size())
    throw std::runtime_error("rank out of range for --ranks");
  int ndev = 0;
  pmg_amd::hip_check(hipGetDeviceCount(&ndev), "hipGetDeviceCount");
  if (ndev < 1)
    throw std::runtime_error("no GPU");
  pmg_amd::hip_check(hipSetDevice(env_int({"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK"}, o.rank) % ndev), "hipSetDevice");
}

/// Every rank's fixed-size record in rank order, through files next to the id file (rank r writes <id-file>.w<r>).
template <typename R>
std::vector<R> file_allgather(const RankOptions& o, const R& mine)
{
  static_assert(std::is_trivially_copyable_v<R>);
  const auto started = std::filesystem::file_time_type::clock::now() - std::chrono::seconds(30);
  auto name = [&](int r) { return o.id_file + ".w" + std::to_string(r); };
  {
    const std::string tmp = name(o.rank) + ".tmp";
    {
      std::ofstream f(tmp, std::ios::binary);
      f.write(reinterpret_cast<const char*>(&mine), sizeof(R));
    }
    std::rename(tmp.c_str(), name(o.rank).c_str()); // atomic: readers never see a partial record
  }
  std::vector<R> all((std::size_t)o.size());
  for (int r = 0; r < o.size(); ++r)
    for (int tries = 0;; ++tries)
    {
      std::error_code ec;
      const auto written = std::filesystem::last_write_time(name(r), ec);
      if (!ec && written >= started)
      {
        std::ifstream f(name(r), std::ios::binary);
        if (f && f.read(reinterpret_cast<char*>(&all[(std::size_t)r]), sizeof(R)))
          break;
      }
      if (tries > 600)
        throw std::runtime_error("timed out waiting for rank " + std::to_string(r) + "'s record in " + name(r));
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  return all;
}

/// The library's communicator over all ranks (null on one rank without --native-comm): RCCL, or with --comm windows
/// the communicator made of windows (no transport library; the ranks may then even share a GPU).
inline std::shared_ptr<const pmg_amd::Communicator> bootstrap(const RankOptions& o)
{
  const int size = o.size();
  if (size == 1 && !o.native_comm)
    return nullptr;
  if (o.window_comm)
  {
    using WH = pmg_amd::Communicator::WindowHandle;
    auto comm = std::make_shared<const pmg_amd::Communicator>(
        o.rank, size, [&](const WH& mine) { return file_allgather<WH>(o, mine); });
    (void)comm->allgather(o.rank); // every rank has read every record: the files can go
    std::remove((o.id_file + ".w" + std::to_string(o.rank)).c_str());
    return comm;
  }
  std::array<char, PMG_COMM_ID_BYTES> id{};
  // A file left by an earlier run must not hand a stale id to this one: rank 0 removes it before it creates the
  // new id, and the other ranks accept only a file written after they started (minus a margin for launchers
  // that start rank 0 first).
  const auto started = std::filesystem::file_time_type::clock::now() - std::chrono::seconds(30);
  if (o.rank == 0)
  {
    std::remove(o.id_file.c_str());
    id = pmg_amd::Communicator::unique_id();
    const std::string tmp = o.id_file + ".tmp";
    {
      std::ofstream f(tmp, std::ios::binary);
      f.write(id.data(), id.size());
    }
    std::rename(tmp.c_str(), o.id_file.c_str()); // atomic: readers never see a partial id
  }
  else
  {
    for (int tries = 0;; ++tries)
    {
      std::error_code ec;
      const auto written = std::filesystem::last_write_time(o.id_file, ec);
      if (!ec && written >= started)
      {
        std::ifstream f(o.id_file, std::ios::binary);
        if (f && f.read(id.data(), id.size()))
          break;
      }
      if (tries > 600)
        throw std::runtime_error("timed out waiting for the communicator id in " + o.id_file);
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  auto comm = std::make_shared<const pmg_amd::Communicator>(o.rank, size, id); // collective: every rank has read the id
  if (o.rank == 0)
    std::remove(o.id_file.c_str()); // a finished bootstrap leaves no file behind
  return comm;
}
} // namespace examples
