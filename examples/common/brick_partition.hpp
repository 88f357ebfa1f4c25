// One rank's share of a structured hex mesh of the unit cube: px x py x pz bricks with one layer of
// ghost cells, tensor-product GLL dof numbering, and both sides of every halo list -- what the
// reference's multi-rank drivers get from dolfinx (create_box + ghost_layer_mesh, src/mesh.hpp:16-98;
// create_functionspace; the IndexMap / Scatterer index lists of src/vector.hpp:83-96).  Host only, no
// communication: every rank derives its own brick, its neighbours' bricks and the lists from the
// geometry alone.
//
// Rules (the same as pmg-dolfinx_amd/mesh.py, the Python harness):
//   rank = (rx * py + ry) * pz + rz; the cells of an axis are split into balanced contiguous pieces;
//   a rank owns the dofs of its brick except those on an interface with a LOWER brick;
//   it holds every cell that shares a vertex with its brick, so every owned row is complete locally
//   and only the forward (owner -> ghost) halo is needed (src/mesh.hpp:11-12);
//   local numbering: owned dofs first, lexicographic (x slowest) in the owned dof box, then the
//   ghosts grouped by owner rank (ascending), lexicographic inside each owner's box;
//   cells: owned cells first (lexicographic), then ghost cells.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace examples
{
struct Interval
{
  int lo, hi; // closed
  bool empty() const { return lo > hi; }
  int size() const { return empty() ? 0 : hi - lo + 1; }
};
inline Interval intersect(Interval a, Interval b) { return {a.lo > b.lo ? a.lo : b.lo, a.hi < b.hi ? a.hi : b.hi}; }

struct PartitionLevel
{
  int degree = 0;
  std::int32_t size_local = 0, num_ghosts = 0;
  std::vector<std::int32_t> dofmap;        // [ncells][(P+1)^3] local dof indices
  std::vector<std::int8_t> bc_marker;      // [size_local + num_ghosts]
  std::vector<std::int64_t> local_to_global;
  std::vector<double> x;                   // [size_local + num_ghosts][3] dof coordinates
  std::vector<std::int32_t> neighbors, send_counts, recv_counts;
  std::vector<std::int32_t> send_indices;  // owned local indices grouped by neighbour
  std::vector<std::int32_t> recv_indices;  // ghost positions (relative to size_local), grouped by neighbour
  std::int32_t ndofs() const { return size_local + num_ghosts; }
};

class BrickPartition
{
public:
  std::array<int, 3> n, dims, coords;
  int rank, size;
  std::array<Interval, 3> own, ext; // cells: half-open as [lo, hi + 1) stored closed
  std::int32_t ncells_owned = 0, ncells = 0;
  std::vector<std::array<int, 3>> cell_coords; // global cell coordinates of the local cells
  std::vector<double> xgeom;                    // [npoints][3]
  std::vector<std::int32_t> geom_dofmap;        // [ncells][8], k = i*4 + j*2 + l

  BrickPartition(int n_, std::array<int, 3> dims_, int rank_) : n{n_, n_, n_}, dims(dims_), rank(rank_)
  {
    size = dims[0] * dims[1] * dims[2];
    if (rank < 0 || rank >= size)
      throw std::runtime_error("BrickPartition: rank out of range");
    for (int a = 0; a < 3; ++a)
      if (dims[a] > n[a])
        throw std::runtime_error("BrickPartition: more bricks than cells along an axis");
    coords = rank_coords(rank);
    own = own_cells(coords);
    ext = ext_cells(coords);
    std::vector<std::array<int, 3>> ghosts;
    for (int i = ext[0].lo; i <= ext[0].hi; ++i)
      for (int j = ext[1].lo; j <= ext[1].hi; ++j)
        for (int k = ext[2].lo; k <= ext[2].hi; ++k)
        {
          const bool owned = i >= own[0].lo && i <= own[0].hi && j >= own[1].lo && j <= own[1].hi && k >= own[2].lo
                             && k <= own[2].hi;
          (owned ? cell_coords : ghosts).push_back({i, j, k});
        }
    ncells_owned = (std::int32_t)cell_coords.size();
    cell_coords.insert(cell_coords.end(), ghosts.begin(), ghosts.end());
    ncells = (std::int32_t)cell_coords.size();
    // vertices of the extended brick, lexicographic
    const int vs[3] = {ext[0].size() + 1, ext[1].size() + 1, ext[2].size() + 1};
    xgeom.resize((std::size_t)3 * vs[0] * vs[1] * vs[2]);
    for (int i = 0; i < vs[0]; ++i)
      for (int j = 0; j < vs[1]; ++j)
        for (int k = 0; k < vs[2]; ++k)
        {
          const std::size_t v = ((std::size_t)i * vs[1] + j) * vs[2] + k;
          xgeom[3 * v + 0] = (double)(ext[0].lo + i) / n[0];
          xgeom[3 * v + 1] = (double)(ext[1].lo + j) / n[1];
          xgeom[3 * v + 2] = (double)(ext[2].lo + k) / n[2];
        }
    geom_dofmap.resize((std::size_t)8 * ncells);
    for (std::int32_t c = 0; c < ncells; ++c)
    {
      const int lx = cell_coords[c][0] - ext[0].lo, ly = cell_coords[c][1] - ext[1].lo, lz = cell_coords[c][2] - ext[2].lo;
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          for (int l = 0; l < 2; ++l)
            geom_dofmap[(std::size_t)8 * c + i * 4 + j * 2 + l] = ((lx + i) * vs[1] + (ly + j)) * vs[2] + lz + l;
    }
  }

  std::int64_t global_ndofs(int P) const
  {
    return (std::int64_t)(n[0] * P + 1) * (n[1] * P + 1) * (n[2] * P + 1);
  }

  /// gll: the P + 1 GLL points on [0, 1] (pmg_gll_table)
  PartitionLevel level(int P, const std::vector<double>& gll) const
  {
    PartitionLevel lv;
    lv.degree = P;
    const int nd = P + 1;
    const std::int64_t G[3] = {n[0] * P + 1, n[1] * P + 1, n[2] * P + 1};
    const auto Bme = local_dof_box(coords, P), Ome = owned_dof_box(coords, P);
    const int bs[3] = {Bme[0].size(), Bme[1].size(), Bme[2].size()};
    std::vector<std::int32_t> lid((std::size_t)bs[0] * bs[1] * bs[2], -1);
    auto at = [&](int gx, int gy, int gz) -> std::int32_t& {
      return lid[((std::size_t)(gx - Bme[0].lo) * bs[1] + (gy - Bme[1].lo)) * bs[2] + (gz - Bme[2].lo)];
    };
    std::int32_t next = 0;
    for (int i = Ome[0].lo; i <= Ome[0].hi; ++i)
      for (int j = Ome[1].lo; j <= Ome[1].hi; ++j)
        for (int k = Ome[2].lo; k <= Ome[2].hi; ++k)
          at(i, j, k) = next++;
    lv.size_local = next;
    std::vector<std::vector<std::int32_t>> sends;
    for (int q = 0; q < size; ++q)
    {
      if (q == rank)
        continue;
      const auto qc = rank_coords(q);
      const auto Oq = owned_dof_box(qc, P), Bq = local_dof_box(qc, P);
      std::array<Interval, 3> rbox, sbox;
      std::int64_t nrecv = 1, nsend = 1;
      for (int a = 0; a < 3; ++a)
      {
        rbox[a] = intersect(Oq[a], Bme[a]);
        sbox[a] = intersect(Ome[a], Bq[a]);
        nrecv *= rbox[a].size();
        nsend *= sbox[a].size();
      }
      if (nrecv == 0 && nsend == 0)
        continue;
      lv.neighbors.push_back(q);
      lv.recv_counts.push_back((std::int32_t)nrecv);
      lv.send_counts.push_back((std::int32_t)nsend);
      if (nrecv)
        for (int i = rbox[0].lo; i <= rbox[0].hi; ++i)
          for (int j = rbox[1].lo; j <= rbox[1].hi; ++j)
            for (int k = rbox[2].lo; k <= rbox[2].hi; ++k)
              at(i, j, k) = next++;
      if (nsend)
        for (int i = sbox[0].lo; i <= sbox[0].hi; ++i)
          for (int j = sbox[1].lo; j <= sbox[1].hi; ++j)
            for (int k = sbox[2].lo; k <= sbox[2].hi; ++k)
              lv.send_indices.push_back(at(i, j, k));
    }
    lv.num_ghosts = next - lv.size_local;
    for (std::int32_t v : lid)
      if (v < 0)
        throw std::runtime_error("BrickPartition: the ghost layer reaches a dof with no owner (brick too thin?)");
    lv.recv_indices.resize(lv.num_ghosts);
    for (std::int32_t g = 0; g < lv.num_ghosts; ++g)
      lv.recv_indices[g] = g;
    // dofmap, t = a*nd^2 + b*nd + c (src/laplacian.hpp:173)
    lv.dofmap.resize((std::size_t)ncells * nd * nd * nd);
    for (std::int32_t c = 0; c < ncells; ++c)
    {
      std::int32_t* d = lv.dofmap.data() + (std::size_t)c * nd * nd * nd;
      for (int a = 0; a < nd; ++a)
        for (int b = 0; b < nd; ++b)
          for (int e = 0; e < nd; ++e)
            d[(a * nd + b) * nd + e] = at(cell_coords[c][0] * P + a, cell_coords[c][1] * P + b, cell_coords[c][2] * P + e);
    }
    // local -> global, Dirichlet marker on the whole boundary, coordinates
    lv.local_to_global.resize(next);
    lv.bc_marker.assign(next, 0);
    lv.x.resize((std::size_t)3 * next);
    std::vector<double> line[3];
    for (int a = 0; a < 3; ++a)
    {
      line[a].resize(G[a]);
      for (int c = 0; c < n[a]; ++c)
        for (int i = 0; i < nd; ++i)
          line[a][c * P + i] = (c + gll[i]) / n[a];
    }
    for (int i = Bme[0].lo; i <= Bme[0].hi; ++i)
      for (int j = Bme[1].lo; j <= Bme[1].hi; ++j)
        for (int k = Bme[2].lo; k <= Bme[2].hi; ++k)
        {
          const std::int32_t l = at(i, j, k);
          lv.local_to_global[l] = ((std::int64_t)i * G[1] + j) * G[2] + k;
          lv.bc_marker[l] = (i == 0 || j == 0 || k == 0 || i == G[0] - 1 || j == G[1] - 1 || k == G[2] - 1) ? 1 : 0;
          lv.x[3 * (std::size_t)l + 0] = line[0][i];
          lv.x[3 * (std::size_t)l + 1] = line[1][j];
          lv.x[3 * (std::size_t)l + 2] = line[2][k];
        }
    return lv;
  }

  std::array<int, 3> rank_coords(int r) const { return {r / (dims[1] * dims[2]), (r / dims[2]) % dims[1], r % dims[2]}; }

private:
  std::array<Interval, 3> own_cells(const std::array<int, 3>& c) const
  {
    std::array<Interval, 3> o;
    for (int a = 0; a < 3; ++a)
      o[a] = {(int)((long long)n[a] * c[a] / dims[a]), (int)((long long)n[a] * (c[a] + 1) / dims[a]) - 1};
    return o;
  }
  std::array<Interval, 3> ext_cells(const std::array<int, 3>& c) const
  {
    auto o = own_cells(c);
    for (int a = 0; a < 3; ++a)
      o[a] = {o[a].lo > 0 ? o[a].lo - 1 : 0, o[a].hi + 1 < n[a] ? o[a].hi + 1 : n[a] - 1};
    return o;
  }
  std::array<Interval, 3> owned_dof_box(const std::array<int, 3>& c, int P) const
  {
    auto o = own_cells(c);
    std::array<Interval, 3> d;
    for (int a = 0; a < 3; ++a)
      d[a] = {o[a].lo * P + (c[a] > 0 ? 1 : 0), (o[a].hi + 1) * P};
    return d;
  }
  std::array<Interval, 3> local_dof_box(const std::array<int, 3>& c, int P) const
  {
    auto e = ext_cells(c);
    std::array<Interval, 3> d;
    for (int a = 0; a < 3; ++a)
      d[a] = {e[a].lo * P, (e[a].hi + 1) * P};
    return d;
  }
};

/// Host-only consistency check of a partition: every global dof has exactly one owner, and what
/// rank p sends to q is, entry by entry, what q expects to receive from p (same global dofs, same
/// order).  Returns an empty string or the first inconsistency.
inline std::string check_partition(int n, std::array<int, 3> dims, int P, const std::vector<double>& gll)
{
  const int size = dims[0] * dims[1] * dims[2];
  std::vector<PartitionLevel> lv;
  std::int64_t owned = 0, global = 0;
  for (int r = 0; r < size; ++r)
  {
    BrickPartition part(n, dims, r);
    lv.push_back(part.level(P, gll));
    owned += lv.back().size_local;
    global = part.global_ndofs(P);
  }
  if (owned != global)
    return "owned dofs do not add up to the global count";
  for (int p = 0; p < size; ++p)
  {
    std::size_t so = 0;
    for (std::size_t i = 0; i < lv[p].neighbors.size(); ++i)
    {
      const int q = lv[p].neighbors[i];
      std::size_t ro = 0, j = 0;
      for (; j < lv[q].neighbors.size() && lv[q].neighbors[j] != p; ++j)
        ro += lv[q].recv_counts[j];
      if (j == lv[q].neighbors.size() || lv[q].recv_counts[j] != lv[p].send_counts[i])
        return "send and receive counts of a pair of ranks differ";
      for (int k = 0; k < lv[p].send_counts[i]; ++k)
      {
        const std::int64_t gs = lv[p].local_to_global[lv[p].send_indices[so + k]];
        const std::int64_t gr = lv[q].local_to_global[lv[q].size_local + lv[q].recv_indices[ro + k]];
        if (gs != gr)
          return "a sent dof is not the dof the receiver expects at that position";
      }
      so += lv[p].send_counts[i];
    }
  }
  return "";
}
} // namespace examples
