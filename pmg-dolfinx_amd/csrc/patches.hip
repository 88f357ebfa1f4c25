// Host-side construction of the cell patches (see patches.hpp).  Set-up only.
#include "patches.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>

namespace pmg
{
long long g_merge_below = -1; // pmg_set_merge_threshold; < 0 = the measured defaults
namespace
{
// Rank transform of one coordinate: cluster values closer than tol, return the
// cluster index of every entry and the number of clusters.
int rank_axis(const std::vector<float>& v, float tol, std::vector<int32_t>& idx)
{
  const size_t n = v.size();
  std::vector<int32_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return v[a] < v[b]; });
  idx.assign(n, 0);
  int ncl = 0;
  float last = 0.f;
  for (size_t i = 0; i < n; ++i)
  {
    float x = v[order[i]];
    if (i == 0 || x - last > tol)
    {
      ++ncl;
      last = x;
    }
    idx[order[i]] = ncl - 1;
  }
  return ncl;
}

inline uint64_t spread3(uint32_t v) // 21 bits -> every third bit
{
  uint64_t x = v & 0x1fffff;
  x = (x | x << 32) & 0x1f00000000ffffULL;
  x = (x | x << 16) & 0x1f0000ff0000ffULL;
  x = (x | x << 8) & 0x100f00f00f00f00fULL;
  x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
  x = (x | x << 2) & 0x1249249249249249ULL;
  return x;
}

// Group the cells of one list into patches of <= K cells; returns groups of cell ids.
void group_cells(const int32_t* cells, int32_t n, const float* centroid, PatchShape shp,
                 std::vector<std::vector<int32_t>>& groups)
{
  groups.clear();
  if (n == 0)
    return;
  const int K = shp.K();
  if (K == 1)
  {
    for (int32_t i = 0; i < n; ++i)
      groups.push_back({cells[i]});
    return;
  }
  std::vector<float> c[3];
  float lo[3], hi[3];
  for (int a = 0; a < 3; ++a)
  {
    c[a].resize(n);
    lo[a] = 1e30f;
    hi[a] = -1e30f;
    for (int32_t i = 0; i < n; ++i)
    {
      float x = centroid[3 * (size_t)cells[i] + a];
      c[a][i] = x;
      lo[a] = std::min(lo[a], x);
      hi[a] = std::max(hi[a], x);
    }
  }
  // tensor-grid detection: few distinct centroid coordinates per axis
  std::vector<int32_t> gi[3];
  int ncl[3];
  bool tensor = true;
  double cap = 4.0 * std::cbrt((double)n) + 8.0;
  for (int a = 0; a < 3; ++a)
  {
    float ext = std::max(hi[a] - lo[a], 1e-30f);
    ncl[a] = rank_axis(c[a], 1e-5f * ext, gi[a]);
    if (ncl[a] > cap * 4)
      tensor = false;
  }
  if (tensor && (double)ncl[0] * ncl[1] * ncl[2] > 64.0 * n + 4096.0)
    tensor = false;

  std::vector<int32_t> order(n);
  std::iota(order.begin(), order.end(), 0);
  if (tensor)
  {
    // sort by (block key, position inside the block); equal keys form a patch
    // blocks of at most (bx, by, bz) cells, balanced: ceil(n / b) blocks per axis whose sizes differ by at most
    // one (64 cells in blocks of at most 7: four of 7 and six of 6, not nine of 7 and one of 1)
    const int bmax[3] = {shp.bx, shp.by, shp.bz};
    int nblk[3];
    for (int a = 0; a < 3; ++a)
      nblk[a] = (ncl[a] + bmax[a] - 1) / bmax[a];
    auto block = [&](int a, int32_t g) { return (uint64_t)(((long long)g * nblk[a]) / ncl[a]); };
    std::vector<uint64_t> key(n);
    for (int32_t i = 0; i < n; ++i)
    {
      uint64_t bx = block(0, gi[0][i]), by = block(1, gi[1][i]), bz = block(2, gi[2][i]);
      key[i] = (bx << 42) | (by << 21) | bz;
    }
    std::sort(order.begin(), order.end(),
              [&](int32_t a, int32_t b)
              {
                if (key[a] != key[b])
                  return key[a] < key[b];
                // z-layer major inside a block: a round of the kernel takes one layer
                if (gi[2][a] != gi[2][b])
                  return gi[2][a] < gi[2][b];
                if (gi[0][a] != gi[0][b])
                  return gi[0][a] < gi[0][b];
                return gi[1][a] < gi[1][b];
              });
    for (int32_t i = 0; i < n;)
    {
      int32_t j = i;
      std::vector<int32_t> g;
      while (j < n && key[order[j]] == key[order[i]] && (int)g.size() < K)
        g.push_back(cells[order[j++]]);
      // thin cell sets (the shell of boundary / ghost cells of a brick) leave most
      // blocks partly empty: fold a partial block into its predecessor in key order
      // (the next block along z, then y) while the patch still holds <= K cells
      if (!groups.empty() && groups.back().size() + g.size() <= (size_t)K && (int)g.size() < K)
        groups.back().insert(groups.back().end(), g.begin(), g.end());
      else
        groups.push_back(std::move(g));
      i = j;
    }
  }
  else
  {
    // Morton order of the quantised centroids, chunks of K
    std::vector<uint64_t> key(n);
    for (int32_t i = 0; i < n; ++i)
    {
      uint64_t k = 0;
      for (int a = 0; a < 3; ++a)
      {
        float ext = std::max(hi[a] - lo[a], 1e-30f);
        uint32_t q = (uint32_t)std::min(2097151.0f, (c[a][i] - lo[a]) / ext * 2097151.0f);
        k |= spread3(q) << (2 - a);
      }
      key[i] = k;
    }
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
    for (int32_t i = 0; i < n; i += K)
    {
      std::vector<int32_t> g;
      for (int32_t j = i; j < n && j < i + K; ++j)
        g.push_back(cells[order[j]]);
      groups.push_back(std::move(g));
    }
  }
}
} // namespace

int build_patch_plan(PatchPlan& plan, int P, int32_t ncells, const int32_t* dofmap,
                     const int8_t* bc, int32_t ndofs, const float* centroid,
                     const int32_t* lcells, int32_t n_l, const int32_t* bcells, int32_t n_b)
{
  const PatchShape shp = patch_shape(P);
  const int K = shp.K();
  const int nd = P + 1, N = nd * nd * nd;
  plan = PatchPlan();
  plan.K = K;
  plan.N = N;

  struct Tmp
  {
    std::vector<int32_t> cells;
    std::vector<int32_t> dofs; // sorted unique
    std::vector<uint16_t> lmap;
    int colour = 0;
    int launch = 0;
    int part = 0; // interior set only: 0 / 1 = first / second half (own stream), 2 = separator (joins the boundary set)
  };
  std::vector<Tmp> tmp;
  int set_first[3] = {0, 0, 0};
  for (int set = 0; set < 2; ++set)
  {
    std::vector<std::vector<int32_t>> groups;
    group_cells(set == 0 ? lcells : bcells, set == 0 ? n_l : n_b, centroid, shp, groups);
    // a group whose cells have more than max_m distinct dofs (irregular meshes,
    // Morton chunks) is halved until it fits the kernel's LDS capacity
    std::vector<std::vector<int32_t>> work(groups.rbegin(), groups.rend());
    std::vector<int32_t> scratch;
    while (!work.empty())
    {
      std::vector<int32_t> g = std::move(work.back());
      work.pop_back();
      scratch.clear();
      for (int32_t cell : g)
      {
        if (cell < 0 || cell >= ncells)
          return fail(PMG_ERR_INVALID, "cell %d out of range", cell);
        scratch.insert(scratch.end(), dofmap + (size_t)cell * N, dofmap + (size_t)(cell + 1) * N);
      }
      std::sort(scratch.begin(), scratch.end());
      size_t m = std::unique(scratch.begin(), scratch.end()) - scratch.begin();
      if ((int)m > shp.max_m && g.size() > 1)
      {
        size_t h = g.size() / 2;
        work.emplace_back(g.begin() + h, g.end());
        work.emplace_back(g.begin(), g.begin() + h);
        continue;
      }
      if ((int)m > shp.max_m)
        return fail(PMG_ERR_INVALID, "internal: one cell exceeds the patch capacity");
      Tmp t;
      t.cells = std::move(g);
      tmp.push_back(std::move(t));
    }
    set_first[set + 1] = (int)tmp.size();
  }
  const int np = (int)tmp.size();

  // dof lists and local maps
  for (auto& t : tmp)
  {
    t.dofs.reserve((size_t)t.cells.size() * N);
    for (int32_t cell : t.cells)
    {
      const int32_t* dm = dofmap + (size_t)cell * N;
      for (int k = 0; k < N; ++k)
      {
        if (dm[k] < 0 || dm[k] >= ndofs)
          return fail(PMG_ERR_INVALID, "dofmap entry %d of cell %d out of range [0,%d)", dm[k], cell,
                      ndofs);
        t.dofs.push_back(dm[k]);
      }
    }
    std::sort(t.dofs.begin(), t.dofs.end());
    t.dofs.erase(std::unique(t.dofs.begin(), t.dofs.end()), t.dofs.end());
    t.lmap.assign((size_t)K * N, 0);
    for (size_t s = 0; s < t.cells.size(); ++s)
    {
      const int32_t* dm = dofmap + (size_t)t.cells[s] * N;
      for (int k = 0; k < N; ++k)
        t.lmap[s * N + table_index(nd, k)]
            = (uint16_t)(std::lower_bound(t.dofs.begin(), t.dofs.end(), dm[k]) - t.dofs.begin());
    }
    plan.max_M = std::max(plan.max_M, (int)t.dofs.size());
  }

  // ---- two halves of the interior, one per stream (see PatchPlan::launch_stream) ----
  // OFF unless PMG_APPLY_STREAMS=1 (levels of >= 2048 interior patches whose colours are launched one by one) or =2
  // (small levels too: tests).  Measured, round 4 (profiles/kernel_tuning_r04.md): two INDEPENDENT half-size operators
  // on two streams take 8 % less than back to back -- each fills the other's launch tails -- but inside one operator
  // the fork, the ordering event and the join cost more than that: 440 against 431 us at degree 4, 64^3; 614 against
  // 585 at degree 2, 128^3; the same inside a replayed graph.
  bool split = false;
  {
    const char* e = std::getenv("PMG_APPLY_STREAMS");
    const bool allowed = e && (e[0] == '1' || e[0] == '2');
    long long interior_pdofs = 0;
    for (int p = set_first[0]; p < set_first[1]; ++p)
      interior_pdofs += (long long)tmp[p].dofs.size();
    const long long merge_below = g_merge_below >= 0 ? g_merge_below : (long long)(P <= 1 ? 2 : 6) << 20;
    const int ni = set_first[1] - set_first[0];
    const int min_patches = e && e[0] == '2' ? 16 : 2048;
    if (allowed && interior_pdofs > merge_below && ni >= min_patches)
    {
      // cut at the middle patch position along the axis with the most distinct positions
      std::vector<float> pc[3];
      for (int a = 0; a < 3; ++a)
        pc[a].resize(ni);
      for (int p = 0; p < ni; ++p)
        for (int a = 0; a < 3; ++a)
        {
          double v = 0;
          for (int32_t c : tmp[set_first[0] + p].cells)
            v += centroid[3 * (size_t)c + a];
          pc[a][p] = (float)(v / (double)tmp[set_first[0] + p].cells.size());
        }
      int best = -1, best_n = 0;
      float cut = 0.f;
      for (int a = 0; a < 3; ++a)
      {
        std::vector<float> u(pc[a]);
        std::sort(u.begin(), u.end());
        const float ext = std::max(u.back() - u.front(), 1e-30f);
        std::vector<float> d;
        for (float v : u)
          if (d.empty() || v - d.back() > 1e-4f * ext)
            d.push_back(v);
        if ((int)d.size() > best_n)
        {
          best_n = (int)d.size();
          best = a;
          cut = d.size() > 1 ? 0.5f * (d[d.size() / 2 - 1] + d[d.size() / 2]) : d[0];
        }
      }
      if (best >= 0 && best_n >= 4)
      {
        int n1 = 0;
        for (int p = 0; p < ni; ++p)
          n1 += (tmp[set_first[0] + p].part = pc[best][p] < cut ? 0 : 1);
        split = n1 > 0 && n1 < ni;
      }
    }
  }

  // greedy colouring per set: a patch takes the lowest colour none of its dofs has seen
  std::vector<int> ncolours(2, 0);
  {
    std::vector<uint64_t> mask(ndofs);
    for (int set = 0; set < 2; ++set)
    {
      std::fill(mask.begin(), mask.end(), 0);
      for (int p = set_first[set]; p < set_first[set + 1]; ++p)
      {
        uint64_t used = 0;
        for (int32_t d : tmp[p].dofs)
          used |= mask[d];
        int col = 0;
        while (col < 64 && (used >> col) & 1)
          ++col;
        if (col >= 64)
          return fail(PMG_ERR_INVALID, "patch colouring needs more than 64 colours");
        tmp[p].colour = col;
        ncolours[set] = std::max(ncolours[set], col + 1);
        for (int32_t d : tmp[p].dofs)
          mask[d] |= (uint64_t)1 << col;
      }
    }
  }
  // The two halves meet in a layer of patches on either side of the cut.  The second half launches the colours of ITS
  // layer first, the first half the colours of its layer last, and the first of those waits (an event) for the last of
  // the others: every shared dof is then touched by the second half strictly before the first half, which is the order
  // the launch indices (and with them the PD_ACC flags) state.  On a tensor grid the two layers have opposite parity
  // along the cut axis, i.e. disjoint colours, and the wait sits in the middle of both sequences; if the colouring does
  // not separate them that way (kb >= ka below), the level keeps the single sequence.
  std::vector<int> seq[2]; // colours of each half in launch order
  int ka = 0, kb = -1;
  if (split)
  {
    std::vector<char> in_part[2] = {std::vector<char>(ndofs, 0), std::vector<char>(ndofs, 0)};
    for (int p = set_first[0]; p < set_first[1]; ++p)
      for (int32_t d : tmp[p].dofs)
        in_part[tmp[p].part][d] = 1;
    std::vector<char> present[2] = {std::vector<char>(ncolours[0], 0), std::vector<char>(ncolours[0], 0)};
    std::vector<char> edge[2] = {std::vector<char>(ncolours[0], 0), std::vector<char>(ncolours[0], 0)};
    for (int p = set_first[0]; p < set_first[1]; ++p)
    {
      const int h = tmp[p].part;
      present[h][tmp[p].colour] = 1;
      for (int32_t d : tmp[p].dofs)
        if (in_part[1 - h][d])
        {
          edge[h][tmp[p].colour] = 1;
          break;
        }
    }
    for (int c = 0; c < ncolours[0]; ++c) // first half: colours without layer patches first
      if (present[0][c] && !edge[0][c])
        seq[0].push_back(c);
    ka = (int)seq[0].size();
    for (int c = 0; c < ncolours[0]; ++c)
      if (present[0][c] && edge[0][c])
        seq[0].push_back(c);
    for (int c = 0; c < ncolours[0]; ++c) // second half: colours with layer patches first
      if (present[1][c] && edge[1][c])
        seq[1].push_back(c);
    kb = (int)seq[1].size() - 1;
    for (int c = 0; c < ncolours[0]; ++c)
      if (present[1][c] && !edge[1][c])
        seq[1].push_back(c);
    if (kb >= ka || kb < 0)
      split = false;
  }
  // launch index = position in stream order: lcells colours (two halves: position q of the first half, position q of
  // the second half, position q + 1 ...; a half that has run out of colours leaves empty launches), then bcells colours
  int nl_interior = ncolours[0];
  if (split)
  {
    const int nq = (int)std::max(seq[0].size(), seq[1].size());
    nl_interior = 2 * nq;
    std::vector<int> pos[2] = {std::vector<int>(ncolours[0], 0), std::vector<int>(ncolours[0], 0)};
    for (int h = 0; h < 2; ++h)
      for (size_t q = 0; q < seq[h].size(); ++q)
        pos[h][seq[h][q]] = (int)q;
    for (int p = set_first[0]; p < set_first[1]; ++p)
      tmp[p].launch = 2 * pos[tmp[p].part][tmp[p].colour] + tmp[p].part;
  }
  else
    for (int p = set_first[0]; p < set_first[1]; ++p)
      tmp[p].launch = tmp[p].colour;
  for (int p = set_first[1]; p < np; ++p)
    tmp[p].launch = nl_interior + tmp[p].colour;
  const int nlaunch = nl_interior + ncolours[1];
  plan.n_launch_l = nl_interior;
  std::vector<int32_t> order(np);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(),
                   [&](int32_t a, int32_t b) { return tmp[a].launch < tmp[b].launch; });
  plan.launch_first.assign(nlaunch, 0);
  plan.launch_count.assign(nlaunch, 0);
  for (int i = 0; i < np; ++i)
    plan.launch_count[tmp[order[i]].launch]++;
  for (int l = 1; l < nlaunch; ++l)
    plan.launch_first[l] = plan.launch_first[l - 1] + plan.launch_count[l - 1];

  // first launch that touches each dof
  std::vector<int32_t> first(ndofs, INT32_MAX);
  for (int p = 0; p < np; ++p)
    for (int32_t d : tmp[p].dofs)
      first[d] = std::min(first[d], (int32_t)tmp[p].launch);

  // emit in launch order, de-duplicating the local maps
  plan.npatch = np;
  plan.pcell.assign((size_t)np * K, -1);
  plan.pncell.assign(np, 0);
  plan.poff.assign(np + 1, 0);
  plan.lmap_id.assign(np, 0);
  std::map<std::vector<uint16_t>, int32_t> uniq;
  size_t total = 0;
  for (int i = 0; i < np; ++i)
    total += tmp[order[i]].dofs.size();
  plan.pdofs.reserve(total);
  for (int i = 0; i < np; ++i)
  {
    Tmp& t = tmp[order[i]];
    for (size_t s = 0; s < t.cells.size(); ++s)
      plan.pcell[(size_t)i * K + s] = t.cells[s];
    plan.pncell[i] = (int32_t)t.cells.size();
    for (int32_t d : t.dofs)
    {
      uint32_t v = (uint32_t)d;
      if ((uint32_t)d > PD_MASK)
        return fail(PMG_ERR_INVALID, "dof index too large for the patch encoding");
      if (bc[d])
        v |= PD_BC;
      if (first[d] != t.launch)
        v |= PD_ACC;
      plan.pdofs.push_back(v);
    }
    plan.poff[i + 1] = (int32_t)plan.pdofs.size();
    auto it = uniq.find(t.lmap);
    if (it == uniq.end())
    {
      it = uniq.emplace(t.lmap, (int32_t)uniq.size()).first;
      plan.lmaps.insert(plan.lmaps.end(), t.lmap.begin(), t.lmap.end());
    }
    plan.lmap_id[i] = it->second;
  }
  plan.nuniq = (int)uniq.size();
  // Launch list.  The boundary colours are always merged into one atomic launch (a thin
  // shell: many small launches otherwise).  The interior colours are merged as well when
  // the level is small: eight launches that cannot fill the GPU are launch- and ramp-bound,
  // one launch that adds with atomics is not, as long as the atomics stay few.  Measured
  // (stiffness kernel alone, colours -> merged): P=1 16^3 27 -> 4 us, 64^3 43 -> 23 us;
  // P=2 32^3 79 -> 13, 48^3 90 -> 47, 64^3 (2.8 M patch dofs) 110 -> 116 (whole apply with its zero-fill, timed
  // in the stream: 110 -> 103), 96^3 (9.5 M) 295 -> 286, 128^3 677 -> 894; P=1 128^3 213 -> 227; P=3 32^3 59 -> 31,
  // 48^3 100 -> 96; P=4 24^3 102 -> 29, 32^3 111 -> 66, 40^3 (5.3 M) 150 -> 126, 48^3 (9.2 M)
  // 206 -> 219, 64^3 427 -> 570; P=6 24^3 141 -> 94; P=8 16^3 174 -> 73.  Hence the
  // thresholds below, in patch dofs (= atomically added values) of the interior list.
  {
    const int nl = nl_interior;
    const int32_t bfirst = nl < nlaunch ? plan.launch_first[nl] : np;
    const long long interior_pdofs = plan.poff[bfirst];
    const long long merge_below = g_merge_below >= 0 ? g_merge_below : (long long)(P <= 1 ? 2 : 6) << 20;
    const bool merge_interior = nl > 1 && interior_pdofs <= merge_below;
    std::vector<int32_t> lf, lc;
    if (merge_interior)
    {
      lf.push_back(0);
      lc.push_back(bfirst);
      plan.n_plain = 0;
      plan.n_launch_l = 1;
    }
    else
    {
      lf.assign(plan.launch_first.begin(), plan.launch_first.begin() + nl);
      lc.assign(plan.launch_count.begin(), plan.launch_count.begin() + nl);
      plan.n_plain = nl;
      plan.n_launch_l = nl;
    }
    if (bfirst < np)
    {
      lf.push_back(bfirst);
      lc.push_back(np - bfirst);
    }
    plan.launch_first = lf;
    plan.launch_count = lc;
    if (split && !merge_interior)
    {
      plan.launch_stream.assign(lf.size(), 0);
      for (int l = 0; l < nl; ++l)
        plan.launch_stream[l] = (int8_t)(l % 2);
      plan.launch_signal = 2 * kb + 1; // the second half's last launch with layer patches ...
      plan.launch_wait = 2 * ka;       // ... must be complete before this launch of the first half starts
    }
    // dofs whose first writer is an atomic launch must be zero beforehand
    const int first_atomic_colour = merge_interior ? 0 : nl;
    for (int32_t d = 0; d < ndofs; ++d)
      if (first[d] != INT32_MAX && first[d] >= first_atomic_colour && !bc[d])
        plan.bzero.push_back(d);
  }
  return PMG_OK;
}

// ---- chains of interior patches (patches.hpp, ChainPlan) ----
int build_chain_plan(ChainPlan& cp, const PatchPlan& plan, int32_t ndofs, const int8_t* bc, const float* centroid,
                     int min_chains)
{
  cp = ChainPlan();
  const int K = plan.K;
  // only a level whose interior colours are launched one by one (a merged small level has nothing to gain)
  if (plan.n_plain <= 1 || plan.n_plain != plan.n_launch_l || !plan.launch_stream.empty())
    return PMG_OK;
  int npi = 0; // interior patches: the first npi of the plan
  for (int l = 0; l < plan.n_launch_l; ++l)
    npi += plan.launch_count[l];
  if (npi < 2 || (uint32_t)ndofs > CD_MASK || plan.max_M >= (int)CC_NONE)
    return PMG_OK;

  // patch positions: rank transform of the patch centroids; a tensor grid of patches is required
  std::vector<float> pc[3];
  for (int a = 0; a < 3; ++a)
    pc[a].resize(npi);
  for (int p = 0; p < npi; ++p)
  {
    double v[3] = {0, 0, 0};
    int n = 0;
    for (int s = 0; s < K; ++s)
    {
      const int32_t c = plan.pcell[(size_t)p * K + s];
      if (c < 0)
        continue;
      for (int a = 0; a < 3; ++a)
        v[a] += centroid[3 * (size_t)c + a];
      ++n;
    }
    for (int a = 0; a < 3; ++a)
      pc[a][p] = (float)(v[a] / std::max(n, 1));
  }
  std::vector<int32_t> gi[3];
  int ng[3];
  for (int a = 0; a < 3; ++a)
  {
    const auto mm = std::minmax_element(pc[a].begin(), pc[a].end());
    const float ext = std::max(*mm.second - *mm.first, 1e-30f);
    ng[a] = rank_axis(pc[a], 1e-4f * ext, gi[a]);
  }
  if ((long long)ng[0] * ng[1] * ng[2] != npi)
    return PMG_OK;
  // the chain axis: the one with the fewest positions (ties: z, then y)
  int ax = 2;
  for (int a = 1; a >= 0; --a)
    if (ng[a] < ng[ax])
      ax = a;
  const int o1 = ax == 0 ? 1 : 0, o2 = ax == 2 ? 1 : 2;
  const int nchain = ng[o1] * ng[o2], len = ng[ax];
  std::vector<int32_t> grid((size_t)npi, -1); // [chain][position] -> patch
  for (int p = 0; p < npi; ++p)
  {
    const size_t at = ((size_t)gi[o1][p] * ng[o2] + gi[o2][p]) * len + gi[ax][p];
    if (grid[at] >= 0)
      return PMG_OK; // two patches at one grid position: not a tensor grid
    grid[at] = p;
  }

  // colour the chains (greedy, as for patches) and check that only CONSECUTIVE patches of a chain share dofs
  std::vector<int> colour(nchain, 0);
  int ncol = 0;
  {
    std::vector<uint64_t> mask(ndofs, 0);
    std::vector<int32_t> stamp(ndofs, -1), last(ndofs, 0);
    for (int ch = 0; ch < nchain; ++ch)
    {
      uint64_t used = 0;
      for (int c = 0; c < len; ++c)
      {
        const int p = grid[(size_t)ch * len + c];
        for (int i = plan.poff[p]; i < plan.poff[p + 1]; ++i)
        {
          const int32_t d = (int32_t)(plan.pdofs[i] & PD_MASK);
          if (stamp[d] == ch)
          {
            if (last[d] < c - 1)
              return PMG_OK; // shared by two patches that are not neighbours in the chain
          }
          else
          {
            used |= mask[d];
            stamp[d] = ch;
          }
          last[d] = c;
        }
      }
      int col = 0;
      while (col < 64 && (used >> col) & 1)
        ++col;
      if (col >= 64)
        return PMG_OK;
      colour[ch] = col;
      ncol = std::max(ncol, col + 1);
      for (int c = 0; c < len; ++c)
      {
        const int p = grid[(size_t)ch * len + c];
        for (int i = plan.poff[p]; i < plan.poff[p + 1]; ++i)
          mask[plan.pdofs[i] & PD_MASK] |= (uint64_t)1 << col;
      }
    }
  }
  std::vector<int32_t> count(ncol, 0);
  for (int ch = 0; ch < nchain; ++ch)
    count[colour[ch]]++;
  for (int c = 0; c < ncol; ++c)
    if (count[c] < min_chains)
      return PMG_OK; // one workgroup per chain would leave the GPU partly idle
  // (fewer launches than the patch colours, or the form has no point)
  if (ncol >= plan.n_plain)
    return PMG_OK;

  // first colour that touches each dof
  std::vector<int32_t> first(ndofs, INT32_MAX);
  for (int ch = 0; ch < nchain; ++ch)
    for (int c = 0; c < len; ++c)
    {
      const int p = grid[(size_t)ch * len + c];
      for (int i = plan.poff[p]; i < plan.poff[p + 1]; ++i)
      {
        int32_t& f = first[plan.pdofs[i] & PD_MASK];
        f = std::min(f, (int32_t)colour[ch]);
      }
    }

  // the lists
  cp.cdofs.assign(plan.pdofs.size(), 0);
  cp.ccar.assign(plan.pdofs.size(), CC_NONE);
  for (int ch = 0; ch < nchain; ++ch)
    for (int c = 0; c < len; ++c)
    {
      const int p = grid[(size_t)ch * len + c];
      const int prev = c > 0 ? grid[(size_t)ch * len + c - 1] : -1;
      const int next = c + 1 < len ? grid[(size_t)ch * len + c + 1] : -1;
      // the patch lists are sorted by dof: two-pointer walks against the neighbours' lists
      int ip = prev >= 0 ? plan.poff[prev] : 0, ipe = prev >= 0 ? plan.poff[prev + 1] : 0;
      int in = next >= 0 ? plan.poff[next] : 0, ine = next >= 0 ? plan.poff[next + 1] : 0;
      for (int i = plan.poff[p]; i < plan.poff[p + 1]; ++i)
      {
        const uint32_t d = plan.pdofs[i] & PD_MASK;
        while (ip < ipe && (plan.pdofs[ip] & PD_MASK) < d)
          ++ip;
        while (in < ine && (plan.pdofs[in] & PD_MASK) < d)
          ++in;
        const bool in_prev = ip < ipe && (plan.pdofs[ip] & PD_MASK) == d;
        const bool in_next = in < ine && (plan.pdofs[in] & PD_MASK) == d;
        uint32_t v = d, w = CC_NONE;
        if (bc[d])
        {
          v |= CD_BC;
          if (first[d] == colour[ch] && !in_prev)
            v |= CD_BCFIRST;
        }
        else
        {
          if (in_next)
            v |= CD_SKIP;
          if (in_prev)
            w = (uint32_t)(ip - plan.poff[prev]);
          else if (first[d] < colour[ch])
            w |= CC_ACC;
        }
        cp.cdofs[i] = v;
        cp.ccar[i] = w;
      }
    }

  // chains in colour order
  std::vector<int32_t> order(nchain);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return colour[a] < colour[b]; });
  cp.chain_off.assign(nchain + 1, 0);
  cp.chain_patch.reserve(npi);
  for (int i = 0; i < nchain; ++i)
  {
    for (int c = 0; c < len; ++c)
      cp.chain_patch.push_back(grid[(size_t)order[i] * len + c]);
    cp.chain_off[i + 1] = (int32_t)cp.chain_patch.size();
  }
  cp.launch_count = count;
  cp.launch_first.assign(ncol, 0);
  for (int c = 1; c < ncol; ++c)
    cp.launch_first[c] = cp.launch_first[c - 1] + count[c - 1];
  cp.ok = true;
  return PMG_OK;
}
} // namespace pmg

extern "C" int pmg_set_merge_threshold(long long patch_dofs)
{
  pmg::g_merge_below = patch_dofs;
  return PMG_OK;
}
