// Cell patches: the scatter side of the operator without global atomics.
//
// The reference adds every (cell, dof) contribution to y with a global FP64
// atomicAdd (src/laplacian.hpp:277).  On MI355X float atomics execute at the
// memory side at a fixed request rate (MI355X_MICROARCH.md "Global float
// atomics"); at P = 4 the 10 M atomic requests of one apply take longer than
// streaming the whole 2 GB of operator data.  Instead, cells are grouped into
// compact patches (2x2x2 cells at P = 4); one workgroup applies the operator to
// a whole patch, sums the contributions of its cells in LDS, and writes each
// patch dof once.  Patches are coloured so that no two patches of a colour share
// a dof; colours are launched one after the other on the stream, so the write is
// a plain store for the first patch that touches a dof and a plain
// read-modify-write for the later ones: no global atomics and no zero-fill of y
// (inside a patch the cell sums meet in LDS in arrival order, so two runs agree
// to rounding, not bit for bit).
#pragma once

#include "common.hpp"

#include <cstdint>
#include <vector>

namespace pmg
{
constexpr uint32_t PD_BC = 0x80000000u;  // patch-dof flag: Dirichlet dof
constexpr uint32_t PD_ACC = 0x40000000u; // patch-dof flag: an earlier launch already wrote this dof
constexpr uint32_t PD_MASK = 0x3fffffffu;

// Patch geometry per degree: a block of bx*by*bz cells on a tensor grid (long in
// z, the direction in which a lexicographic dof numbering is contiguous, so the
// patch's x gather / y store move long runs), processed by one workgroup.  max_m bounds the number of distinct dofs of a
// patch (LDS size of the kernel); the builder closes a patch early rather than
// exceed it, so any mesh and any cell order is handled.
struct PatchShape
{
  int bx, by, bz; // cells per patch along x, y, z
  int max_m;      // LDS capacity in dofs
  int K() const { return bx * by * bz; }
};
inline constexpr PatchShape patch_shape(int P)
{
  switch (P)
  {
  case 1:
#ifdef PMG_P1_SHAPE
    return {PMG_P1_SHAPE};
#else
    return {4, 4, 16, 448};   // M = 5*5*17  = 425 (round 3: 21.2 us against 23.3 for 4x4x8 at 64^3)
#endif
  case 2:
#ifdef PMG_P2_SHAPE
    return {PMG_P2_SHAPE};
#else
    return {4, 4, 16, 2688};  // M = 9*9*33  = 2673 (round 3, 64^3 / 128^3: 82 / 580 us against 97 / 660 for 4x4x8,
                              // 134 / 997 for 2x2x8, 82 / 626 for 4x4x32: profiles/kernel_tuning_r03.md section 11)
#endif
  case 3:
#ifdef PMG_P3_SHAPE
    return {PMG_P3_SHAPE};
#else
    return {2, 2, 8, 1280};   // M = 7*7*25  = 1225
#endif
  case 4:
#ifdef PMG_P4_SHAPE
    return {PMG_P4_SHAPE};
#else
    return {2, 2, 8, 2688};   // M = 9*9*33  = 2673
#endif
  case 5:
#ifdef PMG_P5_SHAPE
    return {PMG_P5_SHAPE};
#else
    return {1, 1, 7, 1344};   // M = 6*6*36  = 1296: one item of 7 cells per workgroup (laplacian.hip, Shape)
#endif
  case 6:
#ifdef PMG_P6_SHAPE
    return {PMG_P6_SHAPE};
#else
    return {1, 1, 8, 2432};   // M = 7*7*49  = 2401 (round 3, 43^3: 463 us against 515 for 2x2x2, 478 for 1x2x4 / 1x1x4,
                              // 481 for 1x1x12, 536 for 1x2x8, 630 for 1x1x16)
#endif
  case 7:
#ifdef PMG_P7_SHAPE
    return {PMG_P7_SHAPE};
#else
    return {2, 2, 3, 4976};   // M = 15*15*22 = 4950 (round 3, 36^3: 389 us against 413-424 for 2x2x2, 403 for 1x1x8)
#endif
  default:
#ifdef PMG_P8_SHAPE
    return {PMG_P8_SHAPE};
#else
    return {1, 1, 3, 2048};   // M = 9*9*25  = 2025: one item of 3 cells per workgroup (laplacian.hip, Shape)
#endif
  }
}

// The per-cell tables are stored layer by layer (the stiffness kernel marches along c):
// index c*nd^2 + a*nd + b for the dofmap's t = a*nd^2 + b*nd + c.
inline int table_index(int nd, int t)
{
  const int a = t / (nd * nd), b = (t / nd) % nd, c = t % nd;
  return c * nd * nd + a * nd + b;
}

// Host description of the patches of one operator (both cell lists).
struct PatchPlan
{
  int K = 0, N = 0;
  int npatch = 0;
  std::vector<int32_t> pcell;   // [npatch*K] cell id per slot, -1 = empty slot (cells first)
  std::vector<int32_t> pncell;  // [npatch] number of cells of the patch
  std::vector<int32_t> poff;    // [npatch+1] offsets into pdofs
  std::vector<uint32_t> pdofs;  // patch dof lists (sorted by dof), flags in the top bits
  std::vector<int32_t> lmap_id; // [npatch] index of the patch's local map
  std::vector<uint16_t> lmaps;  // [nuniq][K*N] position of (slot, t) in the patch dof list
  int nuniq = 0;
  // launches: contiguous patch ranges, in stream order; first n_launch_l belong to lcells
  std::vector<int32_t> launch_first, launch_count;
  int n_launch_l = 0;
  int n_plain = 0; // the first n_plain launches store (colours); the rest add with atomics
  // Two halves of the interior on two streams (round 4): launch_stream[l] = 1 for the launches of the second half
  // (empty = everything on the caller's stream), so that each half fills the other's launch tails.  The halves share
  // dofs only across the cut; launch `launch_wait` (first half) must not start before launch `launch_signal` (second
  // half) is complete, which orders every shared dof as the launch indices say (patches.hip).
  std::vector<int8_t> launch_stream;
  int launch_signal = -1, launch_wait = -1;
  int max_M = 0;
  // The boundary cell list is a thin shell: its colours would be many small
  // launches.  They are issued as ONE launch whose patches add their sums to y
  // with atomics (few, in long runs); so are the interior colours of a small level.
  // bzero lists the dofs whose first writer is such a launch, which must be zero
  // beforehand.  The PD_ACC flags still name a unique first patch per dof (the
  // transfers and the Dirichlet rows rely on that).
  std::vector<int32_t> bzero;
};

// What other components (the p-transfer) need to know about an operator's patches.
struct PatchView
{
  int P = 0, K = 0, N = 0, npatch = 0, max_m = 0;
  // host copies
  const std::vector<int32_t>* pcell_h = nullptr;  // [npatch*K]
  const std::vector<int32_t>* pncell_h = nullptr; // [npatch]
  const std::vector<int32_t>* launch_first = nullptr;
  const std::vector<int32_t>* launch_count = nullptr;
  int n_launch_l = 0;
  bool merged = false; // the operator's launches all add with atomics (a small level: PatchPlan::n_plain == 0)
  // device arrays
  const int32_t* pcell = nullptr;
  const int32_t* pncell = nullptr;
  const int32_t* poff = nullptr;
  const uint32_t* pdofs = nullptr;
  const int32_t* lmap_id = nullptr;
  const uint16_t* lmaps = nullptr;
  long long npdofs = 0; // total entries of pdofs
};

// Chains (round 4, second half): the interior patches of a large level strung together along the axis with the fewest
// patch positions (z on a box whose patches are long in z), one PERSISTENT workgroup per chain (stiffness_chain_kernel,
// laplacian.hip).  A chain is walked in order, so the dofs two consecutive patches share never pass through global
// memory: the later patch starts its accumulator from the earlier one's LDS sums ("carry") and the earlier one does not
// store them ("skip").  Chains are coloured like patches (no two chains of a colour share a dof): on a tensor grid FOUR
// launches instead of eight, each workgroup alive for a whole column of patches, the gather of patch c + 1 and the
// write-back of patch c - 1 under the cell loop of patch c.
// Lists parallel to PatchPlan::pdofs (same offsets; only the interior patches' entries are filled):
//   cdofs: dof | CD_BC | CD_SKIP | CD_BCFIRST        ccar: position of the dof in the previous patch of the chain
//   (CC_NONE = none) | CC_ACC (an earlier LAUNCH touched the dof and no carry brings its sum: read y)
constexpr uint32_t CD_BC = 0x80000000u;      // Dirichlet dof
constexpr uint32_t CD_SKIP = 0x40000000u;    // the next patch of the chain carries this dof on: no store
constexpr uint32_t CD_BCFIRST = 0x20000000u; // Dirichlet dof whose row y = x this entry writes (exactly one per dof)
constexpr uint32_t CD_MASK = 0x1fffffffu;
constexpr uint32_t CC_ACC = 0x80000000u;
constexpr uint32_t CC_NONE = 0xffffu;
struct ChainPlan
{
  bool ok = false; // false: the level keeps its coloured patch launches (no tensor grid of patches, too few chains ...)
  std::vector<uint32_t> cdofs, ccar;
  std::vector<int32_t> chain_off, chain_patch; // patches of chain i: chain_patch[chain_off[i] .. chain_off[i + 1])
  std::vector<int32_t> launch_first, launch_count; // chains of each colour, in stream order
};
// min_chains: a colour with fewer chains than this cannot fill the GPU with one workgroup per chain -> not ok
int build_chain_plan(ChainPlan& cp, const PatchPlan& plan, int32_t ndofs, const int8_t* bc, const float* centroid,
                     int min_chains);

// Build the plan.  `centroid` [ncells*3] drives the grouping (tensor-grid blocks
// when the centroids form a tensor grid, Morton-ordered chunks otherwise);
// correctness does not depend on the grouping, only the amount of sharing does.
int build_patch_plan(PatchPlan& plan, int P, int32_t ncells, const int32_t* dofmap,
                     const int8_t* bc, int32_t ndofs, const float* centroid,
                     const int32_t* lcells, int32_t n_l, const int32_t* bcells, int32_t n_b);
} // namespace pmg
