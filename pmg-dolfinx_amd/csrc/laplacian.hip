// Matrix-free GLL-collocated Laplacian for gfx950: geometry tensor, the
// sum-factorised stiffness kernel, matrix-free diagonal, collocated load vector.
// Replaces src/laplacian.hpp (geometry_computation :22-113, stiffness_operator
// :143-278, MatFreeLaplacian :284-526) of the reference.
//
// Work decomposition: cells are grouped into coloured patches (patches.hpp: 2x2x8
// cells at P = 4); one workgroup of up to 8 wavefronts applies the operator to one
// patch.  A wavefront takes whole cells (2 at P = 4; at P = 8 two wavefronts share
// one), a lane owns the column of nd points above (a, b), keeps it in registers and
// marches through the nd layers (stiffness_column_kernel below).
//
// HBM layout (owned by the handle):
//   G      [slot][layer c][3][nd*nd] double2 : (G00,G01) (G02,G11) (G12,G22) per
//          quadrature point, slot = patch * K + position in the patch, so one layer
//          of one cell is 3 contiguous runs of nd*nd double2 (the reference stores
//          [cell][q][6] AoS, 48-byte stride per lane, src/laplacian.hpp:221-227);
//          P = 2 uses the line-aligned flat variant described at gflat() below.
//   pdofs  [poff[p] .. poff[p+1]) uint32 : sorted dofs of patch p, Dirichlet and
//          "already written" flags in the top bits -- replaces the per-thread
//          dofmap load + dependent 1-byte bc_marker gather (:182-189) and the
//          zero-fill of y (:466).
//   lmaps  [table][K*N] uint16 : position of (cell slot, layer-major local dof) in the
//          patch list; identical patches share one table (a structured box has one
//          table for all interior patches, so it stays in L2).
//   D      [nd][nd] double : 1-D derivative table (LDS -> registers / SGPRs).
//
// LDS per workgroup: patch x values and patch y accumulators (2 * max_m doubles, 43 KB
// at P = 4) plus three nd x nd slices per cell in flight.  Cell contributions are summed
// with LDS FP64 atomics (ds_add_f64), the global write is a plain store /
// read-modify-write (coloured launches) or one global atomic per patch dof (merged
// launches of small levels and of the boundary shell).
//
// Roofline: HBM-bound, AI 0.85 (P=1) .. 2.05 (P=8) flop/B; algorithmic bytes per
// cell 48N + 4N + 8 + 17U (model "storedG").
#include "common.hpp"
#include "patches.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

using namespace pmg;

struct pmg_laplacian_s
{
  pmg_layout layout = nullptr;
  int P = 0, nd = 0, N = 0, K = 0;
  int32_t ncells = 0, npoints = 0;
  // caller-owned
  const double* kappa = nullptr;
  const int32_t* dofmap = nullptr;
  const double* xgeom = nullptr;
  const int32_t* geom_dofmap = nullptr;
  const int8_t* bc = nullptr;
  // cell-local node order of the caller's arrays (pmg_amd.h): the kernels index nodes by ascending coordinate, so a
  // caller in another order gets an ascending copy of its dofmap (op->dofmap then points to it) and its
  // quadrature-indexed arrays are permuted on the way in (tables) and out (get_geometry)
  int node_order = PMG_NODES_ASCENDING;
  int32_t* dofmap_own = nullptr; // [ncells * N], ascending order; nullptr = the caller's array is used as it is
  int32_t* qperm = nullptr;      // [N] device: caller's cell-local number -> ascending; nullptr = identity
  // owned
  double2* G = nullptr;        // [nslots][3][N]
  double* Gaff = nullptr;      // [nslots][6] constant tensor K K^T / detJ of each (affine) cell
  double* W1 = nullptr;        // [nd] 1-D GLL weights
  bool all_affine = false;     // every listed cell is a parallelepiped
  int geometry_mode = 0;       // 0 = stored G (reference data structure), 1 = affine cells
  double* D = nullptr;         // [nd*nd]
  double* dphi_geom = nullptr; // [3][N][8]
  double* gweights = nullptr;  // [N]
  int32_t* pcell = nullptr;    // [npatch*K]
  int32_t* pncell = nullptr;   // [npatch]
  int32_t* bzero = nullptr;    // dofs first written by the (atomic) boundary launch
  int32_t n_bzero = 0;
  int32_t* poff = nullptr;     // [npatch+1]
  uint32_t* pdofs = nullptr;
  int32_t* lmap_id = nullptr;  // [npatch]
  uint16_t* lmaps = nullptr;   // [nuniq][K*N]
  int32_t npatch = 0;
  std::vector<int32_t> launch_first, launch_count;
  // two halves of the interior on two streams (PatchPlan::launch_stream): the second stream and its fork / order /
  // join events
  std::vector<int8_t> launch_stream;
  int launch_signal = -1, launch_wait = -1;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_order = nullptr, ev_join = nullptr;
  std::vector<int32_t> pcell_h, pncell_h; // host copies for components that share the patches
  long long npdofs = 0;
  int max_m = 0;
  int n_launch_l = 0;
  int n_plain = 0;
  bool needs_zero = false; // some local dof belongs to no listed cell
  // chain form of the interior launches (patches.hpp ChainPlan, stiffness_chain_kernel): available / in use
  bool chain_ok = false, chain_on = false;
  uint32_t* cdofs = nullptr;
  uint32_t* ccar = nullptr;
  int32_t* chain_off = nullptr;
  int32_t* chain_patch = nullptr;
  std::vector<int32_t> chain_first, chain_count; // chains of each colour
  bool stream_policy = true; // the stored tensor exceeds the Infinity Cache: nt loads / stores (launch_stiffness)
  double* diag_inv = nullptr; // [size_local + num_ghosts]
  bool have_diag = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  long long launches = 0; // full operator applications' kernel launches since creation
  long long applies = 0;
  // geometry batching (src/laplacian.hpp:383-396): > 0 = G is not resident; it is recomputed for
  // `batch_patches` patches at a time into a buffer of that size, in every application
  int32_t batch_patches = 0;
  // in-situ timing of the stiffness launches (pmg_laplacian_set_profiling)
  bool profiling = false;
  std::vector<hipEvent_t> prof_events; // pairs: before / after a run of launches
  size_t prof_used = 0;
  long long prof_launches = 0;
};

namespace
{
template <int P>
struct Shape
{
  static constexpr int ND = P + 1;
  static constexpr int N = ND * ND * ND;
  static constexpr PatchShape PS = patch_shape(P);
  static constexpr int K = PS.bx * PS.by * PS.bz; // cells per patch
  static constexpr int MAXM = PS.max_m;           // patch dofs held in LDS
  // An item is CW whole cells worked on by WPC waves, a lane one (a, b) column of one of the cells.  Where nd^2 divides
  // 64 badly a wave per cell group idles many lanes (P = 5: 36 of 64, P = 8: 81 of 128); there four waves share an
  // item of 7 (P = 5: 252 of 256 lanes) or 3 (P = 8: 243 of 256) cells and exchange their slices through workgroup
  // barriers instead of wave-private fences -- worth it only with ONE item per workgroup, i.e. patches of exactly CW
  // cells (round 3, profiles/kernel_tuning_r03.md section 12: P = 5 532 -> 466 us, P = 8 495 -> 421; P = 4 and P = 6,
  // 78 % of the lanes busy, lose or gain nothing this way).
  static constexpr int NQ2 = ND * ND;
  static constexpr bool SHARED_ITEM = P == 5 || P == 8;
#ifdef PMG_ITEM_P // experiment: -DPMG_ITEM_P=<degree> -DPMG_ITEM_CW=<cells> -DPMG_ITEM_WPC=<waves>
  static constexpr int CW = P == PMG_ITEM_P ? PMG_ITEM_CW : (NQ2 <= 64 ? 64 / NQ2 : 1);
  static constexpr int WPC = P == PMG_ITEM_P ? PMG_ITEM_WPC : (NQ2 + 63) / 64;
#else
  static constexpr int CW = P == 5 ? 7 : P == 8 ? 3 : (NQ2 <= 64 ? 64 / NQ2 : 1);
  static constexpr int WPC = SHARED_ITEM ? 4 : (NQ2 + 63) / 64; // waves that share one item
#endif
  static_assert(CW * NQ2 <= 64 * WPC, "an item's columns need a lane each");
  static constexpr int ITEMS = (K + CW - 1) / CW;  // wave-items per full patch
  // measured (profiles/kernel_roofline_r01.md, profiles/kernel_tuning_r02.md): 4 waves and more
  // workgroups per CU for the register-heavy degrees and P = 3, 8 waves otherwise
#ifdef PMG_NWMAX_P
  static constexpr int NWMAX = P == PMG_NWMAX_P ? PMG_NWMAX_V : ((P == 3 || P == 5 || P == 6 || P == 8) ? 4 : 8);
#else
  static constexpr int NWMAX = (P == 3 || P == 5 || P == 6 || P == 8) ? 4 : 8;
#endif
  static constexpr int NG = ITEMS < NWMAX / WPC ? ITEMS : NWMAX / WPC; // items in flight per workgroup
  static constexpr int NW = NG * WPC;                                  // waves per workgroup
  static constexpr int WTHREADS = NW * 64;
  static constexpr int WITER = (MAXM + WTHREADS - 1) / WTHREADS;
  static_assert(MAXM <= 65535, "patch positions are 16-bit");
};

constexpr int Shape_wpc(int P)
{
  switch (P)
  {
  case 1: return Shape<1>::WPC;
  case 2: return Shape<2>::WPC;
  case 3: return Shape<3>::WPC;
  case 4: return Shape<4>::WPC;
  case 5: return Shape<5>::WPC;
  case 6: return Shape<6>::WPC;
  case 7: return Shape<7>::WPC;
  default: return Shape<8>::WPC;
  }
}

// ---- geometry: J, adj(J), det at one quadrature point (src/laplacian.hpp:72-97) ----
__device__ inline void jacobian(const double* __restrict__ xgeom,
                                const int32_t* __restrict__ gdofs, const double* __restrict__ dphi,
                                int nq, int q, double K[3][3], double& detJ)
{
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int k = 0; k < 8; ++k)
  {
    const double* xk = xgeom + 3 * (size_t)gdofs[k];
    double x0 = xk[0], x1 = xk[1], x2 = xk[2];
    double d0 = dphi[(0 * nq + q) * 8 + k], d1 = dphi[(1 * nq + q) * 8 + k],
           d2 = dphi[(2 * nq + q) * 8 + k];
    J[0][0] += x0 * d0;
    J[0][1] += x0 * d1;
    J[0][2] += x0 * d2;
    J[1][0] += x1 * d0;
    J[1][1] += x1 * d1;
    J[1][2] += x1 * d2;
    J[2][0] += x2 * d0;
    J[2][1] += x2 * d1;
    J[2][2] += x2 * d2;
  }
  K[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  K[0][1] = -J[0][1] * J[2][2] + J[0][2] * J[2][1];
  K[0][2] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
  K[1][0] = -J[1][0] * J[2][2] + J[1][2] * J[2][0];
  K[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
  K[1][2] = -J[0][0] * J[1][2] + J[0][2] * J[1][0];
  K[2][0] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  K[2][1] = -J[0][0] * J[2][1] + J[0][1] * J[2][0];
  K[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  // full cofactor expansion along the first row (the reference's :97 drops to
  // the diagonal-J special case)
  detJ = J[0][0] * K[0][0] + J[0][1] * K[1][0] + J[0][2] * K[2][0];
}

// Layout of the stored geometry tensor G (double2 pairs (G00,G01)(G02,G11)(G12,G22)).
//   default: [slot][layer c][pair][a*nd+b]
//   flat (P = 2, gflat()): [patch][item][layer c][pair][cell of the item][a*nd+b],
//     every (item, layer) block padded to whole 128-byte lines -- a wavefront then reads its item's
//     layer as NJ full-width loads of 64 consecutive double2 (whole lines, none shared between two
//     load instructions) and hands the values to the lanes that use them through LDS.  Pays where
//     the per-cell planes are short and misaligned: P = 2 (seven 144-byte pieces per load
//     instruction otherwise), 765 -> 674 us at 128^3; slower at P = 1, 3, 4 (1485 -> 1524, 513 -> 549,
//     458 -> 480 us), so only P = 2 uses it.
//   ring (P >= RING_FROM, gring()): [patch][item group g][item j of the group][layer c][pair][cell of the item][a*nd+b],
//     dense -- the (item, layer) blocks of one item group (a wavefront, or the two wavefronts that share a cell at
//     P = 8) follow each other in the order the group consumes them, across its items.  The group streams them
//     through a ring of LDS slots with LDS-direct loads (stiffness_ring_kernel below).
__host__ __device__ constexpr bool gflat(int nd) { return nd == 3; } // P = 2 only
//   dense (gdense(), round 4): [patch][item][layer c][pair][cell of the item][a*nd+b] with NO padding and DIRECT per-lane
//     loads -- lane (cell of the item, a, b) of the item's waves reads element `lane` of the (item, layer, pair) run, so a
//     load instruction covers one contiguous run of 16 * CW * nd^2 bytes (800 B at P = 4) instead of CW runs of
//     16 * nd^2 bytes that lie 48 N bytes apart (two misaligned 400-byte runs at P = 4); no LDS hand-over.
#ifndef PMG_GDENSE_MASK // bit P set = dense layout at degree P
#define PMG_GDENSE_MASK 0
#endif
__host__ __device__ constexpr bool gdense(int nd) { return !gflat(nd) && ((PMG_GDENSE_MASK >> (nd - 1)) & 1); }
__host__ __device__ constexpr int gcw(int nd) // cells of an item (Shape<P>::CW)
{
  return nd == 6 ? 7 : nd == 9 ? 3 : nd * nd <= 64 ? 64 / (nd * nd) : 1;
}
__host__ __device__ constexpr int gls(int nd) { return ((3 * gcw(nd) * nd * nd + 7) / 8) * 8; } // layer stride
// ring kernel configuration per degree: item groups per workgroup, ring depth in (item, layer) blocks; depth 0 = the
// degree runs on the column kernel with register-held G.  ALL degrees do: the ring kernel is parity-green (every
// degree, both cell lists, odd meshes) and slower -- P = 5 / 6 / 7 / 8: 596 / 577 / 431 / 649 us against 536 / 511 /
// 412 / 496 us for the column kernel (profiles/kernel_tuning_r03.md: the LDS-direct loads themselves cost the time,
// not the waits; the cell loop is bound by the LDS pipe, which the ring loads further).  Kept as a build option
// (-DPMG_RING_P8='{2,3}' ...) for the record of that measurement.
struct RingCfg
{
  int groups, depth;
};
#ifndef PMG_RING_P4
#define PMG_RING_P4 {0, 0}
#endif
#ifndef PMG_RING_P5
#define PMG_RING_P5 {0, 0} // measured with {4, 3}: slower, see above
#endif
#ifndef PMG_RING_P6
#define PMG_RING_P6 {0, 0} // measured with {4, 3}: slower, see above
#endif
#ifndef PMG_RING_P7
#define PMG_RING_P7 {0, 0} // measured with {8, 2}: slower, see above
#endif
#ifndef PMG_RING_P8
#define PMG_RING_P8 {0, 0} // measured with {2, 3}: slower, see above
#endif
__host__ __device__ constexpr RingCfg ring_cfg(int nd)
{
  switch (nd)
  {
  case 5:
    return PMG_RING_P4;
  case 6:
    return PMG_RING_P5;
  case 7:
    return PMG_RING_P6;
  case 8:
    return PMG_RING_P7;
  case 9:
    return PMG_RING_P8;
  default:
    return {0, 0};
  }
}
__host__ __device__ constexpr bool gring(int nd) { return ring_cfg(nd).depth > 0; }
__host__ __device__ constexpr int ring_ipg(int nd, int K) // items per group
{
  return ((K + gcw(nd) - 1) / gcw(nd) + ring_cfg(nd).groups - 1) / (gring(nd) ? ring_cfg(nd).groups : 1);
}
__host__ __device__ constexpr long long gpatch(int nd, int K)
{
  if (gring(nd))
    return (long long)ring_cfg(nd).groups * ring_ipg(nd, K) * nd * 3 * gcw(nd) * nd * nd;
  if (gdense(nd))
    return (long long)((K + gcw(nd) - 1) / gcw(nd)) * nd * 3 * gcw(nd) * nd * nd;
  return gflat(nd) ? (long long)((K + gcw(nd) - 1) / gcw(nd)) * nd * gls(nd) : (long long)K * 3 * nd * nd * nd;
}
// absolute position of (patch slot, quadrature point q = (a,b,c), component pair)
__device__ __forceinline__ size_t gpos(int nd, int K, long long slot, int q, int pair)
{
  const int nsq = nd * nd, N = nsq * nd;
  const int a = q / nsq, b = (q - a * nsq) / nd, c = q - a * nsq - b * nd;
  if (gring(nd))
  {
    const long long p = slot / K;
    const int sl = (int)(slot - p * K), cw = gcw(nd), it = sl / cw, ci = sl - it * cw;
    const int ng = ring_cfg(nd).groups, g = it % ng, j = it / ng, WL = cw * nsq;
    return (size_t)p * gpatch(nd, K) + ((size_t)(g * ring_ipg(nd, K) + j) * nd + c) * (3 * WL) + pair * WL + ci * nsq
           + a * nd + b;
  }
  if (!gflat(nd) && !gdense(nd))
    return (size_t)slot * 3 * N + (c * 3 + pair) * nsq + a * nd + b;
  const long long p = slot / K;
  const int sl = (int)(slot - p * K), cw = gcw(nd), item = sl / cw, ci = sl - item * cw;
  if (gdense(nd))
    return (size_t)p * gpatch(nd, K) + ((size_t)(item * nd + c) * 3 + pair) * (cw * nsq) + ci * nsq + a * nd + b;
  return (size_t)p * gpatch(nd, K) + (size_t)(item * nd + c) * gls(nd) + pair * (cw * nsq) + ci * nsq + a * nd + b;
}

// G for the patch slots [slot0, slot0 + nslots) and every q, paired layout (absolute positions:
// in batch mode G points `slot0` slots before its buffer)
__global__ void geometry_kernel(long long slot0, long long nslots, int nd, int K,
                                const int32_t* __restrict__ pcell,
                                const double* __restrict__ xgeom,
                                const int32_t* __restrict__ geom_dofmap,
                                const double* __restrict__ dphi, const double* __restrict__ w,
                                double2* __restrict__ G)
{
  const int nq = nd * nd * nd;
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nslots * nq)
    return;
  long long slot = gid / nq;
  int q = (int)(gid - slot * nq);
  slot += slot0;
  int c = pcell[slot];
  double g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0;
  if (c >= 0)
  {
    double K[3][3], detJ;
    jacobian(xgeom, geom_dofmap + (size_t)c * 8, dphi, nq, q, K, detJ);
    double s = w[q] / detJ;
    g0 = (K[0][0] * K[0][0] + K[0][1] * K[0][1] + K[0][2] * K[0][2]) * s; // :99-111
    g1 = (K[1][0] * K[0][0] + K[1][1] * K[0][1] + K[1][2] * K[0][2]) * s;
    g2 = (K[2][0] * K[0][0] + K[2][1] * K[0][1] + K[2][2] * K[0][2]) * s;
    g3 = (K[1][0] * K[1][0] + K[1][1] * K[1][1] + K[1][2] * K[1][2]) * s;
    g4 = (K[2][0] * K[1][0] + K[2][1] * K[1][1] + K[2][2] * K[1][2]) * s;
    g5 = (K[2][0] * K[2][0] + K[2][1] * K[2][1] + K[2][2] * K[2][2]) * s;
  }
  G[gpos(nd, K, slot, q, 0)] = make_double2(g0, g1);
  G[gpos(nd, K, slot, q, 1)] = make_double2(g2, g3);
  G[gpos(nd, K, slot, q, 2)] = make_double2(g4, g5);
}

// Constant geometry tensor of an affine cell: K K^T / detJ at the cell centre
// (q-independent when the cell is a parallelepiped); G_q = w_q * this.
__global__ void affine_geometry_kernel(long long nslots, const int32_t* __restrict__ pcell,
                                       const double* __restrict__ xgeom,
                                       const int32_t* __restrict__ geom_dofmap,
                                       double* __restrict__ Gaff)
{
  long long slot = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nslots)
    return;
  int c = pcell[slot];
  double g[6] = {0, 0, 0, 0, 0, 0};
  if (c >= 0)
  {
    const int32_t* gd = geom_dofmap + (size_t)c * 8;
    const double* x0 = xgeom + 3 * (size_t)gd[0];
    const double* xz = xgeom + 3 * (size_t)gd[1]; // (0,0,1)
    const double* xy = xgeom + 3 * (size_t)gd[2]; // (0,1,0)
    const double* xx = xgeom + 3 * (size_t)gd[4]; // (1,0,0)
    double J[3][3];
    for (int i = 0; i < 3; ++i)
    {
      J[i][0] = xx[i] - x0[i];
      J[i][1] = xy[i] - x0[i];
      J[i][2] = xz[i] - x0[i];
    }
    double K[3][3];
    K[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1];
    K[0][1] = -J[0][1] * J[2][2] + J[0][2] * J[2][1];
    K[0][2] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
    K[1][0] = -J[1][0] * J[2][2] + J[1][2] * J[2][0];
    K[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
    K[1][2] = -J[0][0] * J[1][2] + J[0][2] * J[1][0];
    K[2][0] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
    K[2][1] = -J[0][0] * J[2][1] + J[0][1] * J[2][0];
    K[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const double detJ = J[0][0] * K[0][0] + J[0][1] * K[1][0] + J[0][2] * K[2][0];
    const double s = 1.0 / detJ;
    g[0] = (K[0][0] * K[0][0] + K[0][1] * K[0][1] + K[0][2] * K[0][2]) * s;
    g[1] = (K[1][0] * K[0][0] + K[1][1] * K[0][1] + K[1][2] * K[0][2]) * s;
    g[2] = (K[2][0] * K[0][0] + K[2][1] * K[0][1] + K[2][2] * K[0][2]) * s;
    g[3] = (K[1][0] * K[1][0] + K[1][1] * K[1][1] + K[1][2] * K[1][2]) * s;
    g[4] = (K[2][0] * K[1][0] + K[2][1] * K[1][1] + K[2][2] * K[1][2]) * s;
    g[5] = (K[2][0] * K[2][0] + K[2][1] * K[2][1] + K[2][2] * K[2][2]) * s;
  }
  for (int d = 0; d < 6; ++d)
    Gaff[slot * 6 + d] = g[d];
}

// paired slot layout -> the reference's [cell][q][6]
// (qperm: the caller's quadrature-point number -> ascending; nullptr = the caller's order is ascending)
__global__ void geometry_export_kernel(long long slot0, long long nslots, int nd, int K,
                                       const int32_t* __restrict__ pcell, const int32_t* __restrict__ qperm,
                                       const double2* __restrict__ G, double* __restrict__ out)
{
  const int nq = nd * nd * nd;
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nslots * nq)
    return;
  long long slot = gid / nq;
  const int qc = (int)(gid - slot * nq); // the caller's number of the point
  const int q = qperm ? qperm[qc] : qc;
  slot += slot0;
  int c = pcell[slot];
  if (c < 0)
    return;
  double2 a = G[gpos(nd, K, slot, q, 0)], b = G[gpos(nd, K, slot, q, 1)],
          d = G[gpos(nd, K, slot, q, 2)];
  double* o = out + ((size_t)c * nq + qc) * 6;
  o[0] = a.x;
  o[1] = a.y;
  o[2] = b.x;
  o[3] = b.y;
  o[4] = d.x;
  o[5] = d.y;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains
// the vector-memory counter (s_waitcnt vmcnt(0)), which would stall every wave on
// the G loads issued at the top of the kernel; nothing in this kernel passes data
// between threads through global memory, so LDS ordering is all that is needed
// (cdna_hip_programming.md, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// ---- write-back of a patch's sums, shared by the kernels below --------------------------------------------
// Round 4, read at the ISA level: written as one loop "load the list entry, branch on it, store", the write-back was six
// DEPENDENT round trips per thread -- `global_load_dword; s_waitcnt vmcnt(0); ... global_store` per entry, and since
// the counter retires in order every one of those waits also covered the acknowledgement of the previous entry's
// store.  So: the list entries are re-read in ONE pass of unconditional loads (clamped index), issued by every
// wavefront as it leaves the cell loop, BEFORE the barrier that ends the accumulation; behind the barrier the stores
// go out back to back; the rare Dirichlet rows (y = x, src/laplacian.hpp:273-274: one more load each) come last, in a
// pass of their own that interior patches skip.
template <int ITER, int THREADS>
__device__ __forceinline__ void patch_list_reload(uint32_t (&mk)[ITER], const uint32_t* __restrict__ pd, int M, int t)
{
#pragma unroll
  for (int k = 0; k < ITER; ++k)
  {
    const int i = t + k * THREADS;
    mk[k] = pd[i < M ? i : M - 1];
  }
}
template <int ITER, int THREADS, bool NT>
__device__ __forceinline__ void patch_write_back(const uint32_t (&mk)[ITER], int M, int t, const double* sy,
                                                 const double* __restrict__ x, double* __restrict__ y, int atomic_out)
{
  bool bc_row = false;
#pragma unroll
  for (int k = 0; k < ITER; ++k)
  {
    const int i = t + k * THREADS;
    const uint32_t dof = mk[k] & PD_MASK;
    const bool mine = i < M;
    if (mine && !(mk[k] & PD_BC))
    {
      const double v = sy[i];
      if (atomic_out)
        atomicAdd(&y[dof], v); // merged launch (global_atomic_add_f64)
      else if constexpr (NT)
        __builtin_nontemporal_store(v, &y[dof]);
      else
        y[dof] = v;
    }
    bc_row |= mine && (mk[k] & (PD_BC | PD_ACC)) == PD_BC;
  }
  if (__builtin_amdgcn_ballot_w64(bc_row) != 0) // wave-uniform: interior patches never enter
  {
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const int i = t + k * THREADS;
      if (i < M && (mk[k] & (PD_BC | PD_ACC)) == PD_BC)
        y[mk[k] & PD_MASK] = x[mk[k] & PD_MASK]; // :273-274
    }
  }
}

// ---- diagnostic build only (-DPMG_STAMPS): where a wavefront spends its time -----------------------------------
// Every wavefront keeps up to eight readings of the constant 100 MHz clock (s_memrealtime: 10 ns ticks) in scalar
// registers and lane 0 stores them once, at its end, to a buffer of their own: [workgroup][wavefront][8].  No stamp
// executes in the product build; the timings of a stamped build are read for their SHARES only
// (cdna_hip_programming.md, In-kernel stamps).
#ifdef PMG_STAMPS
__device__ unsigned long long* g_stamp_buffer = nullptr;
__device__ int g_stamp_capacity = 0; // workgroups the buffer has room for
#define PMG_STAMP_DECL unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PMG_STAMP(i)                                                                                                  \
  do                                                                                                                  \
  {                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    stamp_[i] = __builtin_amdgcn_s_memrealtime();                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  } while (0)
#define PMG_STAMP_FLUSH(nwaves)                                                                                       \
  do                                                                                                                  \
  {                                                                                                                   \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* 7: the stores of the write-back are acknowledged */            \
    PMG_STAMP(7);                                                                                                     \
    if ((threadIdx.x & 63) == 0 && g_stamp_buffer && (int)blockIdx.x < g_stamp_capacity)                              \
      for (int i_ = 0; i_ < 8; ++i_)                                                                                  \
        g_stamp_buffer[((size_t)blockIdx.x * (nwaves) + (threadIdx.x >> 6)) * 8 + i_] = stamp_[i_];                   \
  } while (0)
#else
#define PMG_STAMP_DECL
#define PMG_STAMP(i)
#define PMG_STAMP_FLUSH(nwaves)
#endif

// ---- touching the tensor ahead of its use (round 4) ------------------------------------------
// What a wavefront can have in flight of the tensor is what its registers hold: two layers (4.8 KB), sixteen
// wavefronts per compute unit, ~60 KB -- and with all 256 units streaming the loaded latency is such that this
// depth gives 4.6 - 4.9 TB/s (profiles/kernel_tuning_r04.md section 10: the same cell loop takes 5 us per patch with
// 16 units active and 10.7 us with 256).  A TOUCH costs no register: one LDS-direct load of 4 bytes per lane, every
// lane a different 128-byte line, into a scratch word per lane that nobody reads -- 8 KB of lines requested by one
// instruction.  A wavefront touches the 12 KB of its NEXT item when it starts an item (and its first item in the
// gather phase), default cache policy, so that the layer loads proper (nt) find their lines in L2 or in the
// Infinity Cache instead of HBM.  Inline assembly (no destination register, so nothing the compiler could reuse
// early); the counter retires in order, and a touch is no slower than the tensor loads queued behind it.
typedef __attribute__((address_space(3))) char lds_byte_t;
__device__ __forceinline__ void lds_touch(const void* sbase, unsigned voff, unsigned lds_byte)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_byte)
               : "memory");
}
#ifndef PMG_TOUCH_MASK // bit P set = the degree-P column kernel touches ahead
#define PMG_TOUCH_MASK 0
#endif
constexpr bool touch_ahead(int P) { return (PMG_TOUCH_MASK >> P) & 1; }
#ifndef PMG_TOUCH_STRIDE
#define PMG_TOUCH_STRIDE 128
#endif

// ---- the hot kernel, column form --------------------------------------
//
// One workgroup per patch, NW wavefronts.  Phase 0 / write-back as in the block
// kernel.  In between every wavefront works on its own: it takes CW whole cells
// (2 at P = 4), a lane owns the column of nd points above (a, b), keeps the
// column's dofs and results in registers and marches through the nd layers
// (the register-blocked 2-D scheme of libParanumal / hipBone).  Per layer the x
// and y contractions exchange one nd x nd slice through a wave-private LDS
// region -- LDS executes a wave's instructions in order, so no s_barrier and no
// waitcnt is needed inside the cell loop, only a compiler fence; the z
// contraction stays in registers with wave-uniform table entries (scalar loads).
// A cell costs 4*nd LDS reads per point instead of 12*nd, the 1-D tables for the
// lane's a and b sit in registers, and the layer-(k+1) slice of G is in flight
// while layer k is computed.
__device__ __forceinline__ void wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cache policy.  The G stream is read exactly once per application and is ~6x larger
// than the MALL: it is loaded non-temporally (nt bit) so that it does not displace x, y,
// the dof lists and the shared tables from L2 / MALL, and the write-back stores of y are
// non-temporal as well.  Measured at P = 4, 64^3: 505 -> 475 us with nt stores, -> 430 us
// with nt G loads on top; the sc0 / sc1 bits make no difference; nt on the x / y gathers
// or on the dof lists is slower.
// At P = 1 (8 quadrature points per cell, every dof shared by 8 cells) the default
// policy is faster (1610 vs 1750 us at 256^3), so the hint starts at P = 2.
constexpr int NT_FROM = 2;
typedef double gvec2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ double2 gload(const double2* p)
{
  if constexpr (NT)
  {
    gvec2 v = __builtin_nontemporal_load(reinterpret_cast<const gvec2*>(p));
    return make_double2(v.x, v.y);
  }
  else
    return *p;
}

// Slice reads.  The compiler pairs neighbouring LDS reads into ds_read2_b64; issued one by one
// (volatile LDS loads are not paired) the kernel is 7 % faster at P = 5 and 4 % at P = 8, unchanged at
// P <= 4 and slower at P = 6, 7 (profiles/kernel_tuning_r02.md).
typedef __attribute__((address_space(3))) volatile double lds_vdouble;
template <bool UNPAIRED>
__device__ __forceinline__ double slice_load(double& v)
{
  if constexpr (UNPAIRED)
    return *(lds_vdouble*)&v;
  else
    return v;
}
#ifdef PMG_UNPAIRED_MASK // experiment: bit P set = unpaired slice reads at degree P
constexpr bool unpaired_slice_reads(int P) { return (PMG_UNPAIRED_MASK >> P) & 1; }
#else
constexpr bool unpaired_slice_reads(int P) { return P == 5 || P == 8; }
#endif

// degrees whose kernel keeps the patch's dof list in LDS for the write-back (4 bytes per patch dof).  Round 4, two
// rounds on one box against the list re-read from global memory: P = 5 454 against 460 - 470 us, P = 7 368 - 372 against
// 376, P = 8 410 - 411 against 420; P = 4 +0.5 % (within noise, left out); P = 6 474 - 482 against 441 - 445 (the 9.6 KB
// cost it a workgroup per CU).
#ifndef PMG_LIST_LDS_MASK
#define PMG_LIST_LDS_MASK ((1 << 5) | (1 << 7) | (1 << 8))
#endif
constexpr bool list_in_lds(int P) { return (PMG_LIST_LDS_MASK >> P) & 1; }
// degrees that run on the stream form of the kernel (stiffness_stream_kernel below); bit P of the mask
#ifndef PMG_STREAM_MASK
#define PMG_STREAM_MASK 0
#endif
constexpr bool stream_form(int P)
{
  return ((PMG_STREAM_MASK >> P) & 1) && Shape_wpc(P) == 1 && !gflat(P + 1);
}
// ---- the y contraction inside the wavefront (DPP row shifts) instead of through an LDS slice (round 4) ----
// tools/dpp_probe.hip: a compute unit sustains 26.7 unit clocks per contraction of a wavefront with both directions
// through LDS (the pipe the four SIMDs share), 25.8 with both as DPP shifts (the SIMDs' own ALUs), 18.7 with one each.
// nd = 5 only: the five lanes (a, b = 0 .. 4) of a column group sit side by side, three groups to a 16-lane row; a
// value moves as two v_mov_b32 dpp, lanes outside the group contribute through a zero coefficient (nine coefficient
// registers per direction instead of five).
// The transposed table without its registers.  For Lagrange polynomials on ANY distinct nodes D_ij = (l_j / l_i) /
// (x_i - x_j), i != j, with the barycentric weights l, hence D_ji = -(l_i / l_j)^2 D_ij and
//     (D^T f)_i = 2 D_ii f_i - rho_i sum_j D_ij (f_j / rho_j),       rho_i = (l_i / l_0)^2 = -D_0i / D_i0,
// i.e. the backward contraction is the FORWARD one applied to the scaled fluxes: the five registers of D[.][a] (and
// the nine shift coefficients a DPP direction would need for its transpose) become rho, 1 / rho and 2 D_ii.  Same
// value to rounding (tests: 1e-12 against the oracle), not bit for bit.
// PMG_DPP_MODE: 0 off; 1 the forward y derivative with DPP (transposed tables kept: spills, for the record); 2 both y
// contractions with DPP, both transposes by the identity; 4 the identity alone (both directions through LDS).
#ifndef PMG_DPP_MODE
#define PMG_DPP_MODE 0
#endif
#ifndef PMG_DPP_MASK // bit P set = mode 2 at degree P (degrees whose nd <= 8 columns groups fit a 16-lane row)
#define PMG_DPP_MASK 0
#endif
constexpr int dpp_mode(int P) { return P == 4 && PMG_DPP_MODE != 0 ? PMG_DPP_MODE : (((PMG_DPP_MASK) >> P) & 1) ? 2 : 0; }
#ifndef PMG_IDT_MASK // bit P set = the degree-P column kernel replaces its transposed tables by the identity
#define PMG_IDT_MASK 0x20
#endif
// P = 5 is the one degree where the identity pays: 129 registers instead of 164, and asked for four wavefronts per SIMD
// the allocator finds 128 without a spill -- four workgroups per unit instead of three (33 KB of LDS each): 432 - 444
// against 452 - 459 us at 51^3, 223 - 225 against 230 - 232 us at 40^3 (profiles/kernel_tuning_r04.md section 12).
#ifndef PMG_P5_FOUR_WAVES
#define PMG_P5_FOUR_WAVES 1
#endif
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true); // bound_ctrl: lanes outside the row read zero
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum_s C[s + nd - 1] * v(lane + s), s = -(nd - 1) .. nd - 1, inside the 16-lane row
template <int ND>
__device__ __forceinline__ double dpp_contract(double v, const double (&C)[2 * ND - 1])
{
  static_assert(ND >= 2 && ND <= 8, "groups of nd lanes inside a 16-lane row");
  double acc = C[ND - 1] * v;
#define PMG_DPP_SHIFT(S)                                                                                              \
  if constexpr (S < ND)                                                                                               \
  {                                                                                                                   \
    acc += C[ND - 1 - S] * dpp_move<0x110 + S>(v); /* row_shr:S: the value of lane - S */                             \
    acc += C[ND - 1 + S] * dpp_move<0x100 + S>(v); /* row_shl:S: the value of lane + S */                             \
  }
  PMG_DPP_SHIFT(1)
  PMG_DPP_SHIFT(2)
  PMG_DPP_SHIFT(3)
  PMG_DPP_SHIFT(4)
  PMG_DPP_SHIFT(5)
  PMG_DPP_SHIFT(6)
  PMG_DPP_SHIFT(7)
#undef PMG_DPP_SHIFT
  return acc;
}

// Issue priority by progress inside an item (experiment, -DPMG_PRIO_BALANCE=1 | 2; profiles/kernel_tuning_r04.md
// section 11): the wavefronts of a SIMD are served in age order (section 9), so the older half of a workgroup reaches
// the closing barrier first and idles there with nothing in flight.  =1: a wavefront early in its item ranks higher
// (layers 3 3 2 2 1 ...), =2: one about to finish ranks higher; gather and write-back at 3 either way.
#ifndef PMG_PRIO_BALANCE
#define PMG_PRIO_BALANCE 0
#endif
__device__ __forceinline__ void prio_level(int v) // v is a constant after unrolling: one s_setprio remains
{
  if constexpr (PMG_PRIO_BALANCE != 0)
  {
    switch (v)
    {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
  }
}
__device__ __forceinline__ void layer_prio(int k, int nd)
{
  const int early = 3 - (2 * k + 1) / nd - (k + 1 == nd ? 1 : 0); // nd = 5: 3 3 2 2 1
  prio_level(PMG_PRIO_BALANCE == 2 ? 4 - early : early);          //          1 1 2 2 3
}
// minimum waves per SIMD the register allocation has to leave room for: two workgroups per CU up to
// P = 4; the register-heavy degrees take what they need (profiles/kernel_resources_r02.md)
template <int P>
constexpr int min_waves_per_simd()
{
  // (P = 5 with the identity: the allocator lands on 129 registers; asked for four wavefronts per SIMD it has to find 128)
  return P <= 4 ? (2 * Shape<P>::NW + 3) / 4 : (P == 5 && (((PMG_IDT_MASK) >> 5) & 1) && PMG_P5_FOUR_WAVES) ? 4 : 1;
}
template <int P, bool AFF, bool NT>
__global__ void __launch_bounds__(Shape<P>::WTHREADS, min_waves_per_simd<P>())
    stiffness_column_kernel(const double* __restrict__ x, double* __restrict__ y,
                            const double2* __restrict__ G, const double* __restrict__ Gaff,
                            const double* __restrict__ W1, const int32_t* __restrict__ poff,
                            const uint32_t* __restrict__ pdofs,
                            const int32_t* __restrict__ lmap_id,
                            const uint16_t* __restrict__ lmaps, const int32_t* __restrict__ pcell,
                            const int32_t* __restrict__ pncell, const double* __restrict__ kappa,
                            const double* __restrict__ Dg, int first, int atomic_out)
{
  using Sh = Shape<P>;
  constexpr int ND = Sh::ND, N = Sh::N, K = Sh::K, NQ2 = Sh::NQ2, CW = Sh::CW, NG = Sh::NG, WPC = Sh::WPC;
  constexpr int MAXM = Sh::MAXM, THREADS = Sh::WTHREADS, ITER = Sh::WITER;
  constexpr int WL = CW * NQ2; // columns of one item (a wave, or WPC waves sharing a cell)
  // NT: streaming cache policy for G and the y write-back (chosen per operator, launch_stiffness)
  constexpr bool UNPAIRED = unpaired_slice_reads(P);
  __shared__ double sD[ND * ND];
  __shared__ double skap[K];
  __shared__ double sx[MAXM];
  __shared__ double sy[MAXM];
  // The patch's dof list, kept for the write-back (round 4): under load a dependent global load costs ~2 us even when
  // it hits in L2 (in-kernel stamps, profiles/kernel_tuning_r04.md), and re-reading the list was one such round trip
  // per workgroup behind its closing barrier.
  constexpr bool LIST_IN_LDS = list_in_lds(P);
  __shared__ uint32_t sm[LIST_IN_LDS ? MAXM : 1];
  __shared__ double sq[NG * WL];
  __shared__ double sgr[NG * WL + 1]; // (+ 1: the spare element idle lanes of the DPP layout write)
  __shared__ double sgs[NG * WL];
  // flat G layout: one layer of the item, as loaded (NJ x 64 double2), for the hand-over to the lanes
  constexpr bool FLAT = !AFF && WPC == 1 && gflat(ND);
  constexpr int FL = 3 * WL, NJ = (FL + 63) / 64, LS = gls(ND);
  __shared__ double2 sgb[FLAT ? NG * NJ * 64 : 1];

  PMG_STAMP_DECL;
  PMG_STAMP(0); // entry
  prio_level(3);
  const int p = first + blockIdx.x;
  const int t = threadIdx.x;
  const int off = poff[p];
  const int M = poff[p + 1] - off; // 1 <= M <= MAXM
  const int table = lmap_id[p];
  const int nc = pncell[p];
  // touch-ahead (see lds_touch): the item's tensor block is CW cells of 48 N bytes, contiguous in the default layout
  constexpr bool TOUCH = !AFF && touch_ahead(P) && WPC == 1 && !gflat(ND) && !gdense(ND);
  __shared__ uint32_t stouch[TOUCH ? Sh::NW * 64 : 1];
  auto touch_item = [&](int it) {
    if constexpr (TOUCH)
    {
      constexpr unsigned STRIDE = PMG_TOUCH_STRIDE; // bytes between two touched words
      constexpr unsigned BYTES = CW * 3 * N * 16, LINES = (BYTES + STRIDE - 1) / STRIDE + 1;
      const int w = __builtin_amdgcn_readfirstlane(t >> 6);
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_byte_t*)&stouch[w * 64]);
      const char* base = reinterpret_cast<const char*>(G + ((size_t)p * K + (size_t)(it * CW < K ? it * CW : K - 1)) * 3 * N);
      unsigned ln = (unsigned)(t & 63);
      asm volatile("" : "+v"(ln)); // (opaque: nothing of the offsets is computed ahead and held through the cell loop)
#pragma unroll
      for (unsigned j = 0; j < (LINES + 63) / 64; ++j)
      {
        const unsigned o = (ln + 64 * j) * STRIDE;
        lds_touch(base, o < BYTES ? o : BYTES - 4, dst);
      }
    }
  };

  // ---- phase 0: gather (unconditional loads, clamped indices: counted vmcnt waits)
  {
    uint32_t m[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const int i = t + k * THREADS;
      m[k] = pdofs[off + (i < M ? i : M - 1)];
    }
    const int cellk = pcell[(size_t)p * K + (t < K ? t : K - 1)];
    const double dval = Dg[t < ND * ND ? t : ND * ND - 1];
    touch_item(__builtin_amdgcn_readfirstlane((t >> 6) / WPC)); // the wavefront's first item, under the gather
    double xv[ITER], yv[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const uint32_t dof = m[k] & PD_MASK;
      const bool acc = !atomic_out && (m[k] & (PD_ACC | PD_BC)) == PD_ACC;
#ifdef PMG_ABL_NOGATHER
      xv[k] = 1.0 + dof;
      yv[k] = acc ? 1.0 : 0.0;
#else
      xv[k] = x[dof];
      const double* ya = acc ? (const double*)(y + dof) : (x + dof);
      yv[k] = *ya;
#endif
    }
    const double kapk = kappa[cellk >= 0 ? cellk : 0];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const int i = t + k * THREADS;
      if (i < M)
      {
        const bool acc = !atomic_out && (m[k] & (PD_ACC | PD_BC)) == PD_ACC;
        sx[i] = (m[k] & PD_BC) ? 0.0 : xv[k]; // src/laplacian.hpp:186-189
        sy[i] = acc ? yv[k] : 0.0;
        if constexpr (LIST_IN_LDS)
          sm[i] = m[k];
      }
    }
    if (t < ND * ND)
      sD[t] = dval;
    for (int i = t; i < K; i += THREADS)
      skap[i] = (i == t) ? kapk : kappa[pcell[(size_t)p * K + i] >= 0 ? pcell[(size_t)p * K + i] : 0];
  }
  PMG_STAMP(1); // gathered values written to LDS
  lds_barrier();
  PMG_STAMP(2); // behind the gather's barrier

  // ---- cell loop: each wave on its own
  // (nd^2 > 64, i.e. P = 8: WPC waves share a cell, the slices are exchanged between
  // them, so the fences inside the layer loop become workgroup barriers and every
  // wave runs the same number of items)
  const int wave = (t >> 6) / WPC, lane = (t & 63) + 64 * ((t >> 6) % WPC);
  constexpr int DPPM = (ND >= 4 && ND <= 8 && WPC == 1 && !FLAT) ? dpp_mode(P) : 0;
  constexpr int GPR = 16 / ND; // column groups per 16-lane row
  constexpr int DPPY = DPPM == 4 ? 0 : DPPM;          // 0 / 1 / 2: how much of the y direction goes through DPP
  constexpr bool IDT = DPPM == 2 || DPPM == 4 || (((PMG_IDT_MASK) >> P) & 1); // transposes by the identity
  // DPP layout: lane = 16 row + nd g + b; column group GPR row + g = (cell of the item) nd + a; the lanes of a row past
  // its groups and the groups past the item's columns idle on a copy of the last column (finite values under zero
  // coefficients)
  const int grp_ = GPR * ((t & 63) >> 4) + ((t & 15) / ND);
  const bool lane_ok = DPPY ? ((t & 15) < GPR * ND && grp_ < CW * ND) : lane < WL;
  const int lw = DPPY ? (lane_ok ? grp_ * ND + ((t & 15) - ND * ((t & 15) / ND)) : WL - 1) : (lane_ok ? lane : WL - 1);
  const int cw = lw / NQ2;          // cell of this lane inside the wave item
  const int ab = lw - cw * NQ2;     // column: a = x index, b = y index
  const int a = ab / ND, b = ab - a * ND;
  // The lane's rows / columns of the 1-D table, in registers (4 nd doubles; re-reading them from LDS
  // in every layer frees the registers for one more wave per SIMD but is 15 % slower at every degree,
  // profiles/kernel_tuning_r02.md)
  double Da[ND], Db[ND], DTa[ND], DTb[ND]; // D[a][.], D[b][.], D[.][a], D[.][b]
#pragma unroll
  for (int mm = 0; mm < ND; ++mm)
  {
    Da[mm] = sD[a * ND + mm];
    Db[mm] = sD[b * ND + mm];
    DTa[mm] = sD[mm * ND + a];
    DTb[mm] = sD[mm * ND + b];
  }
  // shift coefficients of the DPP direction: Cb[s + nd - 1] = D[b][b + s] inside the group, else 0
  double Cb[DPPY >= 1 ? 2 * ND - 1 : 1];
  if constexpr (DPPY >= 1)
  {
#pragma unroll
    for (int sft = -(ND - 1); sft <= ND - 1; ++sft)
    {
      const bool in = lane_ok && b + sft >= 0 && b + sft < ND;
      Cb[sft + ND - 1] = in ? sD[b * ND + (in ? b + sft : b)] : 0.0;
    }
  }
  // the identity's constants (see dpp_mode): rho, 1 / rho, 2 D_ii for the lane's a and b
  double rho_a = 1.0, irho_a = 1.0, d2a = 0.0, rho_b = 1.0, irho_b = 1.0, d2b = 0.0;
  if constexpr (IDT)
  {
    rho_a = a == 0 ? 1.0 : -sD[a] / sD[a * ND];
    rho_b = b == 0 ? 1.0 : -sD[b] / sD[b * ND];
    irho_a = 1.0 / rho_a;
    irho_b = 1.0 / rho_b;
    d2a = 2.0 * sD[a * ND + a];
    d2b = 2.0 * sD[b * ND + b];
  }
  const double wab = AFF ? W1[a] * W1[b] : 0.0; // 1-D GLL weights of the lane's column
  double* q_s = sq + wave * WL + cw * NQ2;  // this cell's slices
  double* gr_s = sgr + wave * WL + cw * NQ2;
  double* gs_s = sgs + wave * WL + cw * NQ2;
  double* gr_w = (DPPY != 0 && !lane_ok) ? sgr + NG * WL : gr_s + ab; // where the lane writes its x flux
  const int items = WPC > 1 ? (((nc + CW - 1) / CW + NG - 1) / NG) * NG : (nc + CW - 1) / CW;
  auto slice_sync = [] {
    if constexpr (WPC > 1)
      lds_barrier();
    else
      wave_fence();
  };

  for (int it = wave; it < items; it += NG)
  {
#ifdef PMG_STAMPS
    if (it >= wave + NG)
      PMG_STAMP(3); // first item done (overwritten by later items: the start of the LAST item)
#endif
    const int slot = it * CW + cw;
    const int slotc = slot < K ? slot : K - 1;
    if (it + NG < items)
      touch_item(__builtin_amdgcn_readfirstlane(it + NG)); // the wavefront's next item
    // (scalar bases + 32-bit lane offsets: a 64-bit per-lane pointer costs two registers, and a spilled one is reloaded
    // behind a wait that drains the memory counter)
    const uint16_t* lmb = lmaps + (size_t)table * (K * N);
    const unsigned lmo = (unsigned)(slotc * N + ab);
    // dense layout: the item's (layer, pair) runs of WL elements, element = the lane's (cell, column) index
    constexpr bool DENSE = !AFF && gdense(ND);
    constexpr int GPS = DENSE ? WL : NQ2; // stride between the three pairs of a layer
    const double2* Gb = G + (size_t)p * (DENSE ? gpatch(ND, K) : (long long)K * 3 * N);
    const unsigned Gs = DENSE ? (unsigned)((it < Sh::ITEMS ? it : Sh::ITEMS - 1) * (ND * 3 * WL) + lw)
                              : (unsigned)(slotc * 3 * N + ab);
    int l[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k)
      l[k] = lmb[lmo + (unsigned)(k * NQ2)];
    // storedG: G layers 0 .. GD-1 in flight (empty slots hold zeros).
    // affine cells (AFF): G_q = w_a w_b w_c * Gc with one constant tensor Gc per cell.
#ifndef PMG_GD
#define PMG_GD 1
#endif
    constexpr int GD = P == 4 ? PMG_GD : 1; // G layers in flight per wave (deeper costs registers, i.e. resident waves: no gain)
    double2 gq[AFF ? 1 : GD][3];
    double2 gfl[FLAT ? NJ : 1]; // flat layout: the next layer as loaded
    // (only used when FLAT; the item index is wave-uniform: a scalar base plus 32-bit lane offsets)
    const double2* Gi = G + (size_t)p * gpatch(ND, K) + (size_t)__builtin_amdgcn_readfirstlane(it) * ND * LS;
    int eo[FLAT ? NJ : 1]; // the lane's elements of a layer (clamped: the tail lanes re-read the last one)
    if constexpr (FLAT)
    {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        eo[jj] = lane + 64 * jj < FL ? lane + 64 * jj : FL - 1;
    }
    double gc[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (AFF)
    {
      const double* ga = Gaff + ((size_t)p * K + slotc) * 6;
#pragma unroll
      for (int d = 0; d < 6; ++d)
        gc[d] = ga[d];
    }
    else if constexpr (FLAT)
    {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        gfl[jj] = gload<NT>(Gi + eo[jj]);
    }
    else
    {
#pragma unroll
      for (int d = 0; d < GD; ++d)
      {
        gq[d][0] = gload<NT>(Gb + (Gs + (unsigned)(d * 3 * GPS)));
        gq[d][1] = gload<NT>(Gb + (Gs + (unsigned)(d * 3 * GPS + GPS)));
        gq[d][2] = gload<NT>(Gb + (Gs + (unsigned)(d * 3 * GPS + 2 * GPS)));
      }
    }
    // (identity modes: kappa multiplies the cell's INPUT once -- the operator is linear in it, same value to rounding --
    // and the positions are held two to a register through the layer loop: the registers the DPP form is short of)
    const double kap = IDT ? 1.0 : skap[slotc];
    double u[ND], Aq[ND];
    {
      const double kin = IDT ? skap[slotc] : 1.0;
#pragma unroll
      for (int k = 0; k < ND; ++k)
      {
        u[k] = IDT ? kin * sx[l[k]] : sx[l[k]];
        Aq[k] = 0.0;
      }
    }
    unsigned lp[IDT ? (ND + 1) / 2 : 1];
    if constexpr (IDT)
    {
#pragma unroll
      for (int k = 0; k < ND; k += 2)
        lp[k / 2] = (unsigned)l[k] | (k + 1 < ND ? (unsigned)l[k + 1] << 16 : 0u);
#pragma unroll
      for (int j = 0; j < (ND + 1) / 2; ++j)
        asm volatile("" : "+v"(lp[j])); // (opaque: the unpacked values are not kept alongside)
    }
#pragma unroll
    for (int k = 0; k < ND; ++k)
    {
      double2 g01, g23, g45;
      if constexpr (AFF)
      {
        const double sc = wab * W1[k]; // w_a w_b w_c; W1[k] is wave-uniform (scalar load)
        g01 = make_double2(sc * gc[0], sc * gc[1]);
        g23 = make_double2(sc * gc[2], sc * gc[3]);
        g45 = make_double2(sc * gc[4], sc * gc[5]);
      }
      else if constexpr (FLAT)
      {
        double2* gb = sgb + wave * (NJ * 64);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
          gb[lane + 64 * jj] = gfl[jj]; // as loaded ...
        wave_fence();
        g01 = gb[lw]; // ... and as used: [pair][cell of the item][column]
        g23 = gb[WL + lw];
        g45 = gb[2 * WL + lw];
        if (k + 1 < ND)
        {
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj)
            gfl[jj] = gload<NT>(Gi + (k + 1) * LS + eo[jj]);
        }
      }
      else
      {
        g01 = gq[k % GD][0];
        g23 = gq[k % GD][1];
        g45 = gq[k % GD][2];
        if (k + GD < ND) // refill the slot with layer k + GD
        {
          gq[k % GD][0] = gload<NT>(Gb + (Gs + (unsigned)((k + GD) * 3 * GPS)));
          gq[k % GD][1] = gload<NT>(Gb + (Gs + (unsigned)((k + GD) * 3 * GPS + GPS)));
          gq[k % GD][2] = gload<NT>(Gb + (Gs + (unsigned)((k + GD) * 3 * GPS + 2 * GPS)));
        }
      }
      layer_prio(k, ND);
      q_s[ab] = u[k];
      slice_sync();
      double qr = 0.0, qs = 0.0, qt = 0.0;
      if constexpr (DPPY >= 1)
        qs = dpp_contract<ND>(u[k], Cb); // d/dy inside the wavefront
#pragma unroll
      for (int mm = 0; mm < ND; ++mm)
      {
        qr += Da[mm] * slice_load<UNPAIRED>(q_s[mm * ND + b]);  // d/dx: sum over a, :195-199
        if constexpr (DPPY == 0)
          qs += Db[mm] * slice_load<UNPAIRED>(q_s[a * ND + mm]);  // d/dy: sum over b, :206-210
        qt += Dg[k * ND + mm] * u[mm];    // d/dz: registers, uniform table, :214-218
      }
      const double fr = kap * (g01.x * qr + g01.y * qs + g23.x * qt); // :233
      const double fs = kap * (g01.y * qr + g23.y * qs + g45.x * qt); // :234
      const double ft = kap * (g23.x * qr + g45.x * qs + g45.y * qt); // :235
      double acc = 0.0;
      if constexpr (IDT)
      {
        // the transposes as forward contractions of the scaled fluxes (see dpp_mode)
        // (DPP layout: an idle lane is a copy of the last column whose y derivative came from the wrong neighbours --
        // its flux must not reach the slice the real lane writes)
        *gr_w = fr * irho_a; // (no branch here: one costs the layer loop its register allocation)
        if constexpr (DPPY < 2)
          gs_s[ab] = fs * irho_b;
        slice_sync();
        double sx_ = 0.0, sy_ = 0.0;
        if constexpr (DPPY >= 2)
          sy_ = dpp_contract<ND>(fs * irho_b, Cb); // :255-259 inside the wavefront
#pragma unroll
        for (int mm = 0; mm < ND; ++mm)
        {
          sx_ += Da[mm] * slice_load<UNPAIRED>(gr_s[mm * ND + b]); // :246-251
          if constexpr (DPPY < 2)
            sy_ += Db[mm] * slice_load<UNPAIRED>(gs_s[a * ND + mm]); // :255-259
          Aq[mm] += Dg[k * ND + mm] * ft;     // :263-267
        }
        acc = (d2a * fr + d2b * fs) - (rho_a * sx_ + rho_b * sy_);
      }
      else
      {
        gr_s[ab] = fr;
        gs_s[ab] = fs;
        slice_sync();
#pragma unroll
        for (int mm = 0; mm < ND; ++mm)
        {
          acc += DTa[mm] * slice_load<UNPAIRED>(gr_s[mm * ND + b]); // :246-251
          acc += DTb[mm] * slice_load<UNPAIRED>(gs_s[a * ND + mm]); // :255-259
          Aq[mm] += Dg[k * ND + mm] * ft;     // :263-267
        }
      }
      Aq[k] += acc;
      slice_sync();
    }
    // Every lane adds (no branch: a conditional here lets the compiler sink the
    // whole accumulation into it and keep every layer's operands live); lanes
    // without a cell (idle lanes, slots past the patch's cells -- their indices
    // were clamped onto a real slot) add an exact zero.
    const bool contributes = lane_ok && slot < nc;
#pragma unroll
    for (int k = 0; k < ND; ++k)
    {
      const int lk = IDT ? (int)((lp[k / 2] >> (16 * (k & 1))) & 0xffffu) : l[k];
      atomicAdd(&sy[lk], contributes ? Aq[k] : 0.0); // :270,277 -- in LDS (ds_add_f64)
    }
  }
  // ---- write back (plain stores; the accumulator started from the earlier colours' y)
#ifdef PMG_ABL_NOWB
  lds_barrier();
  if (sy[t % MAXM] == 1.2345e-300)
    y[t] = 1.0;
  return;
#endif
#ifdef PMG_ABL_SERIAL_WB // A/B timing only: the write-back as it was up to round 3 (list entry re-read inside the branch)
  lds_barrier();
#pragma unroll
  for (int k = 0; k < ITER; ++k)
  {
    const int i = t + k * THREADS;
    if (i < M)
    {
      const uint32_t mk = pdofs[off + i];
      const uint32_t dof = mk & PD_MASK;
      if (mk & PD_BC)
      {
        if (!(mk & PD_ACC))
          y[dof] = x[dof];
      }
      else if (atomic_out)
        atomicAdd(&y[dof], sy[i]);
      else if constexpr (NT)
        __builtin_nontemporal_store(sy[i], &y[dof]);
      else
        y[dof] = sy[i];
    }
  }
#else
  {
    // (the thread index made opaque here: otherwise the list addresses are computed ahead of the cell loop and
    // held -- or spilled -- through it)
    PMG_STAMP(4); // cell loop done
    prio_level(3);
    int tw = t;
    asm volatile("" : "+v"(tw));
    uint32_t mk[ITER];
    if constexpr (!LIST_IN_LDS)
      patch_list_reload<ITER, THREADS>(mk, pdofs + off, M, tw);
    lds_barrier();
    if constexpr (LIST_IN_LDS)
    {
#pragma unroll
      for (int k = 0; k < ITER; ++k)
        mk[k] = sm[tw + k * THREADS < M ? tw + k * THREADS : M - 1];
    }
    PMG_STAMP(5); // behind the barrier that ends the accumulation
    patch_write_back<ITER, THREADS, NT>(mk, M, tw, sy, x, y, atomic_out);
    PMG_STAMP(6); // stores issued
    PMG_STAMP_FLUSH(Sh::NW);
  }
#endif
}
// ---- the hot kernel, ring form (P >= 5: ring_cfg) ------------------------------------------
//
// The column kernel above holds ONE layer of G per wavefront in registers, so a compute unit has
// 8 - 16 x 2.4 - 3.9 KB of the tensor in flight against the ~50 KB the stream needs
// (profiles/kernel_resources_r02.md), and a deeper register pipeline spills.  Here the tensor does not
// pass through registers on its way in: every item group (a wavefront; at P = 8 the two wavefronts that
// share a cell) streams its (item, layer) blocks with LDS-direct loads (global_load_lds_dwordx4, 1 KB
// per wave instruction, no destination registers) into a private ring of RD LDS slots, RD - 1 blocks
// ahead of the layer it computes, across item boundaries, from the first instruction of the kernel
// (so the first blocks arrive under the gather phase).  The blocks of a group are contiguous in memory
// in the order of use (layout `ring`, gpos()).
//
// Ordering.  LDS-direct loads count on the wave's vector-memory counter in issue order.  They are issued
// with inline assembly, which the compiler's wait insertion does not see -- deliberately: told about
// them (the builtin), it drains the counter (vmcnt(0)) before every LDS read that might alias.  So
//   * the cell loop contains no other vector-memory instruction (the patch position table is staged in
//     LDS in phase 0, kappa and the 1-D table come from LDS / scalar loads), and the waits for the ring
//     are written by hand: before layer n's values are read, `s_waitcnt vmcnt((blocks issued after n) x
//     (pieces per block and wave))`; with two waves per cell the workgroup barrier of the slice
//     hand-over follows, so the partner's pieces have landed too;
//   * a slot is refilled (block n + RD) only after the layer's fluxes have been formed from its values
//     and, with two waves per cell, after the barrier behind them: no LDS read of the old block can
//     still be in flight when the new one lands;
//   * waits the compiler places for its own loads in the gather phase count only its own loads; older
//     LDS-direct loads make such a wait stricter (the counter retires in order), never weaker.
typedef __attribute__((address_space(3))) char lds_char;
#ifndef PMG_RING_NT
#define PMG_RING_NT " nt"
#endif
__device__ __forceinline__ void lds_dma16s(const void* sbase, unsigned voff, unsigned lds_byte, unsigned long long lanes)
{
  // the lanes of `lanes` (never empty): 16 bytes from sbase + voff to LDS[lds_byte + 16 * lane].  Scalar base, 32-bit
  // per-lane offset; EXEC and M0 are set and restored inside the statement, so no control flow reaches the compiler.
  unsigned keep;
  unsigned long long keepx;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %5\n\ts_mov_b64 exec, %4\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %2, %3" PMG_RING_NT "\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(keepx)
               : "v"(voff), "s"(sbase), "s"(lanes), "s"(lds_byte)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the same for a count the optimiser folds after unrolling (an immediate in the instruction)
__device__ __forceinline__ void wait_vmcnt_n(int n)
{
  switch (n)
  {
#define PMG_W(i)                                                                                   \
  case i:                                                                                          \
    wait_vmcnt<i>();                                                                               \
    break;
    PMG_W(0) PMG_W(1) PMG_W(2) PMG_W(3) PMG_W(4) PMG_W(5) PMG_W(6) PMG_W(7) PMG_W(8) PMG_W(9) PMG_W(10) PMG_W(11)
    PMG_W(12) PMG_W(13) PMG_W(14) PMG_W(15) PMG_W(16) PMG_W(17) PMG_W(18) PMG_W(19) PMG_W(20) PMG_W(21) PMG_W(22)
    PMG_W(23) PMG_W(24)
#undef PMG_W
  default:
    wait_vmcnt<0>(); // stricter than asked for: always safe
    break;
  }
}

template <int P>
struct RShape
{
  using Sh = Shape<P>;
  static constexpr int ND = Sh::ND, N = Sh::N, K = Sh::K, MAXM = Sh::MAXM, NQ2 = Sh::NQ2, CW = Sh::CW, WPC = Sh::WPC;
  static constexpr int ITEMS = Sh::ITEMS;
  static constexpr int NG = ring_cfg(ND).groups > 0 ? ring_cfg(ND).groups : 1; // item groups per workgroup
  static constexpr int RD = ring_cfg(ND).depth > 0 ? ring_cfg(ND).depth : 1;   // ring slots per group
  static constexpr int NW = NG * WPC, THREADS = NW * 64;
  static constexpr int ITER = (MAXM + THREADS - 1) / THREADS;
  static constexpr int IPG = ring_ipg(ND, K);
  static constexpr int WL = CW * NQ2, LB = 3 * WL;     // one (item, layer) block, in double2
  static constexpr int NPIECE = (LB + 63) / 64;        // 1 KB wave instructions per block
  static constexpr int NPW = NPIECE / WPC;             // ... per wave of the group
  static_assert(NPIECE % WPC == 0, "the pieces of a block are dealt evenly to the waves of a group");
  static_assert(RD <= ND, "the prefetch distance stays inside two items");
  static_assert((RD - 1) * NPW < 64, "vmcnt range");
  static_assert((K * N) % 2 == 0, "the position table is staged in 32-bit words");
};
#ifndef PMG_RING_MINWAVES
#define PMG_RING_MINWAVES 2
#endif
#ifndef PMG_RING_MAXWAVES
#define PMG_RING_MAXWAVES 2
#endif
template <int P>
__global__ void __launch_bounds__(RShape<P>::THREADS)
    __attribute__((amdgpu_waves_per_eu(PMG_RING_MINWAVES, PMG_RING_MAXWAVES)))
    stiffness_ring_kernel(const double* __restrict__ x, double* __restrict__ y, const double2* __restrict__ G,
                          const int32_t* __restrict__ poff, const uint32_t* __restrict__ pdofs,
                          const int32_t* __restrict__ lmap_id, const uint16_t* __restrict__ lmaps,
                          const int32_t* __restrict__ pcell, const int32_t* __restrict__ pncell,
                          const double* __restrict__ kappa, const double* __restrict__ Dg, int first, int atomic_out)
{
  using R = RShape<P>;
  constexpr int ND = R::ND, N = R::N, K = R::K, NQ2 = R::NQ2, CW = R::CW, NG = R::NG, WPC = R::WPC, RD = R::RD;
  constexpr int MAXM = R::MAXM, THREADS = R::THREADS, ITER = R::ITER, WL = R::WL, LB = R::LB, NPIECE = R::NPIECE;
  constexpr int NPW = R::NPW, IPG = R::IPG;
  constexpr bool UNPAIRED = unpaired_slice_reads(P);
  __shared__ double sD[ND * ND];
  __shared__ double skap[K];
  __shared__ double sx[MAXM];
  __shared__ double sy[MAXM];
  __shared__ double sq[NG * WL];
  __shared__ double sgr[NG * WL];
  __shared__ double sgs[NG * WL];
  __shared__ uint32_t slm[K * N / 2];          // the patch's position table (uint16 pairs)
  __shared__ double2 ring[NG * RD * LB];       // [group][slot][pair][cell of the item][column]

  const int p = first + blockIdx.x;
  const int t = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = wv / WPC, h = wv - grp * WPC; // item group, wave inside the group
  const int lane64 = t & 63;
  const int nc = pncell[p];
  // items of this group: grp, grp + NG, ...; with several waves per cell every group runs the same count
  const int items_all = (nc + CW - 1) / CW;
  const int nmy = __builtin_amdgcn_readfirstlane(WPC > 1 ? (items_all + NG - 1) / NG
                                                         : (items_all > grp ? (items_all - grp + NG - 1) / NG : 0));
  const double2* Gg = G + (size_t)p * gpatch(ND, K) + (size_t)grp * IPG * ND * LB; // the group's stream
  const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)&ring[grp * RD * LB]);
  // block n of the stream -> ring slot n % RD (this wave's pieces of it)
  const int NB = nmy * ND; // blocks of the group's stream
  const unsigned voff = lane64 * 16;
  auto issue = [&](int n, int slot) {
    // Beyond the end of the stream: the same number of instructions, one lane each, re-reading the stream's first
    // 16 bytes into the (free) slot -- the wait counts of the layer loop then need no special case at the tail.
    // Everything is arithmetic on scalars: a branch inside the unrolled layer loop costs ~100 registers.
    const unsigned long long rmask = 0ull - (unsigned long long)(n < NB);
    const char* sb = reinterpret_cast<const char*>(Gg) + (((size_t)n * (LB * 16)) & rmask);
    const unsigned dst = ring0 + (unsigned)slot * (LB * 16);
#pragma unroll
    for (int i = 0; i < NPW; ++i)
    {
      const int pc = h + WPC * i; // this wave's i-th piece of the block
      const int cnt = LB - pc * 64; // lanes of the piece (>= 1)
      const unsigned long long full = cnt >= 64 ? ~0ull : ((1ull << (cnt & 63)) - 1ull);
      lds_dma16s(sb + ((size_t)(pc * 1024) & rmask), voff, dst + pc * 1024, (full & rmask) | 1ull);
    }
  };
  if (nmy > 0)
  {
#pragma unroll
    for (int n = 0; n < RD; ++n)
      issue(n, n); // RD <= ND <= blocks of the group
  }

  const int off = poff[p];
  const int M = poff[p + 1] - off; // 1 <= M <= MAXM
  const int table = lmap_id[p];

  // ---- phase 0: gather (as in the column kernel) + the position table
  {
    uint32_t m[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const int i = t + k * THREADS;
      m[k] = pdofs[off + (i < M ? i : M - 1)];
    }
    const int cellk = pcell[(size_t)p * K + (t < K ? t : K - 1)];
    const double dval = Dg[t < ND * ND ? t : ND * ND - 1];
    double xv[ITER], yv[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const uint32_t dof = m[k] & PD_MASK;
      const bool acc = !atomic_out && (m[k] & (PD_ACC | PD_BC)) == PD_ACC;
      xv[k] = x[dof];
      const double* ya = acc ? (const double*)(y + dof) : (x + dof);
      yv[k] = *ya;
    }
    const double kapk = kappa[cellk >= 0 ? cellk : 0];
    const uint32_t* lmw = reinterpret_cast<const uint32_t*>(lmaps + (size_t)table * (K * N));
    constexpr int LMW = K * N / 2, LMI = (LMW + THREADS - 1) / THREADS;
    uint32_t lmv[LMI];
#pragma unroll
    for (int k = 0; k < LMI; ++k)
      lmv[k] = lmw[t + k * THREADS < LMW ? t + k * THREADS : LMW - 1];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const int i = t + k * THREADS;
      if (i < M)
      {
        const bool acc = !atomic_out && (m[k] & (PD_ACC | PD_BC)) == PD_ACC;
        sx[i] = (m[k] & PD_BC) ? 0.0 : xv[k]; // src/laplacian.hpp:186-189
        sy[i] = acc ? yv[k] : 0.0;
      }
    }
    if (t < ND * ND)
      sD[t] = dval;
    for (int i = t; i < K; i += THREADS)
      skap[i] = (i == t) ? kapk : kappa[pcell[(size_t)p * K + i] >= 0 ? pcell[(size_t)p * K + i] : 0];
#pragma unroll
    for (int k = 0; k < LMI; ++k)
      if (t + k * THREADS < LMW)
        slm[t + k * THREADS] = lmv[k];
  }
  lds_barrier();

  // ---- cell loop
  const int lane = lane64 + 64 * h;
  const bool lane_ok = lane < WL;
  const int lw = lane_ok ? lane : WL - 1;
  const int cw = lw / NQ2;
  const int ab = lw - cw * NQ2;
  const int a = ab / ND, b = ab - a * ND;
  double Da[ND], Db[ND], DTa[ND], DTb[ND]; // D[a][.], D[b][.], D[.][a], D[.][b]
#pragma unroll
  for (int mm = 0; mm < ND; ++mm)
  {
    Da[mm] = sD[a * ND + mm];
    Db[mm] = sD[b * ND + mm];
    DTa[mm] = sD[mm * ND + a];
    DTb[mm] = sD[mm * ND + b];
  }
  double* q_s = sq + grp * WL + cw * NQ2;
  double* gr_s = sgr + grp * WL + cw * NQ2;
  double* gs_s = sgs + grp * WL + cw * NQ2;
  const uint16_t* slm16 = reinterpret_cast<const uint16_t*>(slm);
  const double2* ringg = ring + grp * RD * LB;
  auto slice_sync = [] {
    if constexpr (WPC > 1)
      lds_barrier();
    else
      wave_fence();
  };

  int slot0 = 0; // ring slot of the item's first layer: (j * ND) % RD
  for (int j = 0; j < nmy; ++j)
  {
    const int it = grp + j * NG;
    const int slot = it * CW + cw;
    const int slotc = slot < K ? slot : K - 1;
    int l[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k)
      l[k] = slm16[slotc * N + k * NQ2 + ab];
    const double kap = skap[slotc];
    double u[ND], Aq[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k)
    {
      u[k] = sx[l[k]];
      Aq[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < ND; ++k)
    {
      // block n = j * ND + k has landed once at most the blocks issued after it are outstanding
#ifndef PMG_ABL_NOWAIT
      wait_vmcnt<(RD - 1) * NPW>();
#endif
      const int sk = (slot0 + k) % RD;
      q_s[ab] = u[k];
      slice_sync();
      const double2* rs = ringg + sk * LB + lw;
      const double2 g01 = rs[0], g23 = rs[WL], g45 = rs[2 * WL];
      double qr = 0.0, qs = 0.0, qt = 0.0;
#pragma unroll
      for (int mm = 0; mm < ND; ++mm)
      {
        qr += Da[mm] * slice_load<UNPAIRED>(q_s[mm * ND + b]); // d/dx, src/laplacian.hpp:195-199
        qs += Db[mm] * slice_load<UNPAIRED>(q_s[a * ND + mm]); // d/dy, :206-210
        qt += Dg[k * ND + mm] * u[mm];                         // d/dz, :214-218
      }
      const double fr = kap * (g01.x * qr + g01.y * qs + g23.x * qt); // :233
      const double fs = kap * (g01.y * qr + g23.y * qs + g45.x * qt); // :234
      const double ft = kap * (g23.x * qr + g45.x * qs + g45.y * qt); // :235
      gr_s[ab] = fr;
      gs_s[ab] = fs;
      slice_sync();
      // the slot is free (its values are in the fluxes; with two waves per cell the barrier above has seen
      // the partner's as well): refill it with block n + RD
#ifndef PMG_ABL_NODMA
      issue(j * ND + k + RD, sk);
#endif
      double acc = 0.0;
#pragma unroll
      for (int mm = 0; mm < ND; ++mm)
      {
        acc += DTa[mm] * slice_load<UNPAIRED>(gr_s[mm * ND + b]); // :246-251
        acc += DTb[mm] * slice_load<UNPAIRED>(gs_s[a * ND + mm]); // :255-259
        Aq[mm] += Dg[k * ND + mm] * ft;                           // :263-267
      }
      Aq[k] += acc;
      slice_sync();
    }
    const bool contributes = lane_ok && slot < nc;
#pragma unroll
    for (int k = 0; k < ND; ++k)
      atomicAdd(&sy[l[k]], contributes ? Aq[k] : 0.0); // :270,277 -- in LDS (ds_add_f64)
    slot0 = (slot0 + ND) % RD;
  }
  lds_barrier();

  // ---- write back
#pragma unroll
  for (int k = 0; k < ITER; ++k)
  {
    const int i = t + k * THREADS;
    if (i < M)
    {
      const uint32_t mk = pdofs[off + i];
      const uint32_t dof = mk & PD_MASK;
      if (mk & PD_BC)
      {
        if (!(mk & PD_ACC))
          y[dof] = x[dof]; // :273-274
      }
      else if (atomic_out)
        atomicAdd(&y[dof], sy[i]);
      else
        __builtin_nontemporal_store(sy[i], &y[dof]);
    }
  }
}

// ---- the hot kernel, stream form (round 4) ---------------------------------------------------
//
// The column kernel above with the G stream of a wavefront made CONTINUOUS.  Read at the ISA level the column
// kernel keeps two layers of the tensor in flight inside an item (the compiler hoists layer k + 2 behind the fluxes
// of layer k), but an item starts cold: the position table comes from global memory, then the first layer's values
// are requested and waited for in full -- one exposed HBM round trip per item, two per wavefront and patch at P = 4,
// and another one at the start of the workgroup, behind the gather's barrier.  Here
//   * a wavefront's (item, layer) pairs form one stream: the refill behind the last two layers of an item requests
//     the first two layers of the wavefront's next item;
//   * the first two layers of the wavefront's first item are requested BEFORE the gather's barrier, behind the
//     gather's own loads (the counter retires in order: the gather never waits for the tensor);
//   * the position table of the patch is staged in LDS with the gather (one 16-byte load per thread), so an item
//     starts with LDS latencies only;
//   * kappa multiplies the item's five input values once instead of every flux (linear: same value to rounding).
// Beyond the end of its stream a wavefront re-reads ONE 16-byte element (every lane the same address): the load
// counts stay static, no branch in the layer loop, no traffic.
template <int P, bool NT>
__global__ void __launch_bounds__(Shape<P>::WTHREADS, min_waves_per_simd<P>())
    stiffness_stream_kernel(const double* __restrict__ x, double* __restrict__ y, const double2* __restrict__ G,
                            const int32_t* __restrict__ poff, const uint32_t* __restrict__ pdofs,
                            const int32_t* __restrict__ lmap_id, const uint16_t* __restrict__ lmaps,
                            const int32_t* __restrict__ pcell, const int32_t* __restrict__ pncell,
                            const double* __restrict__ kappa, const double* __restrict__ Dg, int first, int atomic_out)
{
  using Sh = Shape<P>;
  constexpr int ND = Sh::ND, N = Sh::N, K = Sh::K, NQ2 = Sh::NQ2, CW = Sh::CW, NG = Sh::NG;
  constexpr int MAXM = Sh::MAXM, THREADS = Sh::WTHREADS, ITER = Sh::WITER;
  constexpr int WL = CW * NQ2;
  static_assert(Sh::WPC == 1 && !gflat(ND), "stream kernel: one wavefront per item, default or dense G layout");
  static_assert((K * N * 2) % 16 == 0, "the position table is staged in 16-byte pieces");
  constexpr bool UNPAIRED = unpaired_slice_reads(P);
  constexpr bool DENSE = gdense(ND);
  constexpr int GPS = DENSE ? WL : NQ2;      // stride between the three pairs of a layer
  constexpr int GLS = 3 * GPS;               // ... between two layers
  constexpr int LMV = K * N * 2 / 16, LMI = (LMV + THREADS - 1) / THREADS;
  __shared__ double sD[ND * ND];
  __shared__ double skap[K];
  __shared__ double sx[MAXM];
  __shared__ double sy[MAXM];
  // the three nd x nd slices of a wavefront's item (values, x flux, y flux) side by side: their addresses differ by
  // constants that fold into the DS instructions' offset fields (three address registers instead of seven)
  __shared__ double ssl[NG * 3 * WL];
  __shared__ uint4 slm4[LMV]; // the patch's position table [slot][layer][a*nd+b] uint16

  PMG_STAMP_DECL;
  PMG_STAMP(0); // entry
  const int p = first + blockIdx.x;
  const int t = threadIdx.x;
  const int off = poff[p];
  const int M = poff[p + 1] - off; // 1 <= M <= MAXM
  const int table = lmap_id[p];
  const int nc = pncell[p];

  // the wavefront's lanes and its stream of items
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const bool lane_ok = lane < WL;
  const int lw = lane_ok ? lane : WL - 1;
  const int cw = lw / NQ2;
  const int ab = lw - cw * NQ2;
  const int a = ab / ND, b = ab - a * ND;
  const int items = (nc + CW - 1) / CW;
  const int nmy = items > wave ? (items - wave + NG - 1) / NG : 0; // items of this wavefront (uniform)
  const double2* Gp = G + (size_t)p * gpatch(ND, K);
  // Element (layer 0, pair 0) of this lane in its j-th item, as a 32-bit offset from the patch's tensor (a scalar
  // base + 32-bit lane offsets: 64-bit per-lane pointers cost two registers each, and a spilled one is reloaded with a
  // wait that drains the whole stream); beyond the stream: the patch's first element, every lane the same.
  auto gitem = [&](int j) -> unsigned {
    const int it = wave + j * NG;
    const bool real = j < nmy;
    if constexpr (DENSE)
      return real ? (unsigned)(it * (ND * 3 * WL) + lw) : 0u;
    else
    {
      const int slot = it * CW + cw;
      return real ? (unsigned)((slot < K ? slot : K - 1) * 3 * N + ab) : 0u;
    }
  };
  double2 gq[2][3];
  auto gfetch = [&](int s, unsigned base, int layer) {
    gq[s][0] = gload<NT>(Gp + (base + (unsigned)(layer * GLS)));
    gq[s][1] = gload<NT>(Gp + (base + (unsigned)(layer * GLS + GPS)));
    gq[s][2] = gload<NT>(Gp + (base + (unsigned)(layer * GLS + 2 * GPS)));
  };

  // ---- phase 0: gather (unconditional loads, clamped indices: counted vmcnt waits)
  {
    uint32_t m[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const int i = t + k * THREADS;
      m[k] = pdofs[off + (i < M ? i : M - 1)];
    }
    const int cellk = pcell[(size_t)p * K + (t < K ? t : K - 1)];
    const double dval = Dg[t < ND * ND ? t : ND * ND - 1];
    const uint4* lmt = reinterpret_cast<const uint4*>(lmaps + (size_t)table * (K * N));
    uint4 lmv[LMI];
#pragma unroll
    for (int k = 0; k < LMI; ++k)
      lmv[k] = lmt[t + k * THREADS < LMV ? t + k * THREADS : LMV - 1];
    double xv[ITER], yv[ITER];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const uint32_t dof = m[k] & PD_MASK;
      const bool acc = !atomic_out && (m[k] & (PD_ACC | PD_BC)) == PD_ACC;
      xv[k] = x[dof];
      const double* ya = acc ? (const double*)(y + dof) : (x + dof);
      yv[k] = *ya;
    }
    const double kapk = kappa[cellk >= 0 ? cellk : 0];
#ifndef PMG_STREAM_NO_EARLY
    // the head of the wavefront's tensor stream, behind the gather's loads
    {
      const unsigned g0 = gitem(0);
      gfetch(0, g0, 0);
      gfetch(1, g0, ND > 1 ? 1 : 0);
    }
#endif
#pragma unroll
    for (int k = 0; k < LMI; ++k)
      if (t + k * THREADS < LMV)
        slm4[t + k * THREADS] = lmv[k];
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      const int i = t + k * THREADS;
      if (i < M)
      {
        const bool acc = !atomic_out && (m[k] & (PD_ACC | PD_BC)) == PD_ACC;
        sx[i] = (m[k] & PD_BC) ? 0.0 : xv[k]; // src/laplacian.hpp:186-189
        sy[i] = acc ? yv[k] : 0.0;
      }
    }
    if (t < ND * ND)
      sD[t] = dval;
    for (int i = t; i < K; i += THREADS)
      skap[i] = (i == t) ? kapk : kappa[pcell[(size_t)p * K + i] >= 0 ? pcell[(size_t)p * K + i] : 0];
  }
  PMG_STAMP(1); // gathered values written to LDS
  lds_barrier();
  PMG_STAMP(2); // behind the gather's barrier

  // ---- cell loop
  double Da[ND], Db[ND], DTa[ND], DTb[ND]; // D[a][.], D[b][.], D[.][a], D[.][b]
#pragma unroll
  for (int mm = 0; mm < ND; ++mm)
  {
    Da[mm] = sD[a * ND + mm];
    Db[mm] = sD[b * ND + mm];
    DTa[mm] = sD[mm * ND + a];
    DTb[mm] = sD[mm * ND + b];
  }
  double* q_s = ssl + wave * (3 * WL) + cw * NQ2;
  double* gr_s = q_s + WL;
  double* gs_s = q_s + 2 * WL;
#ifdef PMG_STREAM_NO_EARLY // measurement: the stream starts behind the barrier (the early loads of the first wavefronts
                           // queue ahead of the gathers of the last ones on the same compute unit)
  {
    const unsigned g0 = gitem(0);
    gfetch(0, g0, 0);
    gfetch(1, g0, ND > 1 ? 1 : 0);
  }
#endif
  const uint16_t* slm = reinterpret_cast<const uint16_t*>(slm4);

  // Slots: layer k of an item lives in slot k & 1.  On entry to an item its layers 0 and 1 are in flight in slots 0
  // and 1; the refill behind layer k requests layer k + 2 of the item or, past its end, the layer of the NEXT item that
  // belongs in the freed slot (for odd nd that is layer 1 first, then layer 0: every item then starts alike and one
  // body serves the whole stream).
  for (int j = 0; j < nmy; ++j)
  {
#ifdef PMG_STAMPS
    if (j > 0)
      PMG_STAMP(3); // start of the last item
#endif
    const int it = wave + j * NG;
    const int slot = it * CW + cw;
    const int slotc = slot < K ? slot : K - 1;
    const unsigned gthis = gitem(j), gnext = gitem(j + 1);
    int l[ND]; // the column's positions in the patch list
#pragma unroll
    for (int k = 0; k < ND; ++k)
      l[k] = slm[slotc * N + k * NQ2 + ab];
    // kappa multiplies the cell's INPUT once (the operator is linear in it: same value to rounding) instead of every
    // flux; nothing of it is held through the layer loop
    const double kap = skap[slotc];
    double u[ND], Aq[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k)
    {
      u[k] = kap * sx[l[k]];
      Aq[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < ND; ++k)
    {
      const int s = k & 1;
      q_s[ab] = u[k];
      wave_fence();
      double qr = 0.0, qs = 0.0, qt = 0.0;
#pragma unroll
      for (int mm = 0; mm < ND; ++mm)
      {
        qr += Da[mm] * slice_load<UNPAIRED>(q_s[mm * ND + b]); // d/dx: sum over a, :195-199
        qs += Db[mm] * slice_load<UNPAIRED>(q_s[a * ND + mm]); // d/dy: sum over b, :206-210
        qt += Dg[k * ND + mm] * u[mm];                         // d/dz: registers, uniform table, :214-218
      }
      const double2 g01 = gq[s][0], g23 = gq[s][1], g45 = gq[s][2];
      const double fr = g01.x * qr + g01.y * qs + g23.x * qt; // :233 (kappa: in u, above)
      const double fs = g01.y * qr + g23.y * qs + g45.x * qt; // :234
      const double ft = g23.x * qr + g45.x * qs + g45.y * qt; // :235
      if (k + 2 < ND)
        gfetch(s, gthis, k + 2);
      else
        gfetch(s, gnext, ND > 1 ? s : 0);
      gr_s[ab] = fr;
      gs_s[ab] = fs;
      wave_fence();
      double acc = 0.0;
#pragma unroll
      for (int mm = 0; mm < ND; ++mm)
      {
        acc += DTa[mm] * slice_load<UNPAIRED>(gr_s[mm * ND + b]); // :246-251
        acc += DTb[mm] * slice_load<UNPAIRED>(gs_s[a * ND + mm]); // :255-259
        Aq[mm] += Dg[k * ND + mm] * ft;                           // :263-267
      }
      Aq[k] += acc;
      wave_fence();
    }
    const bool contributes = lane_ok && slot < nc; // lanes without a cell add an exact zero
#pragma unroll
    for (int k = 0; k < ND; ++k)
      atomicAdd(&sy[l[k]], contributes ? Aq[k] : 0.0); // :270,277 -- in LDS (ds_add_f64)
  }
  // ---- write back (plain stores; the accumulator started from the earlier colours' y)
  {
    // (the thread index made opaque here: otherwise the list addresses are computed ahead of the cell loop and
    // held -- or spilled -- through it)
    PMG_STAMP(4); // cell loop done
    int tw = t;
    asm volatile("" : "+v"(tw));
    uint32_t mk[ITER];
    patch_list_reload<ITER, THREADS>(mk, pdofs + off, M, tw);
    lds_barrier();
    PMG_STAMP(5); // behind the barrier that ends the accumulation
    patch_write_back<ITER, THREADS, NT>(mk, M, tw, sy, x, y, atomic_out);
    PMG_STAMP(6); // stores issued
    PMG_STAMP_FLUSH(Sh::NW);
  }
}

// ---- the hot kernel, chain form (round 4) ------------------------------------------------------
//
// One PERSISTENT workgroup of sixteen wavefronts per chain of patches (patches.hpp, ChainPlan), one wavefront per item
// of the patch in hand.  What the column kernel does one phase after the other -- list -> gather -> cells -> list ->
// store, eight launches of two generations of workgroups -- overlaps here:
//   * while the wavefronts are in the cell loop of patch c, the dof list of patch c + 1 lands in LDS (LDS-direct
//     loads: no registers, nobody waits);
//   * leaving the cell loop a wavefront requests its share of patch c + 1's x and y values; they are in flight
//     during the closing barrier and the write-back of patch c and are put into the SECOND pair of patch arrays;
//   * the tensor is one stream per wavefront across the patches of the chain (the refill behind the last two layers
//     of an item requests the first two layers of the wavefront's item in the NEXT patch), so it keeps flowing
//     through both barriers;
//   * dofs shared by consecutive patches of the chain go from one accumulator to the next inside LDS.
// The LDS-direct loads.  Told about them (the builtin), the compiler drains the memory counter before the next LDS
// read of ANY array (measured in the ISA: `s_waitcnt vmcnt(0)` at the top of the cell loop); issued with inline
// assembly it does not see them, and since the counter retires in order every wait it places for a load of its own
// that is YOUNGER than them... is unaffected, for an OLDER one becomes a wait for them as well.  So they sit where no
// such wait follows: behind the layer loop (the last wait for the tensor lies before it), in front of the gather's
// loads, whose waits they precede anyway -- and they fetch the lists of the patch TWO ahead, which nobody reads before
// the iteration after this one (three list buffers, two buffers of carry words and cell ids).
__device__ __forceinline__ void lds_dma4s(const void* sbase, unsigned voff, unsigned lds_byte)
{
  // every lane: 4 bytes from sbase + voff to LDS[lds_byte + 4 * lane]
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_byte)
               : "memory");
}

template <int P>
struct ChainShape
{
  using Sh = Shape<P>;
  static constexpr int NW = 16, THREADS = NW * 64;
  static constexpr int ITER = (Sh::MAXM + THREADS - 1) / THREADS;
  static constexpr int LCAP = (Sh::MAXM + 63) / 64 * 64; // list entries the LDS-direct loads write (whole wavefronts)
  static constexpr bool OK = Sh::WPC == 1 && !gflat(Sh::ND) && !gdense(Sh::ND) && !gring(Sh::ND) && Sh::ITEMS <= NW
                             && Sh::ND >= 2 && Sh::K <= 64;
};
constexpr bool chain_form(int P) { return P == 4; } // degrees the chain kernel is built for
#ifndef PMG_CHAIN_DEFAULT
#define PMG_CHAIN_DEFAULT 0 // the mode when PMG_CHAIN is not set (pmg_laplacian_create_ordered)
#endif

template <int P, bool NT>
__global__ void __launch_bounds__(ChainShape<P>::THREADS)
    stiffness_chain_kernel(const double* __restrict__ x, double* __restrict__ y, const double2* __restrict__ G,
                           const int32_t* __restrict__ poff, const uint32_t* __restrict__ cdofs,
                           const uint32_t* __restrict__ ccar, const int32_t* __restrict__ lmap_id,
                           const uint16_t* __restrict__ lmaps, const int32_t* __restrict__ pcell,
                           const int32_t* __restrict__ pncell, const double* __restrict__ kappa,
                           const double* __restrict__ Dg, const int32_t* __restrict__ chain_off,
                           const int32_t* __restrict__ chain_patch, int first_chain)
{
  using Sh = Shape<P>;
  using Ch = ChainShape<P>;
  constexpr int ND = Sh::ND, N = Sh::N, K = Sh::K, NQ2 = Sh::NQ2, CW = Sh::CW;
  constexpr int MAXM = Sh::MAXM, THREADS = Ch::THREADS, ITER = Ch::ITER, LCAP = Ch::LCAP, NW = Ch::NW;
  constexpr int WL = CW * NQ2;
  constexpr int GPS = NQ2, GLS = 3 * GPS;
  constexpr bool UNPAIRED = unpaired_slice_reads(P);
  static_assert(Ch::OK, "chain kernel: one wavefront per item, default G layout, at most sixteen items per patch");
  __shared__ double sD[ND * ND];
  __shared__ double skap[2 * K + 1]; // (+ 1: the spare element the lanes without an entry write)
  __shared__ double sx[2 * MAXM + 1];
  __shared__ double sy[2 * MAXM + 1];
  __shared__ double ssl[NW * 3 * WL];
  __shared__ uint32_t sm[3 * LCAP];   // dof lists of the patch in hand, the next one and the one after
  __shared__ uint32_t scar[2 * LCAP]; // carry positions / read-y flags of the next patch and the one after
  __shared__ uint32_t spc[2 * 64];    // cell ids, likewise

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const bool lane_ok = lane < WL;
  const int lw = lane_ok ? lane : WL - 1;
  const int cw = lw / NQ2;
  const int ab = lw - cw * NQ2;
  const int a = ab / ND, b = ab - a * ND;
  const int slot = wave * CW + cw;
  const int slotc = slot < K ? slot : K - 1;
  const int ch = first_chain + blockIdx.x;
  const int c0 = chain_off[ch], nck = chain_off[ch + 1] - c0;
#if defined(PMG_CHAIN_PRIO) && PMG_CHAIN_PRIO <= 2 // experiment: issue priority against the age order on a SIMD
  switch (PMG_CHAIN_PRIO == 1 ? wave >> 2 : 3 - (wave >> 2))
  {
  case 1: __builtin_amdgcn_s_setprio(1); break;
  case 2: __builtin_amdgcn_s_setprio(2); break;
  case 3: __builtin_amdgcn_s_setprio(3); break;
  default: break;
  }
#endif
  // PMG_CHAIN_PRIO = 3: priority by PROGRESS -- a wavefront in an earlier layer of its item outranks one in a later
  // layer (3 2 1 0 0 ...), so whoever the age order starves catches up; everything outside the cell loop at 3
  auto chain_prio = [](int v) {
#if defined(PMG_CHAIN_PRIO) && PMG_CHAIN_PRIO == 3
    switch (v)
    {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
#else
    (void)v;
#endif
  };
  chain_prio(3);
  PMG_STAMP_DECL; // (diagnostic build: the phases of the chain's middle patch, tools/stamp_chain.py)
  PMG_STAMP(0);   // entry
#ifdef PMG_STAMPS
#define PMG_CSTAMP(i)                                                                                                 \
  if (c == nck / 2)                                                                                                   \
  PMG_STAMP(i)
#else
#define PMG_CSTAMP(i)
#endif

  // element (layer 0, pair 0) of this lane's column in patch p, as a 32-bit offset from G
  auto gitem = [&](int p) -> unsigned { return (unsigned)p * (unsigned)(K * 3 * N) + (unsigned)(slotc * 3 * N + ab); };
  double2 gq[2][3];
  auto gfetch = [&](int s, unsigned base, int layer) {
    gq[s][0] = gload<NT>(G + (base + (unsigned)(layer * GLS)));
    gq[s][1] = gload<NT>(G + (base + (unsigned)(layer * GLS + GPS)));
    gq[s][2] = gload<NT>(G + (base + (unsigned)(layer * GLS + 2 * GPS)));
  };
  const unsigned sm0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)&sm[wave * 64]);
  const unsigned scar0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)&scar[wave * 64]);
  const unsigned spc0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)&spc[0]);
  // the lists of the chain's patch number cc into the buffers it owns (entries past the list's end repeat the last one)
  auto dma_lists = [&](int cc) {
    const int pn = chain_patch[c0 + (cc < nck ? cc : nck - 1)];
    const int offn = poff[pn], Mn = poff[pn + 1] - offn;
    const int b3 = cc % 3, b2 = cc & 1;
#pragma unroll
    for (int k = 0; k < ITER; ++k)
    {
      if (k * THREADS + wave * 64 < LCAP) // (wave-uniform)
      {
        const int i = t + k * THREADS;
        const unsigned vo = 4u * (unsigned)(i < Mn ? i : Mn - 1);
        lds_dma4s(cdofs + offn, vo, sm0 + 4u * (unsigned)(b3 * LCAP + k * THREADS));
        lds_dma4s(ccar + offn, vo, scar0 + 4u * (unsigned)(b2 * LCAP + k * THREADS));
      }
    }
    if (wave == 0)
    {
      // (the lane number made opaque: hoisted out of the loop, this offset is held -- in the event, spilled -- through
      // the cell loop, and its reload drains the memory counter)
      int lo = lane;
      asm volatile("" : "+v"(lo));
      lds_dma4s(pcell + (size_t)pn * K, 4u * (unsigned)(lo < K ? lo : K - 1), spc0 + 256u * (unsigned)b2);
    }
  };
  // What a thread holds of the next patch between the cell loop and barrier #2 lives in the loop body's scope: nothing
  // of it is carried around the loop (a value defined under one condition and used under another is a loop-carried
  // register as far as the allocator is concerned, live through the cell loop: spilled).  For the same reason the
  // next-patch steps are UNCONDITIONAL: behind its last patch a chain gathers that patch once more into the free
  // buffers (reads only; nobody uses them).
#define PMG_CHAIN_GATHER(pn_, b3_, b2_)                                                                                \
  uint32_t m[ITER], cf[ITER];                                                                                         \
  double xv[ITER], yv[ITER], kapn;                                                                                    \
  {                                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < ITER; ++k)                                                                  \
    {                                                                                                                 \
      /* (a thread past the end of the buffer re-reads its OWN first entry: an entry another wavefront fetched  */     \
      /* may not have landed yet, and what is there instead is no dof number) */                                      \
      const int i = t + k * THREADS < LCAP ? t + k * THREADS : t;                                                     \
      m[k] = sm[(b3_) * LCAP + i];                                                                                    \
      cf[k] = scar[(b2_) * LCAP + i];                                                                                 \
    }                                                                                                                 \
    /* (wave 0 put the cell ids there, and only its lanes use them) */                                                \
    int lo_ = lane;                                                                                                   \
    asm volatile("" : "+v"(lo_));                                                                                     \
    const int cellk = wave == 0 ? (int)spc[(b2_) * 64 + (lo_ < K ? lo_ : K - 1)] : 0;                                 \
    _Pragma("unroll") for (int k = 0; k < ITER; ++k)                                                                  \
    {                                                                                                                 \
      const uint32_t dof = m[k] & CD_MASK;                                                                            \
      xv[k] = x[dof];                                                                                                 \
      const double* ya = (cf[k] & CC_ACC) ? (const double*)(y + dof) : (x + dof);                                     \
      yv[k] = *ya;                                                                                                    \
    }                                                                                                                 \
    /* (the column's offset in the position table from the opaque lane number: a scalar base + a 32-bit offset, */   \
    /* nothing of it held through the cell loop) */                                                                   \
    const int lw_ = lo_ < WL ? lo_ : WL - 1, cw_ = lw_ / NQ2, sl_ = wave * CW + cw_;                                  \
    const unsigned lmo = (unsigned)((sl_ < K ? sl_ : K - 1) * N + (lw_ - cw_ * NQ2));                                 \
    const uint16_t* lm = lmaps + (size_t)lmap_id[pn_] * (K * N);                                                      \
    _Pragma("unroll") for (int k = 0; k < ND; ++k) ln[k] = lm[lmo + (unsigned)(k * NQ2)];                             \
    kapn = kappa[cellk >= 0 ? cellk : 0];                                                                             \
  }
  // (no branch: a load whose only use sits under a condition is SUNK into it -- behind the barrier, with a full wait
  // in front of its use; the lanes past the end of the list write a spare element instead)
#define PMG_CHAIN_INIT(buf_, Mn_)                                                                                      \
  {                                                                                                                   \
    _Pragma("unroll") for (int k = 0; k < ITER; ++k)                                                                  \
    {                                                                                                                 \
      const int i = t + k * THREADS;                                                                                  \
      const int at = i < (Mn_) ? (buf_) * MAXM + i : 2 * MAXM;                                                        \
      sx[at] = (m[k] & CD_BC) ? 0.0 : xv[k]; /* src/laplacian.hpp:186-189 */                                          \
      const unsigned cpos = cf[k] & 0xffffu;                                                                          \
      /* the sum the previous patch of the chain left for this dof */                                                 \
      const double carry = sy[((buf_) ^ 1) * MAXM + (cpos != CC_NONE ? cpos : 0u)];                                   \
      sy[at] = ((cf[k] & CC_ACC) ? yv[k] : 0.0) + (cpos != CC_NONE ? carry : 0.0);                                    \
    }                                                                                                                 \
    skap[t < K ? (buf_) * K + t : 2 * K] = kapn;                                                                      \
    /* the column's positions as indices into the double-buffered patch arrays: consumed HERE, in front of the      */ \
    /* write-back -- left for the top of the next cell loop, their wait is one for the stores issued in between     */ \
    _Pragma("unroll") for (int k = 0; k < ND; ++k) ln[k] += (buf_) * MAXM;                                            \
  }

  // ---- prologue: the first patch of the chain, gathered as every later one will be
  int ln[ND]; // the positions of this lane's column in the list of the patch the next cell loop works on
  int p = chain_patch[c0];
  int M = poff[p + 1] - poff[p];
  {
    dma_lists(0);
    wait_vmcnt<0>();
    dma_lists(1);
    const double dval = Dg[t < ND * ND ? t : ND * ND - 1];
    PMG_CHAIN_GATHER(p, 0, 0)
    {
      const unsigned g0 = gitem(p);
      gfetch(0, g0, 0);
      gfetch(1, g0, 1);
    }
    if (t < ND * ND)
      sD[t] = dval;
    PMG_CHAIN_INIT(0, M) // (no carry in a chain's first patch)
  }
  lds_barrier();

  double Da[ND], Db[ND], DTa[ND], DTb[ND]; // D[a][.], D[b][.], D[.][a], D[.][b]
#pragma unroll
  for (int mm = 0; mm < ND; ++mm)
  {
    Da[mm] = sD[a * ND + mm];
    Db[mm] = sD[b * ND + mm];
    DTa[mm] = sD[mm * ND + a];
    DTb[mm] = sD[mm * ND + b];
  }
  double* q_s = ssl + wave * (3 * WL) + cw * NQ2;
  double* gr_s = q_s + WL;
  double* gs_s = q_s + 2 * WL;

  for (int c = 0; c < nck; ++c)
  {
    const int cur = c & 1, nxt = cur ^ 1;
    const int pn = c + 1 < nck ? chain_patch[c0 + c + 1] : p;
    const int Mn = poff[pn + 1] - poff[pn];
    const int nc = pncell[p];
    const unsigned gthis = gitem(p), gnext = gitem(pn);
    PMG_CSTAMP(1); // top of the patch
    // Every wavefront runs the cell loop, also one whose item lies beyond the patch's cells (a short patch at the edge
    // of the cell list: clamped slot, exact zeros added): a branch around the loop makes the tensor registers values
    // that meet at a join, and the copies there wait for the loads in flight -- the stream would stop at every patch.
    {
      int l[ND];
#pragma unroll
      for (int k = 0; k < ND; ++k)
        l[k] = ln[k];
      const double kap = skap[cur * K + slotc];
      double u[ND], Aq[ND];
#pragma unroll
      for (int k = 0; k < ND; ++k)
      {
        u[k] = kap * sx[l[k]];
        Aq[k] = 0.0;
      }
#pragma unroll
      for (int k = 0; k < ND; ++k)
      {
        const int s = k & 1;
        chain_prio(k < 3 ? 3 - k : 0);
        q_s[ab] = u[k];
        wave_fence();
        double qr = 0.0, qs = 0.0, qt = 0.0;
#pragma unroll
        for (int mm = 0; mm < ND; ++mm)
        {
          qr += Da[mm] * slice_load<UNPAIRED>(q_s[mm * ND + b]); // d/dx: sum over a, :195-199
          qs += Db[mm] * slice_load<UNPAIRED>(q_s[a * ND + mm]); // d/dy: sum over b, :206-210
          qt += Dg[k * ND + mm] * u[mm];                         // d/dz: registers, uniform table, :214-218
        }
        const double2 g01 = gq[s][0], g23 = gq[s][1], g45 = gq[s][2];
        const double fr = g01.x * qr + g01.y * qs + g23.x * qt; // :233 (kappa: in u, above)
        const double fs = g01.y * qr + g23.y * qs + g45.x * qt; // :234
        const double ft = g23.x * qr + g45.x * qs + g45.y * qt; // :235
        if (k + 2 < ND)
          gfetch(s, gthis, k + 2);
        else
          gfetch(s, gnext, s); // the head of the wavefront's item in the next patch, into the slot it belongs in
        gr_s[ab] = fr;
        gs_s[ab] = fs;
        wave_fence();
        double acc = 0.0;
#pragma unroll
        for (int mm = 0; mm < ND; ++mm)
        {
          acc += DTa[mm] * slice_load<UNPAIRED>(gr_s[mm * ND + b]); // :246-251
          acc += DTb[mm] * slice_load<UNPAIRED>(gs_s[a * ND + mm]); // :255-259
          Aq[mm] += Dg[k * ND + mm] * ft;                           // :263-267
        }
        Aq[k] += acc;
        wave_fence();
      }
      chain_prio(3);
      const bool contributes = lane_ok && slot < nc; // lanes without a cell add an exact zero
#pragma unroll
      for (int k = 0; k < ND; ++k)
        atomicAdd(&sy[l[k]], contributes ? Aq[k] : 0.0); // :270,277 -- in LDS (ds_add_f64)
    }
    PMG_CSTAMP(2); // cell loop done
    dma_lists(c + 2); // (behind the last wait for the tensor, in front of the gather: see above)
    PMG_CHAIN_GATHER(pn, (c + 1) % 3, nxt)
    PMG_CSTAMP(3); // next patch's values requested
    lds_barrier(); // #1: the sums of patch c are complete
    PMG_CSTAMP(4); // behind barrier #1
    // the next patch first (its values have been in flight since the wavefront left the cell loop), then the stores:
    // nothing of this iteration waits behind them
    PMG_CHAIN_INIT(nxt, Mn)
    PMG_CSTAMP(5); // next patch's values in LDS
    {
      bool bc_row = false;
#pragma unroll
      for (int k = 0; k < ITER; ++k)
      {
        const int i = t + k * THREADS;
        const uint32_t mk = sm[(c % 3) * LCAP + (i < LCAP ? i : t)];
        const bool mine = i < M;
        if (mine && !(mk & (CD_BC | CD_SKIP)))
        {
          const double v = sy[cur * MAXM + i];
          if constexpr (NT)
            __builtin_nontemporal_store(v, &y[mk & CD_MASK]);
          else
            y[mk & CD_MASK] = v;
        }
        bc_row |= mine && (mk & CD_BCFIRST);
      }
      if (__builtin_amdgcn_ballot_w64(bc_row) != 0) // wave-uniform: interior patches never enter
      {
#pragma unroll
        for (int k = 0; k < ITER; ++k)
        {
          const int i = t + k * THREADS;
          const uint32_t mk = sm[(c % 3) * LCAP + (i < LCAP ? i : t)];
          if (i < M && (mk & CD_BCFIRST))
            y[mk & CD_MASK] = x[mk & CD_MASK]; // :273-274
        }
      }
    }
    PMG_CSTAMP(6); // stores issued
    lds_barrier(); // #2: patch c + 1 is in LDS; the buffers of patch c are free
    p = pn;
    M = Mn;
  }
  PMG_STAMP_FLUSH(NW); // 7: end of the chain, stores acknowledged
#undef PMG_CHAIN_GATHER
#undef PMG_CHAIN_INIT
#undef PMG_CSTAMP
}

__global__ void zero_list_kernel(int n, const int32_t* __restrict__ idx, double* __restrict__ y)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    y[idx[i]] = 0.0;
}

// ---- matrix-free diagonal (replaces the CSR detour of examples/pmg/main.cpp:274-279) ----
__global__ void diagonal_kernel(long long slot0, long long nslots, int nd, int K,
                                const int32_t* __restrict__ pcell,
                                const double2* __restrict__ G, const int32_t* __restrict__ dofmap,
                                const int8_t* __restrict__ bc, const double* __restrict__ kappa,
                                const double* __restrict__ D, double* __restrict__ diag)
{
  const int N = nd * nd * nd, nsq = nd * nd;
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nslots * N)
    return;
  long long slot = gid / N;
  int t = (int)(gid - slot * N);
  slot += slot0;
  int cell = pcell[slot];
  if (cell < 0)
    return;
  int a = t / nsq, b = (t - a * nsq) / nd, c = t - a * nsq - b * nd;
  auto Gq = [&](int q, int pair) { return G[gpos(nd, K, slot, q, pair)]; };
  double s = 0.0;
  for (int q = 0; q < nd; ++q)
  {
    double da = D[q * nd + a], db = D[q * nd + b], dc = D[q * nd + c];
    s += da * da * Gq(q * nsq + b * nd + c, 0).x; // G00 at (q,b,c)
    s += db * db * Gq(a * nsq + q * nd + c, 1).y; // G11 at (a,q,c)
    s += dc * dc * Gq(a * nsq + b * nd + q, 2).y; // G22 at (a,b,q)
  }
  double daa = D[a * nd + a], dbb = D[b * nd + b], dcc = D[c * nd + c];
  s += 2.0
       * (Gq(t, 0).y * daa * dbb + Gq(t, 1).x * daa * dcc + Gq(t, 2).x * dbb * dcc);
  int32_t dof = dofmap[(size_t)cell * N + t];
  if (!bc[dof])
    atomicAdd(&diag[dof], kappa[cell] * s);
}

__global__ void diag_invert_kernel(int n, const int8_t* __restrict__ bc, double* __restrict__ d)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
  {
    double v = d[i];
    d[i] = bc[i] ? 1.0 : (v != 0.0 ? 1.0 / v : 0.0);
  }
}

// ---- GLL-collocated load vector ----
__global__ void rhs_kernel(long long nslots, int nq, const int32_t* __restrict__ pcell,
                           const double* __restrict__ xgeom,
                           const int32_t* __restrict__ geom_dofmap,
                           const double* __restrict__ dphi, const double* __restrict__ w,
                           const int32_t* __restrict__ dofmap, const int8_t* __restrict__ bc,
                           const double* __restrict__ kappa, const double* __restrict__ f,
                           double* __restrict__ b)
{
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nslots * nq)
    return;
  long long slot = gid / nq;
  int q = (int)(gid - slot * nq);
  int c = pcell[slot];
  if (c < 0)
    return;
  int32_t dof = dofmap[(size_t)c * nq + q];
  if (bc[dof])
    return; // set_bc: b[bc] = 0 (b is zeroed first)
  double K[3][3], detJ;
  jacobian(xgeom, geom_dofmap + (size_t)c * 8, dphi, nq, q, K, detJ);
  atomicAdd(&b[dof], kappa[c] * w[q] * detJ * f[dof]);
}

// geometry of the patches [first, first + count) into the (batch) buffer; returns the base pointer
// to hand to kernels that index G by absolute patch slot
const double2* batch_geometry(pmg_laplacian op, int first, int count, hipStream_t s)
{
  double2* base = op->G - (size_t)first * gpatch(op->nd, op->K);
  const long long nslots = (long long)count * op->K, n = nslots * op->N;
  if (n > 0)
    geometry_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((long long)first * op->K, nslots, op->nd, op->K,
                                                              op->pcell, op->xgeom, op->geom_dofmap,
                                                              op->dphi_geom, op->gweights, base);
  return base;
}

// f(first patch, patch count, G base) over all patches: once with the resident tensor, or batch by
// batch with the tensor recomputed into the batch buffer
template <typename F>
int for_each_geometry_chunk(pmg_laplacian op, hipStream_t s, F f)
{
  if (op->npatch == 0)
    return PMG_OK;
  if (op->batch_patches <= 0)
    f(0, op->npatch, op->G);
  else
    for (int first = 0; first < op->npatch; first += op->batch_patches)
    {
      const int count = std::min(op->batch_patches, op->npatch - first);
      f(first, count, batch_geometry(op, first, count, s));
    }
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

// tensor bytes above which the stream is non-temporal.  The Infinity Cache holds 256 MiB, but inside a V-cycle the
// smoother's vectors pass through it between two applications: measured per cycle, default / nt policy, at 24^3
// cells (83 MB) 0.417 / 0.445 ms, 32^3 (196 MB) 0.821 / 0.809, 40^3 1.59 / 1.46, 64^3 5.71 / 5.10.
// PMG_STREAM_POLICY=0 / 1 forces the default / the streaming policy (measurements).
bool streams_past_the_cache(long long tensor_bytes)
{
  if (const char* e = std::getenv("PMG_STREAM_POLICY"))
    return e[0] != '0';
  return tensor_bytes > (128LL << 20);
}

template <int P>
int launch_stiffness(pmg_laplacian op, const double* x, double* y, int first, int count,
                     int atomic_out, hipStream_t s)
{
  if (count <= 0)
    return PMG_OK;
  if (op->batch_patches > 0 && op->geometry_mode == 0 && count > op->batch_patches)
  {
    for (int f = first; f < first + count; f += op->batch_patches) // :384-412
      PMG_TRY(launch_stiffness<P>(op, x, y, f, std::min(op->batch_patches, first + count - f), atomic_out, s));
    return PMG_OK;
  }
  const double2* G = op->G;
  if (op->batch_patches > 0 && op->geometry_mode == 0)
    G = batch_geometry(op, first, count, s); // :391-396
  {
    if constexpr (gring(P + 1))
    {
      if (op->geometry_mode != 1)
      {
        stiffness_ring_kernel<P><<<count, RShape<P>::THREADS, 0, s>>>(x, y, G, op->poff, op->pdofs, op->lmap_id, op->lmaps,
                                                                   op->pcell, op->pncell, op->kappa, op->D, first,
                                                                   atomic_out);
        op->launches++;
        return PMG_OK;
      }
    }
    // Cache policy: a tensor that is read once per application and is larger than the Infinity Cache is streamed
    // (nt), so that it does not displace x, y and the tables; a tensor that fits stays resident between two
    // applications under the default policy (streams_past_the_cache; profiles/kernel_tuning_r03.md).  The affine mode
    // reads no tensor.
    const bool nt = P >= NT_FROM && op->geometry_mode != 1 && op->stream_policy;
    if constexpr (stream_form(P))
    {
      if (op->geometry_mode != 1)
      {
        if (nt)
          stiffness_stream_kernel<P, true><<<count, Shape<P>::WTHREADS, 0, s>>>(
              x, y, G, op->poff, op->pdofs, op->lmap_id, op->lmaps, op->pcell, op->pncell, op->kappa, op->D, first,
              atomic_out);
        else
          stiffness_stream_kernel<P, false><<<count, Shape<P>::WTHREADS, 0, s>>>(
              x, y, G, op->poff, op->pdofs, op->lmap_id, op->lmaps, op->pcell, op->pncell, op->kappa, op->D, first,
              atomic_out);
        op->launches++;
        return PMG_OK;
      }
    }
#define PMG_LAUNCH_COLUMN(AFF_, NT_)                                                                                \
  stiffness_column_kernel<P, AFF_, NT_><<<count, Shape<P>::WTHREADS, 0, s>>>(                                         \
      x, y, G, op->Gaff, op->W1, op->poff, op->pdofs, op->lmap_id, op->lmaps, op->pcell, op->pncell, op->kappa, op->D,  \
      first, atomic_out)
    if (op->geometry_mode == 1)
      PMG_LAUNCH_COLUMN(true, (P >= NT_FROM)); // (only the y stores: as before)
    else if (nt)
      PMG_LAUNCH_COLUMN(false, true);
    else
      PMG_LAUNCH_COLUMN(false, false);
#undef PMG_LAUNCH_COLUMN
  }
  op->launches++;
  return PMG_OK;
}

// one launch of the stiffness kernel over patches [first, first + count)
static int launch_patches(pmg_laplacian op, const double* x, double* y, int first, int count, int atomic_out,
                          hipStream_t s)
{
  switch (op->P)
  {
  case 1:
    return launch_stiffness<1>(op, x, y, first, count, atomic_out, s);
  case 2:
    return launch_stiffness<2>(op, x, y, first, count, atomic_out, s);
  case 3:
    return launch_stiffness<3>(op, x, y, first, count, atomic_out, s);
  case 4:
    return launch_stiffness<4>(op, x, y, first, count, atomic_out, s);
  case 5:
    return launch_stiffness<5>(op, x, y, first, count, atomic_out, s);
  case 6:
    return launch_stiffness<6>(op, x, y, first, count, atomic_out, s);
  case 7:
    return launch_stiffness<7>(op, x, y, first, count, atomic_out, s);
  case 8:
    return launch_stiffness<8>(op, x, y, first, count, atomic_out, s);
  default:
    return fail(PMG_ERR_INVALID, "Unsupported degree"); // src/laplacian.hpp:346,479
  }
}

// the interior of one application as chains: one launch per chain colour (stiffness_chain_kernel)
static int launch_chains(pmg_laplacian op, const double* x, double* y, hipStream_t s)
{
  if constexpr (chain_form(4))
  {
    if (op->P != 4)
      return fail(PMG_ERR_INVALID, "internal: chain form at degree %d", op->P);
    const bool nt = op->stream_policy;
    for (size_t c = 0; c < op->chain_count.size(); ++c)
    {
      if (op->chain_count[c] <= 0)
        continue;
      if (nt)
        stiffness_chain_kernel<4, true><<<op->chain_count[c], ChainShape<4>::THREADS, 0, s>>>(
            x, y, op->G, op->poff, op->cdofs, op->ccar, op->lmap_id, op->lmaps, op->pcell, op->pncell, op->kappa, op->D,
            op->chain_off, op->chain_patch, op->chain_first[c]);
      else
        stiffness_chain_kernel<4, false><<<op->chain_count[c], ChainShape<4>::THREADS, 0, s>>>(
            x, y, op->G, op->poff, op->cdofs, op->ccar, op->lmap_id, op->lmaps, op->pcell, op->pncell, op->kappa, op->D,
            op->chain_off, op->chain_patch, op->chain_first[c]);
      op->launches++;
    }
  }
  return PMG_OK;
}

// launches [l0, l1) of the plan, in stream order
int run_launches(pmg_laplacian op, const double* x, double* y, int l0, int l1, hipStream_t s)
{
  const bool prof = op->profiling && l1 > l0;
  if (prof)
  {
    if (op->prof_used + 2 > op->prof_events.size())
    {
      hipEvent_t a, b;
      PMG_HIP(hipEventCreate(&a));
      op->prof_events.push_back(a);
      PMG_HIP(hipEventCreate(&b));
      op->prof_events.push_back(b);
    }
    PMG_HIP(hipEventRecord(op->prof_events[op->prof_used], s));
  }
  // Two halves of the interior: the launches of the second half go to the operator's second stream, forked from and
  // joined to `s` with events (inside a stream capture these become two branches of the graph).  One ordering event
  // between the halves: patches.hip.
  const bool two = !op->launch_stream.empty() && l0 == 0 && l1 >= op->n_launch_l && l1 > l0 && op->batch_patches == 0;
  if (two)
  {
    PMG_HIP(hipEventRecord(op->ev_fork, s));
    PMG_HIP(hipStreamWaitEvent(op->stream2, op->ev_fork, 0));
  }
  int issued = 0;
  // the interior as chains: one launch per chain colour instead of the patch colours (the whole interior or nothing;
  // the stored tensor, resident)
  const bool chain = op->chain_on && l0 == 0 && l1 >= op->n_launch_l && op->n_launch_l > 0 && !two
                     && op->batch_patches == 0 && op->geometry_mode == 0;
  if (chain)
  {
    PMG_TRY(launch_chains(op, x, y, s));
    issued += (int)op->chain_count.size();
    l0 = op->n_launch_l;
  }
  auto join = [&]() -> int {
    PMG_HIP(hipEventRecord(op->ev_join, op->stream2));
    PMG_HIP(hipStreamWaitEvent(s, op->ev_join, 0));
    return PMG_OK;
  };
#ifdef PMG_ABL_OFFSET // timing only (the deferred patches run out of order): the second half's launch boundaries
                      // shifted by half a launch against the first half's
  int def_first[2] = {0, 0}, def_count[2] = {0, 0};
#endif
  for (int l = l0; l < l1; ++l)
  {
    int first = op->launch_first[l], count = op->launch_count[l];
    const int atomic_out = (l >= op->n_plain) ? 1 : 0; // merged launches add with atomics
    hipStream_t sl = s;
    if (two && l < op->n_launch_l)
    {
      if (l == op->launch_wait)
        PMG_HIP(hipStreamWaitEvent(s, op->ev_order, 0));
      sl = op->launch_stream[l] ? op->stream2 : s;
    }
    if (two && l == op->n_launch_l)
      PMG_TRY(join());
#ifdef PMG_ABL_ONE_LAUNCH // timing only (wrong sums on shared dofs): all colours of a cell list in one launch -- what do
                          // the seven launch boundaries cost?  (profiles/kernel_tuning_r03.md section 14)
    if (!atomic_out)
      while (l + 1 < l1 && l + 1 < op->n_plain && (l + 1 < op->n_launch_l) == (l < op->n_launch_l)
             && op->launch_first[l + 1] == first + count)
        count += op->launch_count[++l];
#endif
#ifdef PMG_ABL_OFFSET
    if (two && (l == 1 || l == op->launch_signal + 2) && l < op->n_launch_l)
    {
      const int k = l == 1 ? 0 : 1;
      def_first[k] = first + count / 2;
      def_count[k] = count - count / 2;
      count = count / 2;
    }
#endif
    // consecutive atomic launches over contiguous patches that the caller asked for together: one launch
    while (atomic_out && l + 1 < l1 && op->launch_first[l + 1] == first + count && !two && op->batch_patches == 0)
      count += op->launch_count[++l];
    if (count > 0)
    {
      PMG_TRY(launch_patches(op, x, y, first, count, atomic_out, sl));
      ++issued;
    }
    if (two && l == op->launch_signal)
    {
#ifdef PMG_ABL_OFFSET
      if (def_count[0] > 0)
        PMG_TRY(launch_patches(op, x, y, def_first[0], def_count[0], 0, op->stream2));
#endif
      PMG_HIP(hipEventRecord(op->ev_order, op->stream2));
    }
  }
#ifdef PMG_ABL_OFFSET
  if (two && def_count[1] > 0)
    PMG_TRY(launch_patches(op, x, y, def_first[1], def_count[1], 0, op->stream2));
#endif
  if (two && l1 == op->n_launch_l)
    PMG_TRY(join());
  PMG_HIP(hipGetLastError());
  if (prof)
  {
    PMG_HIP(hipEventRecord(op->prof_events[op->prof_used + 1], s));
    op->prof_used += 2;
    op->prof_launches += issued;
  }
  return PMG_OK;
}

template <typename T>
int upload(T** dst, const T* src, size_t n, hipStream_t s)
{
  PMG_HIP(hipMalloc(dst, sizeof(T) * (n ? n : 1)));
  if (n)
    PMG_HIP(hipMemcpyAsync(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice, s));
  return PMG_OK;
}

} // namespace

namespace pmg
{
// used by solvers.hip
int laplacian_apply(pmg_laplacian op, double* in, double* out, hipStream_t s);
int laplacian_apply_zeroed(pmg_laplacian op, double* in, double* out, hipStream_t s);
int laplacian_apply_ghosts_current(pmg_laplacian op, double* in, double* out, hipStream_t s);
const double* laplacian_diag_inv(pmg_laplacian op) { return op->diag_inv; }
pmg_layout laplacian_layout(pmg_laplacian op) { return op->layout; }
long long laplacian_launches(pmg_laplacian op) { return op->applies; }
struct LaplacianInputs
{
  int degree;
  int32_t ncells;
  const int32_t* dofmap;
  const int8_t* bc;
  const double* kappa;
};
LaplacianInputs laplacian_inputs(pmg_laplacian op) { return {op->P, op->ncells, op->dofmap, op->bc, op->kappa}; }

// what a captured graph of launches of this operator depends on; -1 = not capturable right now
long long laplacian_capture_state(pmg_laplacian op)
{
  if (op->profiling)
    return -1;
  return ((long long)op->geometry_mode << 40) ^ ((long long)op->batch_patches << 8) ^ (long long)(op->have_diag ? 1 : 0)
         ^ (long long)(op->chain_on ? 2 : 0);
}

PatchView laplacian_patches(pmg_laplacian op)
{
  PatchView v;
  v.P = op->P;
  v.K = op->K;
  v.N = op->N;
  v.npatch = op->npatch;
  v.max_m = op->max_m;
  v.pcell_h = &op->pcell_h;
  v.pncell_h = &op->pncell_h;
  v.launch_first = &op->launch_first;
  v.launch_count = &op->launch_count;
  v.n_launch_l = op->n_launch_l;
  v.merged = op->n_plain == 0;
  v.pcell = op->pcell;
  v.pncell = op->pncell;
  v.poff = op->poff;
  v.pdofs = op->pdofs;
  v.lmap_id = op->lmap_id;
  v.lmaps = op->lmaps;
  v.npdofs = op->npdofs;
  return v;
}

// Does an application zero-fill its whole output first (every first writer adds with atomics: the merged launch of a
// small level)?  Then a caller that hands over an output that is zero already can skip the fill
// (laplacian_apply_zeroed; the smoother's vector kernels clear the vector behind themselves).
static bool zero_fills_output(pmg_laplacian op)
{
  return op->needs_zero || op->launch_first.empty() || 2LL * op->n_bzero > op->layout->total();
}
bool laplacian_wants_zeroed_output(pmg_laplacian op) { return zero_fills_output(op); }

// Interior and boundary patches in ONE launch behind a whole exchange?  Only where both lists add with atomics
// anyway (n_plain == 0: the merged form of a small level), the geometry is resident, and the layout's exchange is one
// launch (halo windows).  The transfers of the level follow the operator (interpolate.hip).
bool laplacian_single_launch(pmg_laplacian op)
{
  return op->n_plain == 0 && op->batch_patches == 0 && (int)op->launch_first.size() == 2
         && op->launch_first[1] == op->launch_first[0] + op->launch_count[0] && layout_exchanges_whole(op->layout);
}

// operator()(in, out), src/laplacian.hpp:462-482 + impl_operator :373-460
static int apply_impl(pmg_laplacian op, double* in, double* out, bool out_is_zero, hipStream_t s,
                      bool exchange = true)
{
  pmg_layout l = op->layout;
  const int nl = (int)op->launch_first.size();
  // :466 -- but only where needed: dofs no patch touches, or (most of the vector) dofs whose
  // first writer adds with atomics; everything else is stored by its first writer
  const bool zero_all = zero_fills_output(op);
  if (zero_all && !out_is_zero)
    launch_zero(l->total(), out, s);
  if (op->n_bzero > 0 && !zero_all)
    zero_list_kernel<<<(op->n_bzero + 255) / 256 > 1024 ? 1024 : (op->n_bzero + 255) / 256, 256, 0, s>>>(
        op->n_bzero, op->bzero, out);
  // A small level (every launch adds with atomics: the merged form) on a window layout: the exchange whole, in one
  // launch, then ALL patches in one launch -- two launches instead of four; there is nothing an interior launch of a
  // few microseconds could hide (laplacian_single_launch).
  if (laplacian_single_launch(op))
  {
    if (exchange)
      PMG_TRY(scatter_fwd_whole(l, in, s));
    PMG_TRY(run_launches(op, in, out, 0, nl, s));
    op->applies++;
    return PMG_OK;
  }
  if (exchange)
    PMG_TRY(pmg_scatter_fwd_begin(l, in, (pmg_stream)s));        // :378
  PMG_TRY(run_launches(op, in, out, 0, op->n_launch_l, s));      // :380-413 interior cells
  if (exchange)
    PMG_TRY(pmg_scatter_fwd_end(l, in, (pmg_stream)s));          // :425
  PMG_TRY(run_launches(op, in, out, op->n_launch_l, nl, s));     // :429-455 boundary cells
  op->applies++;
  return PMG_OK;
}
int laplacian_apply(pmg_laplacian op, double* in, double* out, hipStream_t s) { return apply_impl(op, in, out, false, s); }
// The ghost entries of `in` are current already (the caller's bookkeeping, solvers.hip local_correction): the same
// application without its halo exchange.
int laplacian_apply_ghosts_current(pmg_laplacian op, double* in, double* out, hipStream_t s)
{
  return apply_impl(op, in, out, false, s, false);
}
// `out` is zero over the whole layout already (only meaningful when laplacian_wants_zeroed_output)
int laplacian_apply_zeroed(pmg_laplacian op, double* in, double* out, hipStream_t s)
{
  return apply_impl(op, in, out, zero_fills_output(op), s);
}
} // namespace pmg

extern "C" int pmg_laplacian_create_ordered(
    pmg_laplacian* out, pmg_layout layout, int degree, int32_t ncells, const double* kappa,
    const int32_t* dofmap, const double* xgeom, int32_t npoints, const int32_t* geom_dofmap,
    const double* dphi_geometry, const double* G_weights, const int32_t* lcells, int32_t n_lcells,
    const int32_t* bcells, int32_t n_bcells, const int8_t* bc_marker, int node_order, const int32_t* custom_perm1d,
    pmg_stream stream)
{
  PMG_REQUIRE(out && layout, "pmg_laplacian_create: NULL handle");
  if (degree < 1 || degree > PMG_MAX_DEGREE)
    return fail(PMG_ERR_INVALID, "Unsupported degree"); // src/laplacian.hpp:346
  PMG_REQUIRE(ncells >= 0 && n_lcells >= 0 && n_bcells >= 0 && n_lcells + n_bcells <= ncells,
              "pmg_laplacian_create: cell lists (%d + %d) exceed ncells (%d)", n_lcells, n_bcells,
              ncells);
  PMG_REQUIRE(ncells == 0 || (kappa && dofmap && xgeom && geom_dofmap && bc_marker),
              "pmg_laplacian_create: NULL array");
  PMG_REQUIRE((n_lcells == 0 || lcells) && (n_bcells == 0 || bcells),
              "pmg_laplacian_create: NULL cell list");
  PMG_REQUIRE(npoints >= 0, "pmg_laplacian_create: negative npoints");
  {
    std::vector<char> seen(ncells, 0);
    for (int set = 0; set < 2; ++set)
    {
      const int32_t* list = set ? bcells : lcells;
      const int n = set ? n_bcells : n_lcells;
      for (int i = 0; i < n; ++i)
      {
        PMG_REQUIRE(list[i] >= 0 && list[i] < ncells,
                    "pmg_laplacian_create: cell list entry %d out of range", list[i]);
        PMG_REQUIRE(!seen[list[i]], "pmg_laplacian_create: cell %d listed twice", list[i]);
        seen[list[i]] = 1;
      }
    }
  }

  hipStream_t s = S(stream);
  auto* op = new pmg_laplacian_s;
  HandleGuard<pmg_laplacian> guard(op, pmg_laplacian_destroy);
  op->layout = layout;
  op->P = degree;
  op->nd = degree + 1;
  op->N = op->nd * op->nd * op->nd;
  op->K = patch_shape(degree).K();
  {
    const int kk[9] = {0, Shape<1>::K, Shape<2>::K, Shape<3>::K, Shape<4>::K, Shape<5>::K,
                       Shape<6>::K, Shape<7>::K, Shape<8>::K};
    PMG_REQUIRE(kk[degree] == op->K, "internal: patch shape / kernel shape mismatch");
  }
  op->ncells = ncells;
  op->npoints = npoints;
  op->kappa = kappa;
  op->dofmap = dofmap;
  op->xgeom = xgeom;
  op->geom_dofmap = geom_dofmap;
  op->bc = bc_marker;
  const int nd = op->nd, N = op->N;
  const int total = layout->total();

  // cell-local node order: everything below (patch tables, diagonal, load vector, the transfers and the AMG that
  // read op->dofmap) works on an ascending dofmap; a caller in another order gets a permuted copy, made once
  std::vector<int32_t> perm1d;
  PMG_TRY(node_permutation(node_order, degree, custom_perm1d, perm1d));
  op->node_order = node_order;
  if (!is_identity(perm1d))
  {
    const std::vector<int32_t> p3 = cell_permutation(nd, perm1d);
    PMG_TRY(upload(&op->qperm, p3.data(), p3.size(), s));
    PMG_HIP(hipMalloc(&op->dofmap_own, sizeof(int32_t) * ((size_t)ncells * N ? (size_t)ncells * N : 1)));
    PMG_TRY(permute_rows_i32(ncells, N, op->qperm, dofmap, op->dofmap_own, s));
    PMG_HIP(hipStreamSynchronize(s)); // p3 goes out of scope
    op->dofmap = dofmap = op->dofmap_own;
  }

  // 1-D tables (basix's job in the reference, src/laplacian.hpp:302-317)
  std::vector<double> pts(nd), wts(nd), D(nd * nd);
  gll_table(nd, pts.data(), wts.data());
  lagrange_derivative_table(nd, pts.data(), D.data());
  PMG_TRY(upload(&op->D, D.data(), D.size(), s));

  // trilinear coordinate-element derivatives at the GLL points, [3][N][8], and
  // the 3-D weights (examples/pmg/main.cpp:216-238)
  if (dphi_geometry && G_weights)
  {
    PMG_HIP(hipMalloc(&op->dphi_geom, sizeof(double) * 24 * N));
    PMG_HIP(hipMalloc(&op->gweights, sizeof(double) * N));
    if (op->qperm) // the caller's tables are indexed by ITS quadrature-point numbers
    {
      PMG_TRY(permute_rows_f64(3, N, 8, op->qperm, dphi_geometry, op->dphi_geom, s));
      PMG_TRY(permute_rows_f64(1, N, 1, op->qperm, G_weights, op->gweights, s));
    }
    else
    {
      PMG_HIP(hipMemcpyAsync(op->dphi_geom, dphi_geometry, sizeof(double) * 24 * N,
                             hipMemcpyDeviceToDevice, s));
      PMG_HIP(hipMemcpyAsync(op->gweights, G_weights, sizeof(double) * N, hipMemcpyDeviceToDevice, s));
    }
  }
  else
  {
    std::vector<double> dphi(24 * (size_t)N), w3(N);
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nd; ++b)
        for (int c = 0; c < nd; ++c)
        {
          int q = (a * nd + b) * nd + c;
          w3[q] = wts[a] * wts[b] * wts[c];
          double ph[3][2] = {{1.0 - pts[a], pts[a]}, {1.0 - pts[b], pts[b]}, {1.0 - pts[c], pts[c]}};
          const double dp[2] = {-1.0, 1.0};
          for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
              for (int k = 0; k < 2; ++k)
              {
                int v = i * 4 + j * 2 + k;
                dphi[((size_t)0 * N + q) * 8 + v] = dp[i] * ph[1][j] * ph[2][k];
                dphi[((size_t)1 * N + q) * 8 + v] = ph[0][i] * dp[j] * ph[2][k];
                dphi[((size_t)2 * N + q) * 8 + v] = ph[0][i] * ph[1][j] * dp[k];
              }
        }
    PMG_TRY(upload(&op->dphi_geom, dphi.data(), dphi.size(), s));
    PMG_TRY(upload(&op->gweights, w3.data(), w3.size(), s));
    PMG_HIP(hipStreamSynchronize(s)); // host vectors go out of scope
  }

  // ---- patches (host): needs the dofmap, the Dirichlet marker and cell centroids ----
  PatchPlan plan;
  ChainPlan cplan;
  {
    std::vector<int32_t> h_dofmap((size_t)ncells * N), h_gd((size_t)ncells * 8);
    std::vector<int8_t> h_bc(total);
    std::vector<double> h_x((size_t)npoints * 3);
    PMG_HIP(hipMemcpyAsync(h_dofmap.data(), dofmap, sizeof(int32_t) * h_dofmap.size(),
                           hipMemcpyDeviceToHost, s));
    PMG_HIP(hipMemcpyAsync(h_gd.data(), geom_dofmap, sizeof(int32_t) * h_gd.size(),
                           hipMemcpyDeviceToHost, s));
    PMG_HIP(hipMemcpyAsync(h_bc.data(), bc_marker, sizeof(int8_t) * h_bc.size(),
                           hipMemcpyDeviceToHost, s));
    PMG_HIP(hipMemcpyAsync(h_x.data(), xgeom, sizeof(double) * h_x.size(), hipMemcpyDeviceToHost, s));
    PMG_HIP(hipStreamSynchronize(s));
    std::vector<float> centroid((size_t)ncells * 3);
    for (int32_t c = 0; c < ncells; ++c)
      for (int a = 0; a < 3; ++a)
      {
        double v = 0;
        for (int k = 0; k < 8; ++k)
        {
          int32_t g = h_gd[(size_t)c * 8 + k];
          PMG_REQUIRE(g >= 0 && g < npoints, "pmg_laplacian_create: geometry dofmap entry %d out of range", g);
          v += h_x[3 * (size_t)g + a];
        }
        centroid[(size_t)c * 3 + a] = (float)(v * 0.125);
      }
    PMG_TRY(build_patch_plan(plan, degree, ncells, h_dofmap.data(), h_bc.data(), total,
                             centroid.data(), lcells, n_lcells, bcells, n_bcells));
    // affine (parallelepiped) cells: vertex (i,j,l) = x0 + i e1 + j e2 + l e3
    op->all_affine = (n_lcells + n_bcells) > 0;
    for (int set = 0; set < 2 && op->all_affine; ++set)
    {
      const int32_t* list = set ? bcells : lcells;
      const int n = set ? n_bcells : n_lcells;
      for (int i = 0; i < n && op->all_affine; ++i)
      {
        const int32_t* gd = h_gd.data() + (size_t)list[i] * 8;
        const double* v[8];
        for (int k = 0; k < 8; ++k)
          v[k] = h_x.data() + 3 * (size_t)gd[k];
        double diam = 0, dev = 0;
        for (int a = 0; a < 3; ++a)
        {
          const double e1 = v[4][a] - v[0][a], e2 = v[2][a] - v[0][a], e3 = v[1][a] - v[0][a];
          diam += std::fabs(e1) + std::fabs(e2) + std::fabs(e3);
          dev += std::fabs(v[3][a] - (v[0][a] + e2 + e3)) + std::fabs(v[5][a] - (v[0][a] + e1 + e3))
                 + std::fabs(v[6][a] - (v[0][a] + e1 + e2)) + std::fabs(v[7][a] - (v[0][a] + e1 + e2 + e3));
        }
        if (dev > 1e-12 * diam)
          op->all_affine = false;
      }
    }
    // chain form of the interior launches (stiffness_chain_kernel).  PMG_CHAIN=0 never, =1 where a colour's chains
    // fill the GPU (one workgroup per chain and compute unit), =2 whenever chains exist (tests on small meshes).
    if (chain_form(degree) && (unsigned long long)plan.npatch * (unsigned long long)gpatch(nd, op->K) < (1ull << 32))
    {
      const char* e = std::getenv("PMG_CHAIN");
      const int mode = e ? std::atoi(e) : PMG_CHAIN_DEFAULT;
      if (mode > 0)
      {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess)
          (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        PMG_TRY(build_chain_plan(cplan, plan, total, h_bc.data(), centroid.data(), mode >= 2 ? 1 : (cus * 3) / 4));
      }
    }
    // is every local dof written by some patch?  (else out must be zero-filled first)
    std::vector<char> touched(total, 0);
    for (uint32_t v : plan.pdofs)
      touched[v & PD_MASK] = 1;
    for (int i = 0; i < total && !op->needs_zero; ++i)
      op->needs_zero = !touched[i];
  }
  op->npatch = plan.npatch;
  op->pcell_h = plan.pcell;
  op->pncell_h = plan.pncell;
  op->npdofs = (long long)plan.pdofs.size();
  op->max_m = plan.max_M;
  op->launch_first = plan.launch_first;
  op->launch_count = plan.launch_count;
  op->n_launch_l = plan.n_launch_l;
  op->n_plain = plan.n_plain;
  op->launch_stream = plan.launch_stream;
  op->launch_signal = plan.launch_signal;
  op->launch_wait = plan.launch_wait;
  if (!op->launch_stream.empty())
  {
    PMG_HIP(hipStreamCreateWithFlags(&op->stream2, hipStreamNonBlocking));
    PMG_HIP(hipEventCreateWithFlags(&op->ev_fork, hipEventDisableTiming));
    PMG_HIP(hipEventCreateWithFlags(&op->ev_order, hipEventDisableTiming));
    PMG_HIP(hipEventCreateWithFlags(&op->ev_join, hipEventDisableTiming));
  }
  PMG_TRY(upload(&op->pcell, plan.pcell.data(), plan.pcell.size(), s));
  PMG_TRY(upload(&op->pncell, plan.pncell.data(), plan.pncell.size(), s));
  op->n_bzero = (int32_t)plan.bzero.size();
  PMG_TRY(upload(&op->bzero, plan.bzero.data(), plan.bzero.size(), s));
  PMG_TRY(upload(&op->poff, plan.poff.data(), plan.poff.size(), s));
  PMG_TRY(upload(&op->pdofs, plan.pdofs.data(), plan.pdofs.size(), s));
  PMG_TRY(upload(&op->lmap_id, plan.lmap_id.data(), plan.lmap_id.size(), s));
  PMG_TRY(upload(&op->lmaps, plan.lmaps.data(), plan.lmaps.size(), s));
  if (cplan.ok)
  {
    PMG_TRY(upload(&op->cdofs, cplan.cdofs.data(), cplan.cdofs.size(), s));
    PMG_TRY(upload(&op->ccar, cplan.ccar.data(), cplan.ccar.size(), s));
    PMG_TRY(upload(&op->chain_off, cplan.chain_off.data(), cplan.chain_off.size(), s));
    PMG_TRY(upload(&op->chain_patch, cplan.chain_patch.data(), cplan.chain_patch.size(), s));
    op->chain_first = cplan.launch_first;
    op->chain_count = cplan.launch_count;
    op->chain_ok = op->chain_on = true;
  }

  const long long nslots = (long long)plan.npatch * op->K;
  const long long nq_total = nslots * N;
  PMG_TRY(upload(&op->W1, wts.data(), wts.size(), s));
  PMG_HIP(hipMalloc(&op->Gaff, sizeof(double) * 6 * (nslots ? nslots : 1)));
  if (nslots > 0)
    affine_geometry_kernel<<<(unsigned)((nslots + 255) / 256), 256, 0, s>>>(nslots, op->pcell, xgeom,
                                                                           geom_dofmap, op->Gaff);
  {
    const size_t gsize = (size_t)op->npatch * gpatch(nd, op->K);
    PMG_HIP(hipMalloc(&op->G, sizeof(double2) * (gsize ? gsize : 1)));
    PMG_HIP(hipMemsetAsync(op->G, 0, sizeof(double2) * (gsize ? gsize : 1), s)); // padding, empty slots
  }
  op->stream_policy = streams_past_the_cache((long long)sizeof(double2) * gpatch(nd, op->K) * op->npatch);
  PMG_HIP(hipMalloc(&op->diag_inv, sizeof(double) * (total ? total : 1)));
  PMG_HIP(hipEventCreate(&op->ev0));
  PMG_HIP(hipEventCreate(&op->ev1));
  if (nq_total > 0)
  {
    batch_geometry(op, 0, op->npatch, s);
    PMG_HIP(hipGetLastError());
  }
  PMG_HIP(hipStreamSynchronize(s)); // plan's host vectors are released on return
  *out = guard.release();
  return PMG_OK;
}

extern "C" int pmg_laplacian_create_with_tables(
    pmg_laplacian* out, pmg_layout layout, int degree, int32_t ncells, const double* kappa,
    const int32_t* dofmap, const double* xgeom, int32_t npoints, const int32_t* geom_dofmap,
    const double* dphi_geometry, const double* G_weights, const int32_t* lcells, int32_t n_lcells,
    const int32_t* bcells, int32_t n_bcells, const int8_t* bc_marker, pmg_stream stream)
{
  return pmg_laplacian_create_ordered(out, layout, degree, ncells, kappa, dofmap, xgeom, npoints, geom_dofmap,
                                      dphi_geometry, G_weights, lcells, n_lcells, bcells, n_bcells, bc_marker,
                                      PMG_NODES_ASCENDING, nullptr, stream);
}

extern "C" int pmg_laplacian_node_order(pmg_laplacian op) { return op ? op->node_order : -1; }

extern "C" int pmg_laplacian_create(pmg_laplacian* out, pmg_layout layout, int degree,
                                    int32_t ncells, const double* kappa, const int32_t* dofmap,
                                    const double* xgeom, int32_t npoints,
                                    const int32_t* geom_dofmap, const int32_t* lcells,
                                    int32_t n_lcells, const int32_t* bcells, int32_t n_bcells,
                                    const int8_t* bc_marker, pmg_stream stream)
{
  return pmg_laplacian_create_with_tables(out, layout, degree, ncells, kappa, dofmap, xgeom,
                                          npoints, geom_dofmap, nullptr, nullptr, lcells, n_lcells,
                                          bcells, n_bcells, bc_marker, stream);
}

extern "C" int pmg_laplacian_destroy(pmg_laplacian op)
{
  if (!op)
    return PMG_OK;
  (void)hipFree(op->G);
  (void)hipFree(op->dofmap_own);
  (void)hipFree(op->qperm);
  (void)hipFree(op->D);
  (void)hipFree(op->Gaff);
  (void)hipFree(op->W1);
  (void)hipFree(op->dphi_geom);
  (void)hipFree(op->gweights);
  (void)hipFree(op->pcell);
  (void)hipFree(op->pncell);
  (void)hipFree(op->bzero);
  (void)hipFree(op->poff);
  (void)hipFree(op->pdofs);
  (void)hipFree(op->lmap_id);
  (void)hipFree(op->lmaps);
  (void)hipFree(op->cdofs);
  (void)hipFree(op->ccar);
  (void)hipFree(op->chain_off);
  (void)hipFree(op->chain_patch);
  (void)hipFree(op->diag_inv);
  for (hipEvent_t e : op->prof_events)
    (void)hipEventDestroy(e);
  for (hipEvent_t e : {op->ev_fork, op->ev_order, op->ev_join})
    if (e)
      (void)hipEventDestroy(e);
  if (op->stream2)
    (void)hipStreamDestroy(op->stream2);
  if (op->ev0)
    (void)hipEventDestroy(op->ev0);
  if (op->ev1)
    (void)hipEventDestroy(op->ev1);
  delete op;
  return PMG_OK;
}

#ifdef PMG_STAMPS
// diagnostic builds only (tools/stamp_apply.py); not part of the ABI
extern "C" int pmg_debug_set_stamp_buffer(unsigned long long* buffer, int workgroups)
{
  PMG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buffer), &buffer, sizeof(buffer)));
  PMG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_capacity), &workgroups, sizeof(workgroups)));
  return PMG_OK;
}
#endif

extern "C" int pmg_laplacian_degree(pmg_laplacian op) { return op ? op->P : -1; }

extern "C" int pmg_laplacian_is_affine(pmg_laplacian op) { return op ? (op->all_affine ? 1 : 0) : -1; }

extern "C" int pmg_laplacian_set_geometry_mode(pmg_laplacian op, int mode)
{
  PMG_REQUIRE(op && (mode == 0 || mode == 1), "pmg_laplacian_set_geometry_mode: bad argument");
  if (mode == 1)
  {
    PMG_REQUIRE(op->all_affine, "pmg_laplacian_set_geometry_mode: the mesh has non-affine cells");
  }
  op->geometry_mode = mode;
  return PMG_OK;
}

extern "C" int pmg_laplacian_apply(pmg_laplacian op, double* in, double* out, pmg_stream stream)
{
  PMG_REQUIRE(op && in && out, "pmg_laplacian_apply: NULL argument");
  PMG_REQUIRE(in != out, "pmg_laplacian_apply: in and out alias");
  return laplacian_apply(op, in, out, S(stream));
}

extern "C" int pmg_laplacian_get_diag_inverse(pmg_laplacian op, double* diag_inv,
                                              pmg_stream stream)
{
  PMG_REQUIRE(op && diag_inv, "pmg_laplacian_get_diag_inverse: NULL argument");
  PMG_REQUIRE(op->have_diag, "pmg_laplacian_get_diag_inverse: diagonal not set");
  PMG_HIP(hipMemcpyAsync(diag_inv, op->diag_inv, sizeof(double) * op->layout->total(),
                         hipMemcpyDeviceToDevice, S(stream)));
  return PMG_OK;
}

extern "C" int pmg_laplacian_set_diag_inverse(pmg_laplacian op, const double* diag_inv,
                                              pmg_stream stream)
{
  PMG_REQUIRE(op && diag_inv, "pmg_laplacian_set_diag_inverse: NULL argument");
  PMG_HIP(hipMemcpyAsync(op->diag_inv, diag_inv, sizeof(double) * op->layout->total(),
                         hipMemcpyDeviceToDevice, S(stream)));
  op->have_diag = true;
  return PMG_OK;
}

extern "C" int pmg_laplacian_compute_diag_inverse(pmg_laplacian op, pmg_stream stream)
{
  PMG_REQUIRE(op, "pmg_laplacian_compute_diag_inverse: NULL argument");
  hipStream_t s = S(stream);
  const int total = op->layout->total();
  PMG_HIP(hipMemsetAsync(op->diag_inv, 0, sizeof(double) * total, s));
  PMG_TRY(for_each_geometry_chunk(op, s, [&](int first, int count, const double2* G) {
    const long long nslots = (long long)count * op->K, n = nslots * op->N;
    diagonal_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((long long)first * op->K, nslots, op->nd, op->K,
                                                              op->pcell, G, op->dofmap, op->bc, op->kappa, op->D,
                                                              op->diag_inv);
  }));
  if (total > 0)
    diag_invert_kernel<<<(total + 255) / 256, 256, 0, s>>>(total, op->bc, op->diag_inv);
  PMG_HIP(hipGetLastError());
  op->have_diag = true;
  return PMG_OK;
}

extern "C" int pmg_laplacian_get_geometry(pmg_laplacian op, double* G_out, pmg_stream stream)
{
  PMG_REQUIRE(op && G_out, "pmg_laplacian_get_geometry: NULL argument");
  hipStream_t s = S(stream);
  PMG_HIP(hipMemsetAsync(G_out, 0, sizeof(double) * 6 * (size_t)op->ncells * op->N, s));
  PMG_TRY(for_each_geometry_chunk(op, s, [&](int first, int count, const double2* G) {
    const long long nslots = (long long)count * op->K, n = nslots * op->N;
    geometry_export_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((long long)first * op->K, nslots, op->nd,
                                                                     op->K, op->pcell, op->qperm, G, G_out);
  }));
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_laplacian_assemble_rhs(pmg_laplacian op, const double* f, double* b,
                                          pmg_stream stream)
{
  PMG_REQUIRE(op && f && b, "pmg_laplacian_assemble_rhs: NULL argument");
  hipStream_t s = S(stream);
  PMG_HIP(hipMemsetAsync(b, 0, sizeof(double) * op->layout->total(), s));
  const long long nslots = (long long)op->npatch * op->K;
  const long long n = nslots * op->N;
  if (n > 0)
    rhs_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(nslots, op->N, op->pcell, op->xgeom,
                                                          op->geom_dofmap, op->dphi_geom,
                                                          op->gweights, op->dofmap, op->bc,
                                                          op->kappa, f, b);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}

extern "C" int pmg_laplacian_time_kernel(pmg_laplacian op, const double* in, double* out, int reps,
                                         double* ms_per_launch, pmg_stream stream)
{
  PMG_REQUIRE(op && in && out && ms_per_launch && reps > 0,
              "pmg_laplacian_time_kernel: bad argument");
  hipStream_t s = S(stream);
  const int nl = (int)op->launch_first.size();
  PMG_REQUIRE(nl > 0, "pmg_laplacian_time_kernel: operator has no cells");
  // every launch of one operator application (all colours, both cell lists),
  // exactly the kernels pmg_laplacian_apply issues, without halo or zero-fill
  PMG_HIP(hipEventRecord(op->ev0, s));
  for (int r = 0; r < reps; ++r)
    PMG_TRY(run_launches(op, in, out, 0, nl, s));
  PMG_HIP(hipEventRecord(op->ev1, s));
  PMG_HIP(hipEventSynchronize(op->ev1));
  float ms = 0.f;
  PMG_HIP(hipEventElapsedTime(&ms, op->ev0, op->ev1));
  *ms_per_launch = (double)ms / reps / pmg_laplacian_launches_per_apply(op);
  return PMG_OK;
}

// src/laplacian.hpp:383-396 (examples/mat_free/main.cpp:34-50 --batch_size)
extern "C" int pmg_laplacian_set_geometry_batch(pmg_laplacian op, long long batch_cells)
{
  PMG_REQUIRE(op && batch_cells >= 0, "pmg_laplacian_set_geometry_batch: bad argument");
  int32_t bp = 0;
  if (batch_cells > 0)
  {
    const long long want = (batch_cells + op->K - 1) / op->K;
    bp = (int32_t)std::min<long long>(std::max<long long>(want, 1), std::max(op->npatch, 1));
  }
  if (bp == op->batch_patches)
    return PMG_OK;
  PMG_HIP(hipDeviceSynchronize()); // nothing may still read the tensor
  (void)hipFree(op->G);
  op->G = nullptr;
  const size_t per_patch = (size_t)gpatch(op->nd, op->K);
  const size_t gsize = per_patch * (size_t)(bp > 0 ? bp : op->npatch);
  PMG_HIP(hipMalloc(&op->G, sizeof(double2) * (gsize ? gsize : 1)));
  PMG_HIP(hipMemset(op->G, 0, sizeof(double2) * (gsize ? gsize : 1))); // padding of the flat layout
  PMG_HIP(hipStreamSynchronize(nullptr)); // (null-stream fill: not ordered against the caller's non-blocking stream)
  op->batch_patches = bp;
  op->stream_policy = streams_past_the_cache((long long)sizeof(double2) * gsize);
  if (bp == 0 && op->npatch > 0) // back to the resident tensor
  {
    batch_geometry(op, 0, op->npatch, nullptr);
    PMG_HIP(hipGetLastError());
    PMG_HIP(hipDeviceSynchronize());
  }
  return PMG_OK;
}

extern "C" long long pmg_laplacian_geometry_bytes(pmg_laplacian op)
{
  if (!op)
    return -1;
  return (long long)sizeof(double2) * gpatch(op->nd, op->K) * (op->batch_patches > 0 ? op->batch_patches : op->npatch);
}

extern "C" int pmg_laplacian_set_profiling(pmg_laplacian op, int flag)
{
  PMG_REQUIRE(op, "pmg_laplacian_set_profiling: NULL argument");
  op->profiling = flag != 0;
  op->prof_used = 0;
  op->prof_launches = 0;
  return PMG_OK;
}

extern "C" int pmg_laplacian_read_profile(pmg_laplacian op, double* total_ms, long long* launches)
{
  PMG_REQUIRE(op && total_ms && launches, "pmg_laplacian_read_profile: NULL argument");
  double sum = 0.0;
  for (size_t i = 0; i + 1 < op->prof_used; i += 2)
  {
    PMG_HIP(hipEventSynchronize(op->prof_events[i + 1]));
    float ms = 0.f;
    PMG_HIP(hipEventElapsedTime(&ms, op->prof_events[i], op->prof_events[i + 1]));
    sum += ms;
  }
  *total_ms = sum;
  *launches = op->prof_launches;
  op->prof_used = 0;
  op->prof_launches = 0;
  return PMG_OK;
}

// number of stiffness-kernel launches one operator application issues
extern "C" int pmg_laplacian_launches_per_apply(pmg_laplacian op)
{
  if (!op)
    return -1;
  int n = 0;
  const bool chain = op->chain_on && op->batch_patches == 0 && op->geometry_mode == 0 && op->launch_stream.empty();
  for (size_t l = chain ? (size_t)op->n_launch_l : 0; l < op->launch_count.size(); ++l)
    n += op->launch_count[l] > 0;
  if (chain)
    for (int32_t c : op->chain_count)
      n += c > 0;
  return n;
}

// Chain form of the interior launches (stiffness_chain_kernel): 1 = in use, 0 = not; set: PMG_ERR_INVALID if the
// operator has no chains (small or merged level, no tensor grid of patches, another degree).
extern "C" int pmg_laplacian_chain_form(pmg_laplacian op) { return !op ? -1 : op->chain_on ? 1 : 0; }
extern "C" int pmg_laplacian_chain_available(pmg_laplacian op) { return !op ? -1 : op->chain_ok ? 1 : 0; }
extern "C" int pmg_laplacian_set_chain_form(pmg_laplacian op, int on)
{
  PMG_REQUIRE(op, "pmg_laplacian_set_chain_form: NULL argument");
  if (on && !op->chain_ok)
    return fail(PMG_ERR_INVALID, "pmg_laplacian_set_chain_form: the operator has no chains of patches");
  op->chain_on = on != 0;
  return PMG_OK;
}

// 2 if the interior launches of an application run as two halves on two streams, else 1
extern "C" int pmg_laplacian_apply_streams(pmg_laplacian op)
{
  return !op ? -1 : (!op->launch_stream.empty() && op->batch_patches == 0) ? 2 : 1;
}

