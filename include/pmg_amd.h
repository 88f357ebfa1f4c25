/* pmg_amd.h -- C ABI of the MI355X-native matrix-free p-multigrid hot path.
 *
 * One shared library (pmg-dolfinx_amd/lib/libpmg_amd.so, built for gfx950 by
 * __graft_entry__.build()).  Plain pointers and sizes only; every device
 * pointer is caller-owned unless stated otherwise and must stay valid for the
 * lifetime of the handle that was given it (the reference's objects hold
 * non-owning std::span's of device memory in exactly the same way,
 * src/laplacian.hpp:500-509).  All functions return 0 on success and a
 * negative code on failure; pmg_last_error() returns the message of the last
 * failure on the calling thread (the reference throws std::runtime_error for
 * the same conditions -- src/laplacian.hpp:346,479, src/vector.hpp:343,
 * src/cg.hpp:125,138 -- and print+exit(1)s on HIP errors, src/util.hpp:10-18).
 * Nothing here is thread safe; one host thread per GPU, like the reference.
 *
 * The reference's interface for this path is header-only C++ duck typing over
 * dolfinx types; each entry point below names the reference
 * member it replaces.  dolfinx's IndexMap/Scatterer inputs are flattened to
 * arrays.  INTEGRATION.md shows the thin C++ adapter a maintainer would add on
 * the reference side.
 *
 * All citations are relative to Wells-Group/pmg-dolfinx @ 2024_08_07.
 */
#ifndef PMG_AMD_H
#define PMG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMG_OK 0
#define PMG_ERR_INVALID -1   /* bad argument / unsupported degree / size mismatch */
#define PMG_ERR_HIP -2       /* a HIP runtime call failed */
#define PMG_ERR_NUMERIC -3   /* e.g. TQLI did not converge */
#define PMG_MAX_DEGREE 8

typedef void* pmg_stream; /* hipStream_t (0 = default stream) */

typedef struct pmg_layout_s* pmg_layout;
typedef struct pmg_comm_s* pmg_comm;
typedef struct pmg_laplacian_s* pmg_laplacian;
typedef struct pmg_chebyshev_s* pmg_chebyshev;
typedef struct pmg_cg_s* pmg_cg;
typedef struct pmg_interpolator_s* pmg_interpolator;
typedef struct pmg_multigrid_s* pmg_multigrid;
typedef struct pmg_amg_s* pmg_amg;

const char* pmg_last_error(void);
int pmg_version(void);

/* ---- host-side tables (no GPU needed) ------------------------------------
 * What the reference gets from basix at construction time.
 * pmg_gll_table: n-point Gauss-Lobatto-Legendre rule on [0,1]
 *   (basix make_quadrature(gll, interval), src/laplacian.hpp:307-309).
 * pmg_lagrange_derivative_table: D[q*n+i] = l_i'(x_q) on the GLL nodes
 *   (second half of element1D.tabulate(1,...), src/laplacian.hpp:312-317).
 * pmg_interpolation_table: M[j*(pc+1)+k] = l^coarse_k(x^fine_j)
 *   (1-D factor of basix::compute_interpolation_operator, src/interpolate.hpp:118).
 * pmg_tqli: eigenvalues of a symmetric tridiagonal matrix, in place on d
 *   (tqli(), src/cg.hpp:55-84). */
int pmg_gll_table(int n, double* points, double* weights);
int pmg_lagrange_derivative_table(int n, double* D);
int pmg_interpolation_table(int p_coarse, int p_fine, double* M);
int pmg_tqli(double* d, double* e, int n);

/* ---- cell-local node order --------------------------------------------------
 * The reference takes its dofmaps from a basix tensor-product element
 * (basix::create_tp_element, examples/pmg/main.cpp:83-87) and its 1-D tables and quadrature
 * points from basix as well (src/laplacian.hpp:302-317, src/interpolate.hpp:118): the
 * cell-local index is t = ja*nd^2 + jb*nd + jc in both, but the 1-D index j runs over the
 * nodes in BASIX order -- vertex 0, vertex 1, then the interior nodes from left to right
 * ("endpoints first"; that the dofs and the GLL points share this order is why "phi is the
 * identity" in src/laplacian.hpp:200-202).  The kernels of this library index nodes by
 * ASCENDING coordinate.  Every entry point that receives or returns an array indexed by a
 * cell-local node or quadrature-point number therefore has an `_ordered` form that takes
 * the caller's order; the library applies it ONCE, at construction, where the caller's
 * arrays are turned into its own (a permuted copy of the dofmap: 4 N bytes per cell, held
 * by the handle), so the kernels and everything behind them are unaffected.
 *
 *   PMG_NODES_ASCENDING       j = position by coordinate (the plain entry points)
 *   PMG_NODES_ENDPOINTS_FIRST basix: j = 0 -> x = 0, j = 1 -> x = 1, j >= 2 -> interior node j - 1
 *   PMG_NODES_CUSTOM          perm1d[j] = ascending position of the caller's 1-D node j, j < degree + 1
 *
 * Arrays affected: dofmap (operator, interpolator), dphi_geometry [3][nq][8] and G_weights [nq]
 * when the caller supplies them, the geometry tensor returned by pmg_laplacian_get_geometry
 * ([ncells][nq][6], q in the operator's node order).  NOT affected: vectors and everything indexed by
 * a (global or local) dof number -- bc_marker, f and b of pmg_laplacian_assemble_rhs, diag_inv --,
 * the vertex order of geom_dofmap (two nodes per direction: both orders coincide), cell lists.
 * pmg_node_permutation writes perm1d[degree + 1] for any of the three orders (custom: validated copy).
 * The *_ordered table functions return what basix would: points / weights / D rows and columns /
 * M rows (fine) and columns (coarse) in the given order. */
#define PMG_NODES_ASCENDING 0
#define PMG_NODES_ENDPOINTS_FIRST 1
#define PMG_NODES_CUSTOM 2
int pmg_node_permutation(int node_order, int degree, const int32_t* custom_perm1d, int32_t* perm1d);
int pmg_gll_table_ordered(int n, int node_order, const int32_t* custom_perm1d, double* points, double* weights);
int pmg_lagrange_derivative_table_ordered(int n, int node_order, const int32_t* custom_perm1d, double* D);
int pmg_interpolation_table_ordered(int p_coarse, int p_fine, int node_order, const int32_t* custom_coarse,
                                    const int32_t* custom_fine, double* M);

/* ---- distributed vector layout -------------------------------------------
 * Replaces the data members of acc::Vector (src/vector.hpp:83-96,304-324): a
 * vector is a caller-owned device array of size_local + num_ghosts doubles
 * (block size 1), described by a layout.  send_indices (owned positions packed
 * for the neighbours, == Scatterer::local_indices()) and recv_indices (ghost
 * positions, relative to size_local, == Scatterer::remote_indices()) are device
 * arrays; send_buffer / recv_buffer are caller-owned device staging buffers of
 * n_send / n_recv doubles.
 *
 * The exchange itself is the caller's (the reference delegates it to
 * dolfinx::common::Scatterer over MPI, src/vector.hpp:203-206,215): `exchange`
 * is called with phase 0 after the pack kernel has been enqueued on `stream`
 * (start moving send_buffer -> the neighbours' recv_buffer, asynchronously with
 * respect to `stream`), and with phase 1 before the unpack kernel is enqueued
 * (make `stream` wait for the arrival).  Phases 2 / 3 are the same for the
 * reverse scatter (recv_buffer -> the owners' send_buffer).  It must return 0.  If `exchange`
 * is non-NULL it is called on every scatter, also when n_send == n_recv == 0 (the
 * caller's exchange is typically a collective); pass NULL on a single rank.
 *
 * `allreduce_sum` sums `n` host doubles over all ranks in place (the
 * reference's MPI_Allreduce, src/vector.hpp:350); NULL on a single rank. */
typedef int (*pmg_exchange_fn)(void* user, int phase, pmg_stream stream);
typedef int (*pmg_allreduce_fn)(void* user, double* values, int n);

int pmg_layout_create(pmg_layout* out, int32_t size_local, int32_t num_ghosts, int32_t n_send,
                      const int32_t* send_indices, double* send_buffer, int32_t n_recv,
                      const int32_t* recv_indices, double* recv_buffer, pmg_exchange_fn exchange,
                      pmg_allreduce_fn allreduce_sum, void* user);
int pmg_layout_destroy(pmg_layout l);
int32_t pmg_layout_size_local(pmg_layout l);
int32_t pmg_layout_num_ghosts(pmg_layout l);
/* Optional: the maximum over all ranks of n host doubles, in place (the reference's
 * MPI_Allreduce(MPI_MAX) of norm(linf), src/vector.hpp:383-385); `user` is the pointer given
 * to pmg_layout_create.  Without it pmg_vec_norm(linf) fails on a layout that has an
 * allreduce_sum callback (several ranks), because a sum cannot stand in for a maximum. */
int pmg_layout_set_allreduce_max(pmg_layout l, pmg_allreduce_fn allreduce_max);

/* ---- native communicator: RCCL over xGMI, issued by the library -------------
 * The alternative to the two callbacks above for one-process-per-GPU runs on one node: the
 * library itself posts the neighbour exchange (one group of ncclSend/ncclRecv per scatter, on
 * the communicator's own stream, ordered against the compute stream with events -- no host
 * synchronisation, no callback) and sums the scalars of the reductions with ncclAllReduce on
 * device memory.  This is the role of dolfinx's Scatterer + MPI in the reference
 * (src/vector.hpp:94,203-206,215,350).  RCCL is bound at run time (librccl.so.1).
 *
 *   pmg_comm_unique_id : rank 0 obtains the 128-byte id (ncclGetUniqueId) and hands it to the
 *                        other ranks by whatever means the caller has (a file, MPI_Bcast,
 *                        torch.distributed.broadcast ...);
 *   pmg_comm_create    : collective over all ranks (ncclCommInitRank) on the current device;
 *   pmg_layout_set_comm: attaches the communicator and the neighbour list of the halo plan --
 *                        neighbour i receives send_buffer[sum(send_counts[:i]) ...) and fills
 *                        recv_buffer[sum(recv_counts[:i]) ...), i.e. the index lists given to
 *                        pmg_layout_create are grouped by neighbour in this order.  Every rank
 *                        of the communicator must call every scatter / reduction of a layout
 *                        in the same order (they are collectives), also a rank with no
 *                        neighbours.  The communicator must outlive the layout. */
#define PMG_COMM_ID_BYTES 128
int pmg_comm_unique_id(char* id /* [PMG_COMM_ID_BYTES] */);
int pmg_comm_create(pmg_comm* out, int rank, int nranks, const char* id);
int pmg_comm_destroy(pmg_comm comm);
/* 1 if a halo exchange captured into a hipGraph keeps its overlap with the interior cells (the communicator's stream is
 * forked into the capture), 0 if this process's HIP runtime cannot end such a capture (7.0.x: unbounded recursion in
 * hipStreamEndCapture) and the captured exchange is issued on the capturing stream instead. */
int pmg_comm_capture_overlaps(void);
/* Set-up helper, blocking (where the reference's set-up calls MPI_Allgather): `bytes` bytes of host memory from every
 * rank, in rank order, into recv[size * bytes] on every rank. */
int pmg_comm_allgather(pmg_comm comm, const void* send, size_t bytes, void* recv);
/* values[0 .. n) (device memory) summed over the ranks in place, ordered on `stream` (MPI_Allreduce(MPI_IN_PLACE, SUM)). */
int pmg_comm_allreduce_sum(pmg_comm comm, double* values, int n, pmg_stream stream);
int pmg_comm_rank(pmg_comm comm);
int pmg_comm_size(pmg_comm comm);
int pmg_layout_set_comm(pmg_layout l, pmg_comm comm, int32_t n_neighbors, const int32_t* neighbor_ranks,
                        const int32_t* send_counts, const int32_t* recv_counts);

/* ---- halo windows: the neighbour exchange as direct stores into the neighbours' memory ----
 * A second route for the exchange of src/vector.hpp:186-238 between the GPUs of one node, next to the grouped
 * ncclSend / ncclRecv above: every rank owns a *window* of device memory which its neighbours map (hipIpc) and store
 * their packed values into; arrival and consumption are signalled by counters in a small flag block, double-buffered.
 * An exchange is two kernels on the compute stream (gather + store + signal; wait + copy + acknowledge): no second
 * stream, no host work beyond the two launches, capturable into a hipGraph on any runtime.  The reductions still use
 * the layout's communicator or callbacks.
 *
 *   pmg_window_alloc / _open / _close / _free : window memory (zeroed, fine-grained where the runtime offers it) and
 *       its 64-byte interprocess handle; the caller passes the handles between the ranks by whatever means it has
 *       (a file, MPI, torch.distributed.all_gather_object).  A rank that is its own neighbour passes its own pointers.
 *   pmg_layout_window_describe : for a plan (per-neighbour counts, in the order of the index lists given to
 *       pmg_layout_create) the size of the window in doubles and, per neighbour k, where in MY window k's values land
 *       (fwd_offsets: owner -> ghost; rev_offsets: ghost -> owner).  The flag block has PMG_WINDOW_FLAG_WORDS uint64.
 *   pmg_layout_set_windows : attaches my window and flag block and, per neighbour k, its mapped window and flag
 *       block, the size of its window (what IT got from describe), the offsets IT got from describe for ME, and my
 *       position in ITS neighbour list (nb_slot).  All memory must outlive the layout.
 * Every rank must call every scatter of a layout in the same order.  A wait that exceeds PMG_WINDOW_TIMEOUT_MS
 * (default 5000) ends the kernel and fails the next scatter of the layout with PMG_ERR_HIP. */
#define PMG_WINDOW_HANDLE_BYTES 64
#define PMG_WINDOW_MAX_NEIGHBORS 64
#define PMG_WINDOW_FLAG_WORDS (4 * PMG_WINDOW_MAX_NEIGHBORS + 8)
#define PMG_WINDOW_ERR_NO_ARRIVAL 1
#define PMG_WINDOW_ERR_SLOT_BUSY 2
int pmg_window_alloc(size_t bytes, void** ptr, char* handle /* [PMG_WINDOW_HANDLE_BYTES] */);
/* 1: the last pmg_window_alloc of this process obtained fine-grained device memory, 0: ordinary, -1: none yet */
int pmg_window_fine_grained(void);
int pmg_window_open(const char* handle, void** ptr);
int pmg_window_close(void* ptr);
int pmg_window_free(void* ptr);
/* A communicator made of windows: the reductions and the set-up gathers of the ranks of one node without a transport
 * library (no RCCL, no MPI): every rank allocates one window of PMG_COMM_WINDOW_BYTES with pmg_window_alloc, hands its
 * handle to all ranks, maps theirs (pmg_window_open; windows[rank] is its own pointer) and creates the communicator.
 * An all-reduce of up to PMG_COMM_WINDOW_CHUNK doubles is two kernels on the caller's stream (stores into every rank's window + flags; wait + combine in rank
 * order, so every rank gets the same bits), replays from a hipGraph, and costs no host work beyond the two launches.
 * Layouts on such a communicator need halo windows for their exchange (pmg_layout_set_windows).  Every rank must issue
 * the reductions of the communicator in the same order, one stream at a time.  The windows must outlive it. */
#define PMG_COMM_WINDOW_MAX_RANKS 16
#define PMG_COMM_WINDOW_CHUNK 16384 /* doubles per rank and exchange; longer vectors go chunk by chunk */
#define PMG_COMM_WINDOW_BYTES \
  (8 * (2 * PMG_COMM_WINDOW_MAX_RANKS + 8 + 2 * PMG_COMM_WINDOW_MAX_RANKS * PMG_COMM_WINDOW_CHUNK))
int pmg_comm_create_windows(pmg_comm* out, int rank, int nranks, void* const* windows /* [nranks] */);
int pmg_layout_window_describe(int32_t n_neighbors, const int32_t* send_counts, const int32_t* recv_counts,
                               int64_t* window_doubles, int64_t* fwd_offsets, int64_t* rev_offsets);
int pmg_layout_set_windows(pmg_layout l, int32_t n_neighbors, const int32_t* send_counts, const int32_t* recv_counts,
                           double* window, uint64_t* flags, double* const* nb_window, uint64_t* const* nb_flags,
                           const int64_t* nb_window_doubles, const int64_t* nb_fwd_offset, const int64_t* nb_rev_offset,
                           const int32_t* nb_slot);

/* Vector::scatter_fwd_begin / scatter_fwd_end (src/vector.hpp:186-238): owner ->
 * ghost update of x; pack/unpack run on `stream` without host synchronisation. */
int pmg_scatter_fwd_begin(pmg_layout l, const double* x, pmg_stream stream);
/* Forward scatters issued on the layout since its creation (one per operator application, prolongation or
 * restriction that needs its input's ghosts; the V-cycle skips those whose ghosts are current already, see
 * pmg_multigrid_apply).  For tests and for pricing a cycle's exchanges. */
long long pmg_layout_forward_scatters(pmg_layout l);
int pmg_scatter_fwd_end(pmg_layout l, double* x, pmg_stream stream);
/* Vector::scatter_rev_begin / _end (src/vector.hpp:249-286): ghost -> owner,
 * accumulated into the owned entries.  Not used on the hot path. */
int pmg_scatter_rev_begin(pmg_layout l, const double* x, pmg_stream stream);
int pmg_scatter_rev_end(pmg_layout l, double* x, pmg_stream stream);

/* ---- BLAS-1 (free functions of src/vector.hpp:333-454) ---------------------
 * Ranges follow the reference: set and scale touch owned+ghost entries
 * (:109-115, :413-418); axpy, copy, pointwise_mult and the reductions touch the
 * owned entries only (:398-407, :424-431, :438-447, :334-352). */
int pmg_vec_set(pmg_layout l, double* x, double value, pmg_stream stream);
int pmg_vec_scale(pmg_layout l, double* x, double alpha, pmg_stream stream);
int pmg_vec_copy(pmg_layout l, double* dst, const double* src, pmg_stream stream);
/* r = alpha * x + y */
int pmg_vec_axpy(pmg_layout l, double* r, double alpha, const double* x, const double* y,
                 pmg_stream stream);
/* w = x .* y */
int pmg_vec_pointwise_mult(pmg_layout l, double* w, const double* x, const double* y,
                           pmg_stream stream);
/* Synchronous reductions (block until the value is on the host, then allreduce). */
int pmg_vec_inner_product(pmg_layout l, const double* a, const double* b, double* result,
                          pmg_stream stream);
int pmg_vec_squared_norm(pmg_layout l, const double* a, double* result, pmg_stream stream);
/* norm_type 0 = l2, 1 = linf (max |a_i|; the reference's abs-after-max slip,
 * src/vector.hpp:381-382, is not reproduced). */
int pmg_vec_norm(pmg_layout l, const double* a, int norm_type, double* result, pmg_stream stream);

/* ---- matrix-free Laplacian (acc::MatFreeLaplacian, src/laplacian.hpp:284-526)
 * Arguments mirror the reference constructor (:289-297); all arrays are device
 * pointers except the two cell lists, which are host arrays like the
 * reference's std::vector<int>:
 *   kappa          [ncells]               DG-0 coefficient per cell
 *   dofmap         [ncells * (degree+1)^3] local dof of (cell, t), t = a*nd^2+b*nd+c with a, b, c the
 *                  node numbers along x, y, z by ASCENDING coordinate (a basix / dolfinx dofmap numbers
 *                  the 1-D nodes endpoints first: use pmg_laplacian_create_ordered for it)
 *   xgeom          [3 * npoints]          vertex coordinates
 *   geom_dofmap    [8 * ncells]           cell vertices, tensor-product order k = i*4+j*2+l
 *   bc_marker      [size_local+num_ghosts] 1 on Dirichlet dofs
 *   lcells/bcells  cells that touch no ghost dof / cells that do (+ ghost cells),
 *                  src/mesh.hpp:105-143
 * The reference also takes dphi_geometry and G_weights from the caller
 * (basix tabulations of the trilinear coordinate element and the 3-D GLL
 * weights); they are fully determined by `degree`, so the library builds them
 * itself -- pass them to pmg_laplacian_create_with_tables to override.
 * The constructor precomputes the geometry tensor G for every listed cell
 * (geometry_computation, :22-113, with the determinant expanded correctly and G
 * tied to the cell, not to its position in the launched list -- SURVEY.md quirks
 * Q1, Q2) and groups the cells into coloured patches (its own copy of the
 * dofmap in patch form); it reads dofmap, bc_marker, xgeom and geom_dofmap back
 * to the host once for that.  Degrees 1..PMG_MAX_DEGREE are supported (the
 * reference stops at 5, :335-346). */
int pmg_laplacian_create(pmg_laplacian* out, pmg_layout layout, int degree, int32_t ncells,
                         const double* kappa, const int32_t* dofmap, const double* xgeom,
                         int32_t npoints, const int32_t* geom_dofmap, const int32_t* lcells,
                         int32_t n_lcells, const int32_t* bcells, int32_t n_bcells,
                         const int8_t* bc_marker, pmg_stream stream);
int pmg_laplacian_create_with_tables(pmg_laplacian* out, pmg_layout layout, int degree,
                                     int32_t ncells, const double* kappa, const int32_t* dofmap,
                                     const double* xgeom, int32_t npoints,
                                     const int32_t* geom_dofmap, const double* dphi_geometry,
                                     const double* G_weights, const int32_t* lcells,
                                     int32_t n_lcells, const int32_t* bcells, int32_t n_bcells,
                                     const int8_t* bc_marker, pmg_stream stream);
/* The same for a caller whose cell-local node numbering is not ascending (see "cell-local node order"
 * above): `dofmap`, and dphi_geometry / G_weights when given (both may be NULL), are indexed in
 * `node_order`; custom_perm1d [degree + 1] only for PMG_NODES_CUSTOM.  A dolfinx / basix caller passes
 * PMG_NODES_ENDPOINTS_FIRST with the arrays exactly as the reference's constructor receives them
 * (src/laplacian.hpp:289-297).  The operator keeps its own ascending copy of the dofmap. */
int pmg_laplacian_create_ordered(pmg_laplacian* out, pmg_layout layout, int degree, int32_t ncells,
                                 const double* kappa, const int32_t* dofmap, const double* xgeom, int32_t npoints,
                                 const int32_t* geom_dofmap, const double* dphi_geometry, const double* G_weights,
                                 const int32_t* lcells, int32_t n_lcells, const int32_t* bcells, int32_t n_bcells,
                                 const int8_t* bc_marker, int node_order, const int32_t* custom_perm1d,
                                 pmg_stream stream);
int pmg_laplacian_node_order(pmg_laplacian op); /* PMG_NODES_* the operator was created with */
int pmg_laplacian_destroy(pmg_laplacian op);
/* operator()(in, out), :462-482: zeroes `out` (ghosts included), updates the
 * ghosts of `in` (side effect, as in the reference), interior cells overlap
 * the halo exchange, then the boundary cells. */
int pmg_laplacian_apply(pmg_laplacian op, double* in, double* out, pmg_stream stream);
/* get_diag_inverse / set_diag_inverse, :484-495 (owned+ghost entries). */
int pmg_laplacian_get_diag_inverse(pmg_laplacian op, double* diag_inv, pmg_stream stream);
int pmg_laplacian_set_diag_inverse(pmg_laplacian op, const double* diag_inv, pmg_stream stream);
/* Matrix-free inverse diagonal of the BC-treated operator, stored in the
 * operator: replaces "assemble a CSR to read its diagonal"
 * (examples/pmg/main.cpp:274-279, src/csr.hpp:100-110).  BC rows give 1. */
int pmg_laplacian_compute_diag_inverse(pmg_laplacian op, pmg_stream stream);
/* The precomputed geometry tensor in the reference's layout [ncells][nq][6]
 * (:99-111), for parity tests.  `G_out` is a device array; q = ja*nd^2 + jb*nd + jc in the node order
 * the operator was created with (ascending unless pmg_laplacian_create_ordered said otherwise). */
int pmg_laplacian_get_geometry(pmg_laplacian op, double* G_out, pmg_stream stream);
/* GLL-collocated load vector b_i = sum_cells kappa * w_q * detJ_q * f_i at q = i
 * (what dolfinx assemble_vector does for L = inner(f, v)*dx with the GLL rule,
 * examples/pmg/poisson.py:40; examples/pmg/main.cpp:289-295), then set_bc:
 * b[bc] = 0.  `f` holds the nodal values of the source term. */
int pmg_laplacian_assemble_rhs(pmg_laplacian op, const double* f, double* b, pmg_stream stream);
int pmg_laplacian_degree(pmg_laplacian op);
/* Geometry mode of the apply.  0 (default): the reference's data structure, the
 * stored tensor G[cell][q][6] is streamed (48 bytes per quadrature point).
 * 1: affine cells -- when every cell is a parallelepiped, G_q = w_q * Gc with one
 * constant tensor per cell, so the kernel reads 48 bytes per CELL instead (same
 * result to rounding; not in the reference; SURVEY.md 8d calls this byte model
 * separate from storedG).  Fails if the mesh has a non-affine cell. */
int pmg_laplacian_is_affine(pmg_laplacian op);
int pmg_laplacian_set_geometry_mode(pmg_laplacian op, int mode);
/* Geometry batching (src/laplacian.hpp:383-396; examples/mat_free/main.cpp:34-50 --batch_size):
 * with batch_cells > 0 the tensor G is not kept; every application recomputes it for about
 * batch_cells cells at a time (rounded up to whole patches) into a buffer of that size, right
 * before the cells' stiffness launch -- the reference's memory / time trade, bit-identical
 * results.  0 (default) keeps G resident: 48 bytes per quadrature point, 1.57 GB for 64^3 cells of degree 4 (the
 * card has 288 GB).  pmg_laplacian_geometry_bytes
 * returns the size of the tensor buffer currently held. */
int pmg_laplacian_set_geometry_batch(pmg_laplacian op, long long batch_cells);
long long pmg_laplacian_geometry_bytes(pmg_laplacian op);
/* One operator application issues one stiffness-kernel launch per patch colour of the
 * interior cell list (8 on a structured box) plus one for the boundary list; on a small
 * level -- fewer patch dofs in the interior list than 2 M (degree 1) / 6 M (degree >= 2) --
 * the colours are merged into one launch that accumulates with atomics.
 * pmg_set_merge_threshold overrides that limit for operators created afterwards (0: always
 * coloured launches, a huge value: always merged, negative: the defaults above); process-wide,
 * for tests and tuning.
 *
 * Opt-in (PMG_APPLY_STREAMS=1 in the environment when the operator is created; =2 also on small levels, for tests):
 * the interior of a large level (>= 2048 patches, colours launched one by one) is cut in two halves whose colour
 * sequences run on two streams -- the caller's and one the operator owns, forked and joined with events; inside a stream
 * capture they become two branches of the graph -- so that each half fills the other's launch tails.  The halves share
 * dofs only across the cut; one event between the sequences orders those (patches.hip).  Off by default: the three
 * events per application cost more than the tails they hide (profiles/kernel_tuning_r04.md).
 * pmg_laplacian_apply_streams returns 1 or 2. */
int pmg_laplacian_launches_per_apply(pmg_laplacian op);
int pmg_laplacian_apply_streams(pmg_laplacian op);
/* Chain form of the interior launches (degree 4; csrc/patches.hpp ChainPlan, stiffness_chain_kernel): the interior
 * patches of a large level strung together along the axis with the fewest patch positions, one persistent workgroup
 * of sixteen wavefronts per chain; the gather of the next patch and the write-back of the previous one run under the
 * cell loop of the patch in hand, dofs shared by consecutive patches stay in LDS, a chain colour is one launch (four
 * on a box instead of eight patch colours).  Same sums as the patch launches to rounding.  Built when the operator is
 * created if PMG_CHAIN=1 (where every colour's chains fill the GPU) or =2 (whenever the patches form a tensor grid;
 * tests); PMG_CHAIN=0 never.  _chain_available: 1 if the operator has chains; _chain_form: 1 if they are in use;
 * _set_chain_form switches (PMG_ERR_INVALID when there are none).  The boundary cell list, merged small levels,
 * the affine mode and batched geometry keep the patch launches. */
int pmg_laplacian_chain_available(pmg_laplacian op);
int pmg_laplacian_chain_form(pmg_laplacian op);
int pmg_laplacian_set_chain_form(pmg_laplacian op, int on);
int pmg_set_merge_threshold(long long patch_dofs);
/* Dominant-kernel timing hook for bench.py: enqueue `reps` times every
 * stiffness-kernel launch of one operator application (no halo, no zero-fill)
 * bracketed by HIP events on `stream`; returns the mean milliseconds per launch
 * (= time per application / pmg_laplacian_launches_per_apply). */
int pmg_laplacian_time_kernel(pmg_laplacian op, const double* in, double* out, int reps,
                              double* ms_per_launch, pmg_stream stream);

/* In-situ timing of the same kernels where they run (inside a smoother, a V-cycle, a CG
 * iteration): while profiling is on, every run of stiffness-kernel launches an operator
 * application issues is bracketed by a pair of HIP events on the launch stream (a few
 * microseconds of host time per application -- leave it off in timed loops).
 * pmg_laplacian_read_profile waits for the recorded events, returns the summed milliseconds
 * and the number of stiffness launches they cover, and resets the record. */
int pmg_laplacian_set_profiling(pmg_laplacian op, int flag);
int pmg_laplacian_read_profile(pmg_laplacian op, double* total_ms, long long* launches);

/* ---- Chebyshev smoother (acc::Chebyshev, src/chebyshev.hpp:19-106) -------- */
int pmg_chebyshev_create(pmg_chebyshev* out, pmg_layout layout, double eig_min, double eig_max);
int pmg_chebyshev_destroy(pmg_chebyshev s);
int pmg_chebyshev_set_max_iterations(pmg_chebyshev s, int max_iter);
/* solve(A, x, b, verbose), :46-91.  x is in/out. */
int pmg_chebyshev_solve(pmg_chebyshev s, pmg_laplacian A, double* x, const double* b,
                        pmg_stream stream);

/* ---- Jacobi-PCG + Lanczos eigenvalue estimate (acc::CGSolver, src/cg.hpp:93-250) */
int pmg_cg_create(pmg_cg* out, pmg_layout layout);
int pmg_cg_destroy(pmg_cg s);
int pmg_cg_set_max_iterations(pmg_cg s, int max_iter);
int pmg_cg_set_tolerance(pmg_cg s, double rtol);
int pmg_cg_store_coefficients(pmg_cg s, int flag);
/* solve(A, x, b), :147-222; *iterations receives the iteration count.  If
 * `precond` is non-NULL the V-cycle (zero initial guess) replaces the hard-wired
 * Jacobi preconditioner (not in the reference). */
int pmg_cg_solve(pmg_cg s, pmg_laplacian A, double* x, const double* b, pmg_multigrid precond,
                 int* iterations, pmg_stream stream);
/* Flexible variant (not in the reference): with a V-cycle preconditioner that is not a fixed
 * linear operator -- a Krylov coarse solver inside -- beta = r_new.(z_new - z_old) / r_old.z_old
 * keeps the recurrence residual honest.  No effect without `precond`. */
int pmg_cg_set_flexible(pmg_cg s, int flag);
/* alphas()/betas(), :118-119; returns the number stored. */
int pmg_cg_coefficients(pmg_cg s, double* alphas, double* betas, int capacity);
/* compute_eigenvalues(), :121-142: sorted ascending; returns the count or <0. */
int pmg_cg_compute_eigenvalues(pmg_cg s, double* eigs, int capacity);
int pmg_cg_residual(pmg_cg s, double* rnorm);

/* ---- p-transfer (Interpolator<T>, src/interpolate.hpp:93-329) -------------
 * Coarse (Q1) and fine (Q2) spaces on the same cells; dofmaps as for the
 * operator; lcells/bcells as there. */
int pmg_interpolator_create(pmg_interpolator* out, pmg_layout layout_coarse,
                            pmg_layout layout_fine, int degree_coarse, int degree_fine,
                            int32_t ncells, const int32_t* dofmap_coarse,
                            const int32_t* dofmap_fine, const int32_t* lcells, int32_t n_lcells,
                            const int32_t* bcells, int32_t n_bcells, pmg_stream stream);
/* Same, sharing the cell patches (grouping, colours, launch order) of the
 * fine-level operator: both transfers then run one workgroup per patch with LDS
 * accumulation -- no global atomics, no zero-fill -- and the prolongation can be
 * fused with the correction.  fine_operator may be NULL (== pmg_interpolator_create);
 * otherwise it must outlive the interpolator, which reads its patch tables. */
int pmg_interpolator_create_with_operator(pmg_interpolator* out, pmg_layout layout_coarse,
                                          pmg_layout layout_fine, int degree_coarse,
                                          int degree_fine, int32_t ncells,
                                          const int32_t* dofmap_coarse, const int32_t* dofmap_fine,
                                          const int32_t* lcells, int32_t n_lcells,
                                          const int32_t* bcells, int32_t n_bcells,
                                          pmg_laplacian fine_operator, pmg_stream stream);
/* The same with both dofmaps in the caller's cell-local node order (see "cell-local node order"; the
 * reference's Interpolator receives the dofmaps of two basix tensor-product spaces, src/interpolate.hpp:104-107,
 * and its operator from basix::compute_interpolation_operator in that order, :118).  custom_* [degree + 1] only for
 * PMG_NODES_CUSTOM.  fine_operator may be NULL; if given, it may have been created in any node order. */
int pmg_interpolator_create_ordered(pmg_interpolator* out, pmg_layout layout_coarse, pmg_layout layout_fine,
                                    int degree_coarse, int degree_fine, int32_t ncells,
                                    const int32_t* dofmap_coarse, const int32_t* dofmap_fine,
                                    const int32_t* lcells, int32_t n_lcells, const int32_t* bcells,
                                    int32_t n_bcells, pmg_laplacian fine_operator, int node_order,
                                    const int32_t* custom_coarse, const int32_t* custom_fine, pmg_stream stream);
int pmg_interpolator_destroy(pmg_interpolator ip);
/* interpolate(Q1_vector, Q2_vector), :186-239: prolongation, updates the ghosts of `coarse`. */
int pmg_interpolator_interpolate(pmg_interpolator ip, double* coarse, double* fine,
                                 pmg_stream stream);
/* fine += P coarse in one pass: interpolate + the axpy of src/pmg.hpp:123-129
 * (only for interpolators created with an operator). */
int pmg_interpolator_interpolate_add(pmg_interpolator ip, double* coarse, double* fine,
                                     pmg_stream stream);
/* reverse_interpolate(Q2_vector, Q1_vector), :246-303: restriction (multiplicity-
 * weighted transpose), updates the ghosts of `fine`, zeroes `coarse` first. */
int pmg_interpolator_reverse_interpolate(pmg_interpolator ip, double* fine, double* coarse,
                                         pmg_stream stream);

/* ---- V-cycle (acc::MultigridPreconditioner, src/pmg.hpp:16-184) -----------
 * Levels are ordered coarse -> fine like the reference's vectors.  The handle
 * allocates its own work vectors (:35-41).  coarse solve = smoother[0] unless
 * pmg_multigrid_set_coarse_solver is used (:106-109). */
int pmg_multigrid_create(pmg_multigrid* out, int nlevels, const pmg_layout* layouts,
                         const int8_t* bc_marker_coarsest);
int pmg_multigrid_destroy(pmg_multigrid mg);
int pmg_multigrid_set_operators(pmg_multigrid mg, const pmg_laplacian* ops);          /* :48 */
int pmg_multigrid_set_solvers(pmg_multigrid mg, const pmg_chebyshev* smoothers);      /* :44 */
int pmg_multigrid_set_interpolators(pmg_multigrid mg, const pmg_interpolator* interp); /* :50-53 */
/* set_coarse_solver, :46,106-109: the coarsest level is solved by `coarse` -- pmg_cg_solve on
 * operator[0] from a zero initial guess with the solver's own iteration cap and tolerance --
 * instead of smoother[0]; NULL restores the smoother.  The reference's coarse solver is a Krylov
 * solve too (PETSc KSPCG, at most 60 iterations, src/amg.hpp:36-44) but preconditioned by hypre
 * BoomerAMG, third-party arithmetic that is out of scope here: the library's CG is
 * Jacobi-preconditioned.  `coarse` must live on the coarsest layout and outlive `mg`. */
int pmg_multigrid_set_coarse_solver(pmg_multigrid mg, pmg_cg coarse);
/* The same hook for ANY coarse solver -- the reference's CoarseSolver concept is "a type with
 * solve(Vector& x, Vector& b)" (src/amg.hpp:67, called at src/pmg.hpp:106-107): `solve` receives
 * the coarsest-level solution (zeroed beforehand) and right-hand side as device arrays of
 * size_local + num_ghosts doubles and the stream the cycle runs on; it returns 0 on success.
 * NULL restores the smoother.  (pmg_amg below is the library's own such solver.) */
typedef int (*pmg_coarse_solve_fn)(void* user, double* x, double* b, pmg_stream stream);
int pmg_multigrid_set_coarse_callback(pmg_multigrid mg, pmg_coarse_solve_fn solve, void* user);
/* ---- algebraic multigrid for the coarsest level (the role of CoarseSolverType<T>, src/amg.hpp) ----
 * The reference solves its degree-1 level with PETSc KSPCG (<= 60 iterations, default relative
 * tolerance 1e-5) preconditioned by hypre BoomerAMG (src/amg.hpp:33-47).  Those are third-party;
 * pmg_amg is the library's own solver for that slot: smoothed-aggregation AMG on the assembled
 * degree-1 matrix (built on the host from the operator's geometry tensor; cycles run on the device
 * with the same Chebyshev/Jacobi smoother as the p-levels), used either
 *   - Krylov mode (default, the reference's shape): CG on the matrix-free operator preconditioned
 *     by one AMG V-cycle, zero initial guess, max_iter / rtol as set (60, 1e-5); or
 *   - stationary mode (pmg_amg_set_cycles(n > 0)): n AMG V-cycles from a zero initial guess -- a
 *     fixed linear operator without host synchronisation (single rank only).
 * `op` must be a degree-1 operator and outlive the solver.
 * Several ranks: pmg_amg_create builds the hierarchy of the rank's own block (a block preconditioner
 * for the Krylov mode; iteration counts grow with the number of ranks).  pmg_amg_create_replicated
 * gathers the owned rows of all ranks -- `global_index` [size_local + num_ghosts] (host) is the global
 * number of every local dof, dolfinx's IndexMap::local_to_global; n_global < 2^31 -- into the global
 * matrix on every rank: a solve is then one all-reduce of the zero-padded right-hand side (on the
 * layout's communicator, or through its allreduce callback) and the single-rank solve of the whole
 * coarse problem on every rank; both modes work, with one rank's iteration counts.  Collective. */
int pmg_amg_create(pmg_amg* out, pmg_laplacian op, pmg_stream stream);
int pmg_amg_create_replicated(pmg_amg* out, pmg_laplacian op, const int64_t* global_index, int64_t n_global,
                              pmg_stream stream);
/* The same solver set up WITHOUT gathering the global degree-1 matrix (round 4): every rank aggregates its owned
 * dofs on its own block (ghost couplings lumped), smooths its prolongator rows with that block, obtains the prolongator
 * rows of its ghost dofs through the layout's own forward scatter (one layer of overlap; any exchange mechanism),
 * forms its rows of P^T A P, and only level 1 -- about 1/9 of level 0 -- is gathered and coarsened further on every
 * rank.  The solve is the replicated form's with the distributed fine level (level 0 smoothed on the partitioned
 * operator, one all-reduce of a level-1 vector per cycle); pmg_amg_set_distributed_fine_level(amg, 0) is refused.
 * The hierarchy depends on the partition (aggregates do not cross rank boundaries); iteration counts stay within one
 * of the gathered (= single-rank) hierarchy's: 10 against 10 on eight ranks of 64^3 cells
 * (tools/amg_setup_scaling.py), where the rank-local block preconditioner of pmg_amg_create takes 40.  Collective. */
int pmg_amg_create_distributed(pmg_amg* out, pmg_laplacian op, const int64_t* global_index, int64_t n_global,
                               pmg_stream stream);
int pmg_amg_destroy(pmg_amg amg);
int pmg_amg_set_smoother_iterations(pmg_amg amg, int k); /* Chebyshev degree per pre/post smooth (2) */
int pmg_amg_set_cycles(pmg_amg amg, int cycles);
int pmg_amg_set_krylov(pmg_amg amg, int max_iter, double rtol);
/* Replicated form only.  1 (the default when the hierarchy has more than one level): level 0 of the hierarchy -- the
 * degree-1 level itself, three quarters of a cycle's work -- is smoothed on the PARTITIONED operator (matrix-free
 * application with its halo exchange), each rank restricts its owned residual, ONE all-reduce of a level-1 vector
 * (about 1/9 of the level-0 size) forms the replicated right-hand side of level 1, and only the levels from 1 down are
 * solved redundantly on every rank.  0: the whole hierarchy replicated, one all-reduce of a level-0 vector per solve
 * (the redundant work then grows with the rank count).  Same arithmetic up to summation order either way. */
int pmg_amg_set_distributed_fine_level(pmg_amg amg, int enable);
/* solve(x, b) of src/amg.hpp:67-68; *iterations = CG iterations (or cycles) used. */
int pmg_amg_solve(pmg_amg amg, double* x, const double* b, int* iterations, pmg_stream stream);
/* One V-cycle of the hierarchy from a zero initial guess: x = M b (the preconditioner alone). */
int pmg_amg_cycle(pmg_amg amg, double* x, const double* b, pmg_stream stream);
int pmg_amg_num_levels(pmg_amg amg);
int pmg_amg_level_info(pmg_amg amg, int level, long long* rows, long long* nnz, double* lambda_max);
/* Host copy of the hierarchy (tests): which = 0 the matrix of `level`, 1 the prolongator from
 * level + 1 to `level`; CSR.  Call with NULL arrays first to learn the sizes. */
int pmg_amg_export(pmg_amg amg, int level, int which, long long* rows, long long* cols, long long* nnz,
                   int32_t* rowptr, int32_t* colidx, double* values);
/* set_coarse_solver (src/pmg.hpp:46) with the library's AMG; NULL restores the smoother. */
int pmg_multigrid_set_coarse_amg(pmg_multigrid mg, pmg_amg amg);

/* apply(x = rhs, y = initial guess in / result out, verbose), :56-155.  If
 * rnorm is non-NULL the final residual norm ||b - A y|| is computed (the
 * reference prints it when verbose, :147-150) -- this costs one extra apply and a
 * host synchronisation. */
int pmg_multigrid_apply(pmg_multigrid mg, const double* rhs, double* y, double* rnorm,
                        pmg_stream stream);
/* Exchanges of a cycle on several ranks: every operator application refreshes the ghosts of its input
 * (src/laplacian.hpp:378,425) -- except the first application of a post-smooth when the transfers share the operator's
 * patches: the iterate leaves its pre-smooth with current ghosts (the smoother adds the ghost entries of every applied
 * correction), the prolongation computes the coarse correction on the ghost cells too, so u + P u_c has current ghosts
 * without an exchange of its own.  Three levels, Chebyshev(3): 17 exchanges per cycle instead of 19.
 * PMG_LOCAL_CORRECTION=0 in the environment restores one exchange per application. */
/* hipGraph replay of the cycle.  enable > 0: pmg_multigrid_apply (and the V-cycle preconditioner inside
 * pmg_cg_solve) captures the cycle's ~120 launches into a graph the first time it sees a (rhs, y) pair and replays it
 * afterwards with one hipGraphLaunch on the caller's stream: same kernels, same order, same results.  enable == 0:
 * never.  enable < 0, the default: automatic -- on one rank eager (the launches run ahead of the GPU anyway: a replay
 * buys nothing), on several ranks replayed wherever the capture holds nothing but kernels (every exchange through halo
 * windows, the cycle's reductions through a communicator made of windows), because an eager exchange costs the host
 * more than the GPU; a cycle whose exchanges are RCCL calls is replayed on request only (enable > 0, or
 * PMG_GRAPH_AUTO=rccl in the environment; PMG_GRAPH_AUTO=0 turns the automatic choice off): that capture has run on
 * one GPU, never between two.
 * Captured only when nothing in the cycle needs the host: no exchange callbacks, no Krylov / callback coarse solver, no
 * in-situ profiling; otherwise the call runs eagerly as before.  Layouts with a pmg_comm ARE captured: the grouped
 * send / receive of every halo exchange is recorded into the graph (on a HIP >= 7.2 runtime as a parallel branch that
 * keeps its overlap with the interior cells, pmg_comm_capture_overlaps).  A change of a smoother's iteration count or
 * bound, of a geometry mode or of the coarse solver is noticed (new graph); calling this function again drops the
 * cached graphs (do that after replacing an operator's diagonal or any caller-owned array in place). */
int pmg_multigrid_set_graph(pmg_multigrid mg, int enable);
long long pmg_multigrid_graph_replays(pmg_multigrid mg);
/* Number of stiffness-kernel launches issued by the last pmg_multigrid_apply,
 * per level (coarse -> fine); for the byte accounting in bench.py. */
int pmg_multigrid_apply_counts(pmg_multigrid mg, int* counts, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* PMG_AMD_H */
