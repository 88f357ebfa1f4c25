// Solver layer: 4th-kind Chebyshev smoother, Jacobi-PCG with Lanczos eigenvalue
// estimate, p-multigrid V-cycle.  Replaces src/chebyshev.hpp, src/cg.hpp and
// src/pmg.hpp.  Everything is enqueued on the caller's stream; the only host
// synchronisations are the CG dot products (their values steer the iteration,
// src/cg.hpp:182,195,206) and an optional residual norm of the V-cycle.
//
// The smoother issues one fused vector pass per Chebyshev step (r -= q;
// z = c1 z + c2 D^-1 r; x += z : 5 reads + 3 writes per dof; x takes each correction in
// the pass that computes it, so no pass follows the last apply) where the reference issues
// 5 Thrust launches (src/chebyshev.hpp:73-83), and the V-cycle drops the residual
// recomputations that only feed log lines (src/pmg.hpp:76-89,114-117,132-143):
// the residual a pre-smooth leaves in its recurrence IS b - A u (:86-87), and the
// last A z of a post-smooth changes neither x nor anything that is read again.
#include "common.hpp"

#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace pmg;

namespace pmg
{
int laplacian_apply(pmg_laplacian op, double* in, double* out, hipStream_t s);
const double* laplacian_diag_inv(pmg_laplacian op);
pmg_layout laplacian_layout(pmg_laplacian op);
long long laplacian_launches(pmg_laplacian op);
int interp_prolong(pmg_interpolator ip, double* coarse, double* fine, hipStream_t s);
int interp_restrict(pmg_interpolator ip, double* fine, double* coarse, hipStream_t s);
int interp_prolong_add(pmg_interpolator ip, double* coarse, double* fine, hipStream_t s);
bool interp_is_patched(pmg_interpolator ip);
bool interp_restricts_difference(pmg_interpolator ip);
int interp_restrict_difference(pmg_interpolator ip, double* fine, const double* fine_sub, double* coarse,
                               hipStream_t s);
int amg_solve(pmg_amg amg, double* x, const double* b, hipStream_t s);
pmg_layout amg_layout(pmg_amg amg);
long long amg_capture_state(pmg_amg amg);
long long laplacian_capture_state(pmg_laplacian op);
void launch_cheb_first(int n, double* x, double* r, double* z, const double* q, const double* dinv,
                       double c1, double c2, int x_final, hipStream_t s, double* clear_q = nullptr, int n_total = 0);
bool laplacian_wants_zeroed_output(pmg_laplacian op);
int laplacian_apply_zeroed(pmg_laplacian op, double* in, double* out, hipStream_t s);
int laplacian_apply_ghosts_current(pmg_laplacian op, double* in, double* out, hipStream_t s);
} // namespace pmg

struct pmg_chebyshev_s
{
  pmg_layout layout = nullptr;
  double eig_min = 0, eig_max = 1;
  int max_iter = 1;
  double *r = nullptr, *z = nullptr, *q = nullptr; // work vectors (src/chebyshev.hpp:28-32)
};

struct pmg_cg_s
{
  pmg_layout layout = nullptr;
  int max_iter = 0;
  double rtol = 0;
  bool store = false;
  double *r = nullptr, *y = nullptr, *p = nullptr; // src/cg.hpp:101-104
  double* zold = nullptr;                           // flexible variant only
  bool flexible = false;
  std::vector<double> alphas, betas, residuals;
};

struct pmg_multigrid_s
{
  int L = 0;
  std::vector<pmg_layout> layouts;
  const int8_t* bc0 = nullptr;
  std::vector<pmg_laplacian> ops;
  std::vector<pmg_chebyshev> smoothers;
  std::vector<pmg_interpolator> interps;
  std::vector<double*> u, b; // per level (finest level uses the caller's vectors)
  std::vector<int> counts;
  // optional coarse solver (src/pmg.hpp:106-107): the library's CG, or any solve(x, b) of the caller
  pmg_cg coarse = nullptr;
  pmg_coarse_solve_fn coarse_fn = nullptr;
  void* coarse_user = nullptr;
  pmg_amg coarse_amg = nullptr;
  // hipGraph replay of the cycle (pmg_multigrid_set_graph): one executable graph per
  // (rhs, y, zero-guess, configuration) seen
  struct GraphEntry
  {
    const double* rhs;
    double* y;
    bool y_zero;
    uint64_t config;
    hipGraphExec_t exec;
    std::vector<int> counts;
  };
  int graph_mode = -1; // pmg_multigrid_set_graph: 1 replay when capturable, 0 never, -1 (default) automatic: use_graph()
  std::vector<GraphEntry> graphs;
  hipStream_t capture_stream = nullptr;
  long long graph_replays = 0;
};

namespace pmg
{
// src/chebyshev.hpp:46-91.
//   x_zero : x is known to be 0 on entry (A 0 = 0, so r = b and x := z)
//   need_r : ResidualNone    -- only x is wanted: the loop's last apply is skipped;
//            ResidualUpdated -- w.r = b - A x on return (costs that last apply);
//            ResidualSplit   -- as ResidualUpdated, but the last "r -= q" is left to the consumer when *split
//                               comes back true: b - A x = w.r - w.q (a restriction subtracts while it gathers; a
//                               one-step smoother, whose only kernel updates x and r together, returns false).
int cheb_iterate(const ChebWork& w, const ApplyFn& A, const double* dinv, int n, double lmax, int max_iter,
                 double* x, const double* b, int need_r, bool x_zero, hipStream_t s, bool* split,
                 const ApplyFn* A_zeroed, int n_total, bool track_ghosts, const ApplyFn* A_first)
{
  if (split)
    *split = false;
  double* const clear_q = A_zeroed ? w.q : nullptr; // the vector kernels leave w.q zero for the next application
  const ApplyFn& An = A_zeroed ? *A_zeroed : A;    // ... which then needs no zero-fill
  const double c0 = 4.0 / (3.0 * lmax);
  const int ng = track_ghosts && n_total > n ? n_total - n : 0; // ghost entries of x kept current (see common.hpp)
  if (x_zero)
  {
    launch_cheb_init(n, w.r, w.z, b, nullptr, dinv, c0, s, clear_q, n_total);
    if (ng > 0)
      launch_zero(ng, x + n, s);
  }
  else
  {
    PMG_TRY(A_first ? (*A_first)(x, w.q) : A(x, w.q));                      // :56 (refreshes the ghosts of x)
    launch_cheb_init(n, w.r, w.z, b, w.q, dinv, c0, s, clear_q, n_total); // :57,67-68
  }
  // x absorbs the correction z_{i+1} in the step kernel that computes it (the first step adds z_1 and z_2), so
  // after the last application only the residual is left to update, and only where it is wanted.
  for (int i = 1; i <= max_iter; ++i)
  {
    const bool last = (i == max_iter);
    if (last && !need_r)
    {
      if (max_iter == 1) // a one-step smoother: z_1 is all there is (:73)
      {
        if (x_zero)
          PMG_HIP(hipMemcpyAsync(x, w.z, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
        else
          launch_add(n, x, w.z, s);
      }
      break;
    }
    PMG_TRY(An(w.z, w.q)); // :76
    if (ng > 0)             // the ghosts of z are current now: the iterate's ghosts absorb them
      launch_add(ng, x + n, w.z + n, s);
    if (last) // need_r: the new z would not be used, only x and r are (:73,77)
    {
      if (max_iter == 1)
        launch_cheb_last(n, x, w.r, w.z, w.q, x_zero, s);
      else if (need_r == ResidualSplit && split)
        *split = true;
      else
        launch_cheb_residual(n, w.r, w.q, s);
      break;
    }
    const double c1 = (2.0 * i - 1.0) / (2.0 * i + 3.0);
    const double c2 = (8.0 * i + 4.0) / (2.0 * i + 3.0) / lmax;
    // the last correction enters x here (1); when no residual is wanted either, r and z are dead behind this step (2)
#ifdef PMG_CHEB_KEEP_RZ // timing comparison: always write r and z
    const int x_final = (i + 1 == max_iter) ? 1 : 0;
#else
    const int x_final = (i + 1 == max_iter) ? (need_r == ResidualNone ? 2 : 1) : 0;
#endif
    if (x_zero && i == 1)
      launch_cheb_first(n, x, w.r, w.z, w.q, dinv, c1, c2, x_final, s, clear_q, n_total);
    else
      launch_cheb_step(n, x, w.r, w.z, w.q, dinv, c1, c2, i == 1, x_final, s, clear_q, n_total); // :73,77,80-83
  }
  if (max_iter == 0 && x_zero)
    launch_zero(n, x, s);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}
} // namespace pmg

namespace
{
int mg_apply_graph(pmg_multigrid mg, const double* rhs, double* y, bool y_zero, hipStream_t s, bool* done);
bool use_graph(pmg_multigrid mg);
// every captured cycle holds the device pointers of the objects it was captured with: any change of
// the multigrid's parts invalidates all of them
void drop_graphs(pmg_multigrid mg);

int alloc_vec(pmg_layout l, double** p)
{
  size_t n = l->total() ? l->total() : 1;
  PMG_HIP(hipMalloc(p, sizeof(double) * n));
  PMG_HIP(hipMemset(*p, 0, sizeof(double) * n));
  PMG_HIP(hipStreamSynchronize(nullptr)); // the fill runs on the null stream, which non-blocking streams do not order
  return PMG_OK;
}

// src/chebyshev.hpp:46-91 on the operator `A` (see cheb_iterate)
// track_ghosts / x_ghosts_current: the exchange bookkeeping of the V-cycle, see cheb_iterate (common.hpp)
int cheb_solve(pmg_chebyshev sm, pmg_laplacian A, double* x, const double* b, int need_r,
               bool x_zero, hipStream_t s, bool* split = nullptr, bool track_ghosts = false,
               bool x_ghosts_current = false)
{
  pmg_layout l = sm->layout;
  PMG_REQUIRE(laplacian_layout(A) == l, "Chebyshev: operator and smoother layouts differ");
  const ChebWork w{sm->r, sm->z, sm->q};
  // eig_range[0] is unused (src/chebyshev.hpp:51); no per-call D2D copy of the diagonal (:53)
  const ApplyFn apply = [A, s](double* in, double* out) { return laplacian_apply(A, in, out, s); };
  const ApplyFn apply_zeroed = [A, s](double* in, double* out) { return laplacian_apply_zeroed(A, in, out, s); };
  const ApplyFn apply_local = [A, s](double* in, double* out) { return laplacian_apply_ghosts_current(A, in, out, s); };
  return cheb_iterate(w, apply, laplacian_diag_inv(A), l->size_local, sm->eig_max, sm->max_iter, x, b, need_r, x_zero,
                      s, split, laplacian_wants_zeroed_output(A) ? &apply_zeroed : nullptr, l->total(), track_ghosts,
                      x_ghosts_current ? &apply_local : nullptr);
}

// Exchange bookkeeping on several ranks (round 4).  u_i leaves its pre-smooth with current ghosts (the smoother adds
// the ghosts of every applied correction, cheb_iterate); the patch form of the prolongation computes the correction on
// EVERY local cell, ghost cells included, from the coarse ghosts it has just received, and adds it to every patch dof it
// writes first -- ghost dofs too: u_i + P u_{i-1} therefore has current ghosts without an exchange of its own, and the
// first application of the post-smooth runs without one (19 -> 17 exchanges per cycle of three levels with k = 3).
// Values at a ghost dof are computed from another cell than on the owner: equal to rounding, not bit for bit.
bool local_correction(pmg_multigrid mg, int level)
{
  if (const char* e = std::getenv("PMG_LOCAL_CORRECTION")) // tests / measurements: 0 = an exchange per application
    if (e[0] == '0')
      return false;
  return level > 0 && mg->layouts[level]->num_ghosts > 0 && interp_is_patched(mg->interps[level - 1]);
}

// src/pmg.hpp:56-155 (lean form, see the file header)
int mg_apply(pmg_multigrid mg, const double* rhs, double* y, bool y_zero, hipStream_t s)
{
  const int L = mg->L;
  PMG_REQUIRE((int)mg->ops.size() == L && (int)mg->smoothers.size() == L
                  && (int)mg->interps.size() == L - 1,
              "MultigridPreconditioner: operators / solvers / interpolators not set");
  std::vector<long long> before(L);
  for (int i = 0; i < L; ++i)
    before[i] = laplacian_launches(mg->ops[i]);
  // u[L-1] = y, b[L-1] = rhs (:65,68) -- used in place; u[i<L-1] = 0 (:63-64) is
  // folded into the smoothers' x_zero path.
  Range cycle("pmg:vcycle");
  mg->u[L - 1] = y;
  for (int i = L - 1; i > 0; --i)
  {
    const double* bi = (i == L - 1) ? rhs : mg->b[i];
    const bool zero = (i == L - 1) ? y_zero : true;
    bool split = false;
    {
      Range rg("pmg:pre_smooth");
      // (several ranks: the iterate leaves the pre-smooth with current ghosts, see `local_correction` below)
      PMG_TRY(cheb_solve(mg->smoothers[i], mg->ops[i], mg->u[i], bi,
                         interp_restricts_difference(mg->interps[i - 1]) ? ResidualSplit : ResidualUpdated, zero, s,
                         &split, local_correction(mg, i))); // :83-87
    }
    Range rg("pmg:restrict");
    if (split) // the residual r - q is formed by the restriction's gather
      PMG_TRY(interp_restrict_difference(mg->interps[i - 1], mg->smoothers[i]->r, mg->smoothers[i]->q,
                                         mg->b[i - 1], s));
    else
      PMG_TRY(interp_restrict(mg->interps[i - 1], mg->smoothers[i]->r, mg->b[i - 1], s)); // :92
  }
  if (L > 1)
    launch_mask_bc(mg->layouts[0]->size_local, mg->b[0], mg->bc0, s); // :100-103
  {
    Range rg("pmg:coarse_solve");
    const double* b0 = (L == 1) ? rhs : mg->b[0];
    const bool zero = (L == 1) ? y_zero : true;
    if (mg->coarse_amg && L > 1) // :106-107 with the library's own AMG (amg.hip)
      PMG_TRY(amg_solve(mg->coarse_amg, mg->u[0], mg->b[0], s));
    else if ((mg->coarse || mg->coarse_fn) && L > 1) // :106-107, KSP-style: zero initial guess
    {
      launch_zero(mg->layouts[0]->total(), mg->u[0], s);
      if (mg->coarse_fn)
      {
        if (mg->coarse_fn(mg->coarse_user, mg->u[0], mg->b[0], (pmg_stream)s) != 0)
          return fail(PMG_ERR_INVALID, "the coarse-solver callback failed");
      }
      else
      {
        int its = 0;
        PMG_TRY(pmg_cg_solve(mg->coarse, mg->ops[0], mg->u[0], b0, nullptr, &its, (pmg_stream)s));
      }
    }
    else
      PMG_TRY(cheb_solve(mg->smoothers[0], mg->ops[0], mg->u[0], b0, ResidualNone, zero, s)); // :109
  }
  for (int i = 0; i < L - 1; ++i)
  {
    {
      Range rg("pmg:prolong");
      if (interp_is_patched(mg->interps[i]))
        PMG_TRY(interp_prolong_add(mg->interps[i], mg->u[i], mg->u[i + 1], s)); // :123 + :129 in one pass
      else
      {
        double* du = mg->smoothers[i + 1]->q;                            // work vector as du
        PMG_TRY(interp_prolong(mg->interps[i], mg->u[i], du, s));        // :123
        launch_add(mg->layouts[i + 1]->size_local, mg->u[i + 1], du, s); // :129
      }
    }
    Range rg("pmg:post_smooth");
    const double* bi = (i + 1 == L - 1) ? rhs : mg->b[i + 1];
    // with local_correction the ghosts of u are current here: the post-smooth's first application needs no exchange
    PMG_TRY(cheb_solve(mg->smoothers[i + 1], mg->ops[i + 1], mg->u[i + 1], bi, ResidualNone, false, s, nullptr, false,
                       local_correction(mg, i + 1))); // :138
  }
  for (int i = 0; i < L; ++i)
    mg->counts[i] = (int)(laplacian_launches(mg->ops[i]) - before[i]);
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}
} // namespace

// ------------------------------------------------------------- Chebyshev --
extern "C" int pmg_chebyshev_create(pmg_chebyshev* out, pmg_layout layout, double eig_min,
                                    double eig_max)
{
  PMG_REQUIRE(out && layout, "pmg_chebyshev_create: NULL argument");
  PMG_REQUIRE(eig_max > 0.0, "pmg_chebyshev_create: eig_max must be positive");
  auto* sm = new pmg_chebyshev_s;
  HandleGuard<pmg_chebyshev> guard(sm, pmg_chebyshev_destroy);
  sm->layout = layout;
  sm->eig_min = eig_min;
  sm->eig_max = eig_max;
  PMG_TRY(alloc_vec(layout, &sm->r));
  PMG_TRY(alloc_vec(layout, &sm->z));
  PMG_TRY(alloc_vec(layout, &sm->q));
  *out = guard.release();
  return PMG_OK;
}

extern "C" int pmg_chebyshev_destroy(pmg_chebyshev sm)
{
  if (!sm)
    return PMG_OK;
  (void)hipFree(sm->r);
  (void)hipFree(sm->z);
  (void)hipFree(sm->q);
  delete sm;
  return PMG_OK;
}

extern "C" int pmg_chebyshev_set_max_iterations(pmg_chebyshev sm, int max_iter)
{
  PMG_REQUIRE(sm && max_iter >= 0, "pmg_chebyshev_set_max_iterations: bad argument");
  sm->max_iter = max_iter;
  return PMG_OK;
}

extern "C" int pmg_chebyshev_solve(pmg_chebyshev sm, pmg_laplacian A, double* x, const double* b,
                                   pmg_stream stream)
{
  PMG_REQUIRE(sm && A && x && b, "pmg_chebyshev_solve: NULL argument");
  return cheb_solve(sm, A, x, b, ResidualNone, false, S(stream));
}

// -------------------------------------------------------------------- CG --
extern "C" int pmg_cg_create(pmg_cg* out, pmg_layout layout)
{
  PMG_REQUIRE(out && layout, "pmg_cg_create: NULL argument");
  auto* cg = new pmg_cg_s;
  HandleGuard<pmg_cg> guard(cg, pmg_cg_destroy);
  cg->layout = layout;
  PMG_TRY(alloc_vec(layout, &cg->r));
  PMG_TRY(alloc_vec(layout, &cg->y));
  PMG_TRY(alloc_vec(layout, &cg->p));
  *out = guard.release();
  return PMG_OK;
}

extern "C" int pmg_cg_destroy(pmg_cg cg)
{
  if (!cg)
    return PMG_OK;
  (void)hipFree(cg->r);
  (void)hipFree(cg->y);
  (void)hipFree(cg->p);
  (void)hipFree(cg->zold);
  delete cg;
  return PMG_OK;
}

extern "C" int pmg_cg_set_max_iterations(pmg_cg cg, int max_iter)
{
  PMG_REQUIRE(cg && max_iter >= 0, "pmg_cg_set_max_iterations: bad argument");
  cg->max_iter = max_iter; // src/cg.hpp:107-113
  cg->alphas.reserve(max_iter);
  cg->betas.reserve(max_iter);
  cg->residuals.reserve(max_iter);
  return PMG_OK;
}

extern "C" int pmg_cg_set_tolerance(pmg_cg cg, double rtol)
{
  PMG_REQUIRE(cg, "pmg_cg_set_tolerance: NULL argument");
  cg->rtol = rtol;
  return PMG_OK;
}

extern "C" int pmg_cg_store_coefficients(pmg_cg cg, int flag)
{
  PMG_REQUIRE(cg, "pmg_cg_store_coefficients: NULL argument");
  cg->store = flag != 0;
  return PMG_OK;
}

namespace pmg
{
// src/cg.hpp:147-222 over any operator `A`; the preconditioner is `M` (z = M r) if given, else the
// diagonal `dinv` (the reference's hard-wired Jacobi, :161,192).
int cg_iterate(pmg_cg cg, const ApplyFn& A, const double* dinv, const PrecondFn* M, bool flexible, double* x,
               const double* b, int* iterations, hipStream_t s)
{
  pmg_layout l = cg->layout;
  const int n = l->size_local;
  double *r = cg->r, *y = cg->y, *p = cg->p;
  const bool precond = M != nullptr;
  auto precondition = [&](double* out_z, const double* in_r) -> int
  {
    if (precond)
      return (*M)(out_z, in_r);
    launch_pointwise(n, out_z, in_r, dinv, s); // :161,192
    return PMG_OK;
  };
  // Flexible CG (Polak-Ribiere beta) for a preconditioner that is not a fixed linear operator --
  // the V-cycle with a Krylov coarse solver: one more vector (the previous z), one more dot product
  const bool flex = flexible && precond;
  if (flex && !cg->zold)
    PMG_TRY(alloc_vec(l, &cg->zold));
  PMG_TRY(A(x, y)); // :159
  launch_axpy(n, r, -1.0, y, b, s);     // :160
  PMG_TRY(precondition(p, r));          // :161
  if (flex)
    PMG_HIP(hipMemcpyAsync(cg->zold, p, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
  double rnorm0;
  PMG_TRY(dot_host(l, p, r, &rnorm0, s)); // :163
  double rnorm = rnorm0;
  const double rtol2 = cg->rtol * cg->rtol;
  if (!std::isfinite(rnorm0) || rnorm0 < 0.0) // a poisoned right-hand side / operator, or an indefinite preconditioner
    return fail(PMG_ERR_NUMERIC, "CG: r . M^-1 r = %g at the start is not a non-negative finite number", rnorm0);
  if (rnorm0 == 0.0) // r = 0: x already solves the system; :206 would divide by zero
  {
    if (iterations)
      *iterations = 0;
    cg->residuals.assign(1, rnorm0);
    return PMG_OK;
  }
  // The loop keeps its scalars on the device: p.y, r.z (and r.z_old) are reduced there (one
  // ncclAllReduce each when the layout has a communicator), alpha and beta are formed inside the
  // update kernels, and the host reads the values ONCE per iteration, at its end, for the stopping
  // test and the Lanczos record -- the reference takes two blocking MPI_Allreduce per iteration
  // (src/cg.hpp:182,195).  Result slots: 1 = p.y, 2 = r.z, 3 = r.z_old.
  int k = 0;
  while (k < cg->max_iter)
  {
    ++k;
    Range range("pmg:cg_iteration"); // src/cg.hpp:174,219
    PMG_TRY(A(p, y)); // :179
    PMG_TRY(dot_async(l, p, y, 1, s));
    PMG_TRY(reduce_slots_async(l, 1, 1, false, s));
    if (precond)
    {
      launch_cg_update2(n, x, r, p, y, rnorm, red_slot(l, 1), s); // :182-189
      PMG_TRY(precondition(y, r));
    }
    else
      launch_cg_update(n, x, r, y, p, dinv, rnorm, red_slot(l, 1), s); // :182-192 fused
    PMG_TRY(dot_async(l, r, y, 2, s)); // :195
    if (flex)
      PMG_TRY(dot_async(l, r, cg->zold, 3, s));
    PMG_TRY(reduce_slots_async(l, 2, flex ? 2 : 1, false, s));
    // :196,211 -- issued before the stopping test is known; if the loop ends here the new
    // direction is simply never used
    launch_cg_direction(n, p, y, rnorm, red_slot(l, 2), flex ? red_slot(l, 3) : nullptr, s);
    if (flex)
      PMG_HIP(hipMemcpyAsync(cg->zold, y, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    double v[3] = {0, 0, 0};
    PMG_TRY(fetch_slots(l, 1, flex ? 3 : 2, v, s)); // the iteration's one host synchronisation
    const double alpha = rnorm / v[0]; // :182
    const double rnorm_new = v[1];
    // flexible: r_new . (z_new - z_old) / (r_old . z_old)
    const double beta = (flex ? rnorm_new - v[2] : rnorm_new) / rnorm;
    rnorm = rnorm_new;
    if (rnorm / rnorm0 < rtol2) // :206
      break;
    if (cg->store) // :213-218
    {
      cg->alphas.push_back(alpha);
      cg->betas.push_back(beta);
      cg->residuals.push_back(rnorm);
    }
  }
  if (!cg->store)
  {
    cg->residuals.clear();
    cg->residuals.push_back(rnorm);
  }
  if (iterations)
    *iterations = k;
  PMG_HIP(hipGetLastError());
  return PMG_OK;
}
} // namespace pmg

extern "C" int pmg_cg_solve(pmg_cg cg, pmg_laplacian A, double* x, const double* b,
                            pmg_multigrid precond, int* iterations, pmg_stream stream)
{
  PMG_REQUIRE(cg && A && x && b, "pmg_cg_solve: NULL argument");
  PMG_REQUIRE(laplacian_layout(A) == cg->layout, "pmg_cg_solve: operator and solver layouts differ");
  hipStream_t s = S(stream);
  const ApplyFn apply = [A, s](double* in, double* out) { return laplacian_apply(A, in, out, s); };
  // the V-cycle from a zero initial guess: folded into the smoothers' x_zero path, no memset of z
  const PrecondFn vcycle = [precond, s](double* z, const double* r) -> int
  {
    bool done = false;
    if (use_graph(precond))
      PMG_TRY(mg_apply_graph(precond, r, z, true, s, &done));
    return done ? PMG_OK : mg_apply(precond, r, z, true, s);
  };
  return cg_iterate(cg, apply, laplacian_diag_inv(A), precond ? &vcycle : nullptr, cg->flexible, x, b, iterations, s);
}

extern "C" int pmg_cg_set_flexible(pmg_cg cg, int flag)
{
  PMG_REQUIRE(cg, "pmg_cg_set_flexible: NULL argument");
  cg->flexible = flag != 0;
  return PMG_OK;
}

extern "C" int pmg_cg_coefficients(pmg_cg cg, double* alphas, double* betas, int capacity)
{
  PMG_REQUIRE(cg, "pmg_cg_coefficients: NULL argument");
  int n = (int)cg->alphas.size();
  for (int i = 0; i < n && i < capacity; ++i)
  {
    if (alphas)
      alphas[i] = cg->alphas[i];
    if (betas)
      betas[i] = cg->betas[i];
  }
  return n;
}

// src/cg.hpp:121-142
extern "C" int pmg_cg_compute_eigenvalues(pmg_cg cg, double* eigs, int capacity)
{
  PMG_REQUIRE(cg && eigs, "pmg_cg_compute_eigenvalues: NULL argument");
  const int ne = (int)cg->alphas.size();
  if (ne < 2)
    return fail(PMG_ERR_INVALID, "Insufficient data to compute eigenvalues"); // :125
  PMG_REQUIRE(capacity >= ne, "pmg_cg_compute_eigenvalues: capacity %d < %d", capacity, ne);
  std::vector<double> d(ne, 0.0), e(ne, 0.0);
  for (int i = 0; i < ne; ++i)
    d[i] = 1.0 / cg->alphas[i];
  for (int i = 0; i < ne - 1; ++i)
  {
    d[i + 1] += cg->betas[i] / cg->alphas[i];
    e[i] = std::sqrt(cg->betas[i]) / cg->alphas[i];
  }
  if (pmg_tqli(d.data(), e.data(), ne) != PMG_OK)
    return fail(PMG_ERR_NUMERIC, "Eigenvalue estimate failed"); // :138
  std::sort(d.begin(), d.end());
  for (int i = 0; i < ne; ++i)
    eigs[i] = d[i];
  return ne;
}

extern "C" int pmg_cg_residual(pmg_cg cg, double* rnorm)
{
  PMG_REQUIRE(cg && rnorm, "pmg_cg_residual: NULL argument");
  PMG_REQUIRE(!cg->residuals.empty(), "pmg_cg_residual: no residual recorded");
  *rnorm = cg->residuals.back(); // src/cg.hpp:144
  return PMG_OK;
}

// -------------------------------------------------------------- multigrid --
extern "C" int pmg_multigrid_create(pmg_multigrid* out, int nlevels, const pmg_layout* layouts,
                                    const int8_t* bc_marker_coarsest)
{
  PMG_REQUIRE(out && layouts && nlevels >= 1, "pmg_multigrid_create: bad argument");
  PMG_REQUIRE(bc_marker_coarsest, "pmg_multigrid_create: NULL bc marker");
  auto* mg = new pmg_multigrid_s;
  HandleGuard<pmg_multigrid> guard(mg, pmg_multigrid_destroy);
  mg->L = nlevels;
  mg->layouts.assign(layouts, layouts + nlevels);
  mg->bc0 = bc_marker_coarsest;
  mg->u.assign(nlevels, nullptr);
  mg->b.assign(nlevels, nullptr);
  mg->counts.assign(nlevels, 0);
  for (int i = 0; i < nlevels - 1; ++i) // src/pmg.hpp:35-41 (finest level: caller's vectors)
  {
    PMG_TRY(alloc_vec(layouts[i], &mg->u[i]));
    PMG_TRY(alloc_vec(layouts[i], &mg->b[i]));
  }
  *out = guard.release();
  return PMG_OK;
}

extern "C" int pmg_multigrid_set_coarse_solver(pmg_multigrid mg, pmg_cg coarse)
{
  PMG_REQUIRE(mg, "pmg_multigrid_set_coarse_solver: NULL argument");
  PMG_REQUIRE(!coarse || coarse->layout == mg->layouts[0],
              "pmg_multigrid_set_coarse_solver: the solver is not on the coarsest layout");
  drop_graphs(mg);
  mg->coarse = coarse;
  mg->coarse_fn = nullptr;
  mg->coarse_amg = nullptr;
  return PMG_OK;
}

extern "C" int pmg_multigrid_set_coarse_amg(pmg_multigrid mg, pmg_amg amg)
{
  PMG_REQUIRE(mg, "pmg_multigrid_set_coarse_amg: NULL argument");
  PMG_REQUIRE(!amg || amg_layout(amg) == mg->layouts[0],
              "pmg_multigrid_set_coarse_amg: the solver is not on the coarsest layout");
  drop_graphs(mg);
  mg->coarse_amg = amg;
  if (amg)
  {
    mg->coarse = nullptr;
    mg->coarse_fn = nullptr;
  }
  return PMG_OK;
}

extern "C" int pmg_multigrid_set_coarse_callback(pmg_multigrid mg, pmg_coarse_solve_fn solve, void* user)
{
  PMG_REQUIRE(mg, "pmg_multigrid_set_coarse_callback: NULL argument");
  drop_graphs(mg);
  mg->coarse_fn = solve;
  mg->coarse_user = user;
  if (solve)
  {
    mg->coarse = nullptr;
    mg->coarse_amg = nullptr;
  }
  return PMG_OK;
}

extern "C" int pmg_multigrid_destroy(pmg_multigrid mg)
{
  if (!mg)
    return PMG_OK;
  for (int i = 0; i < mg->L - 1; ++i)
  {
    (void)hipFree(mg->u[i]);
    (void)hipFree(mg->b[i]);
  }
  for (auto& g : mg->graphs)
    (void)hipGraphExecDestroy(g.exec);
  if (mg->capture_stream)
    (void)hipStreamDestroy(mg->capture_stream);
  delete mg;
  return PMG_OK;
}

extern "C" int pmg_multigrid_set_operators(pmg_multigrid mg, const pmg_laplacian* ops)
{
  PMG_REQUIRE(mg && ops, "pmg_multigrid_set_operators: NULL argument");
  for (int i = 0; i < mg->L; ++i)
    PMG_REQUIRE(ops[i] && laplacian_layout(ops[i]) == mg->layouts[i],
                "pmg_multigrid_set_operators: level %d layout mismatch", i);
  drop_graphs(mg);
  mg->ops.assign(ops, ops + mg->L);
  return PMG_OK;
}

extern "C" int pmg_multigrid_set_solvers(pmg_multigrid mg, const pmg_chebyshev* smoothers)
{
  PMG_REQUIRE(mg && smoothers, "pmg_multigrid_set_solvers: NULL argument");
  for (int i = 0; i < mg->L; ++i)
    PMG_REQUIRE(smoothers[i] && smoothers[i]->layout == mg->layouts[i],
                "pmg_multigrid_set_solvers: level %d layout mismatch", i);
  drop_graphs(mg);
  mg->smoothers.assign(smoothers, smoothers + mg->L);
  return PMG_OK;
}

extern "C" int pmg_multigrid_set_interpolators(pmg_multigrid mg, const pmg_interpolator* interp)
{
  PMG_REQUIRE(mg && (interp || mg->L == 1), "pmg_multigrid_set_interpolators: NULL argument");
  drop_graphs(mg);
  mg->interps.clear();
  for (int i = 0; i < mg->L - 1; ++i)
  {
    PMG_REQUIRE(interp[i], "pmg_multigrid_set_interpolators: level %d is NULL", i);
    mg->interps.push_back(interp[i]);
  }
  return PMG_OK;
}

namespace
{
void drop_graphs(pmg_multigrid mg)
{
  for (auto& g : mg->graphs)
    (void)hipGraphExecDestroy(g.exec);
  mg->graphs.clear();
}

// What a captured cycle depends on besides its two vectors; -1: this configuration cannot be
// captured (a host synchronisation or a host callback inside the cycle, in-situ timing events)
// Is the cycle replayed as a hipGraph?  Explicitly on / off (pmg_multigrid_set_graph), or by default: on one rank the
// launches are issued ahead of the GPU anyway and a replay buys nothing (5.40 against 5.42 ms at config 2); on several
// ranks an eager exchange costs the host more than the GPU (profiles/exchange_overhead_r02.txt; four ranks on one GPU,
// round 4: 8.4 ms eager against 6.3 ms replayed at 64^3 in total), so the cycle is captured BY DEFAULT wherever the
// capture holds nothing but kernels -- every exchange through halo windows, every reduction of the cycle through a
// communicator made of windows.  A cycle whose exchanges are RCCL calls is captured on request only: that capture
// has run on one GPU (a rank as its own partner), never between two, and a runtime that misbehaves there would take
// the caller's run down rather than fail a check (PMG_GRAPH_AUTO=rccl includes it in the default, =0 turns the
// default off).
bool use_graph(pmg_multigrid mg)
{
  if (mg->graph_mode >= 0)
    return mg->graph_mode == 1;
  const char* e = std::getenv("PMG_GRAPH_AUTO");
  if (e && e[0] == '0')
    return false;
  const bool rccl_too = e && std::string(e) == "rccl";
  bool several = false;
  for (int i = 0; i < mg->L; ++i)
  {
    pmg_layout l = mg->layouts[i];
    if (!l->multi_rank() && !l->win)
      continue;
    several = true;
    if (l->exchange && !l->comm && !l->win)
      return false; // callbacks: host code in the cycle
    if (!l->win && !rccl_too)
      return false; // grouped ncclSend / ncclRecv in the capture
    if (mg->coarse_amg && i == 0 && l->comm && !l->comm->wcomm && !rccl_too)
      return false; // the replicated coarse solve's ncclAllReduce in the capture
  }
  return several;
}

long long capture_config(pmg_multigrid mg)
{
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
  for (int i = 0; i < mg->L; ++i)
  {
    pmg_layout l = mg->layouts[i];
    // halo: the caller's callbacks are host code inside the cycle.  The library's grouped ncclSend / ncclRecv are
    // captured on the capture stream itself (comm_exchange_begin; forking to the communicator's stream inside a
    // capture crashes RCCL 2.26).
    if (l->exchange && !l->comm && !l->win)
      return -1;
    // first use of a peer connection / of the collective must not fall inside a capture: eager until every level's
    // layout has exchanged once (and, with the replicated AMG's all-reduce, the communicator has reduced once)
    if (!comm_capture_ready(l, i == 0 && mg->coarse_amg != nullptr))
      return -1;
    const long long st = laplacian_capture_state(mg->ops[i]);
    if (st < 0)
      return -1;
    mix((uint64_t)st);
    // identities: a graph replays the device pointers of the objects it was captured with (the setters drop the
    // cache as well; this covers an object replaced by another one at the same address only by accident)
    mix((uint64_t)(uintptr_t)mg->ops[i]);
    mix((uint64_t)(uintptr_t)mg->smoothers[i]);
    mix((uint64_t)(uintptr_t)l->comm);
    mix((uint64_t)(uintptr_t)l->win);
    if (i + 1 < mg->L)
      mix((uint64_t)(uintptr_t)mg->interps[i]);
    mix((uint64_t)mg->smoothers[i]->max_iter);
    uint64_t bits;
    static_assert(sizeof(bits) == sizeof(double), "");
    memcpy(&bits, &mg->smoothers[i]->eig_max, sizeof(bits));
    mix(bits);
  }
  if (mg->coarse || mg->coarse_fn) // Krylov coarse solve / caller's solver: host in the loop
    return -1;
  if (mg->coarse_amg)
  {
    const long long st = amg_capture_state(mg->coarse_amg);
    if (st < 0)
      return -1;
    mix((uint64_t)st);
    mix((uint64_t)(uintptr_t)mg->coarse_amg);
  }
  return (long long)(h >> 1);
}

// mg_apply through a graph: captured on the library's own stream the first time a
// (vectors, configuration) combination is seen, replayed on the caller's stream afterwards
int mg_apply_graph(pmg_multigrid mg, const double* rhs, double* y, bool y_zero, hipStream_t s, bool* done)
{
  *done = false;
  if ((int)mg->ops.size() != mg->L || (int)mg->smoothers.size() != mg->L || (int)mg->interps.size() != mg->L - 1)
    return PMG_OK; // mg_apply reports it
  const long long cfg = capture_config(mg);
  if (cfg < 0)
    return PMG_OK;
  for (auto& g : mg->graphs)
    if (g.rhs == rhs && g.y == y && g.y_zero == y_zero && g.config == (uint64_t)cfg)
    {
      PMG_HIP(hipGraphLaunch(g.exec, s));
      mg->counts = g.counts;
      mg->graph_replays++;
      *done = true;
      return PMG_OK;
    }
  if (!mg->capture_stream)
    PMG_HIP(hipStreamCreateWithFlags(&mg->capture_stream, hipStreamNonBlocking));
  PMG_HIP(hipStreamBeginCapture(mg->capture_stream, hipStreamCaptureModeRelaxed));
  const int rc = mg_apply(mg, rhs, y, y_zero, mg->capture_stream);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(mg->capture_stream, &graph);
  if (rc != PMG_OK)
  {
    if (graph)
      (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess || !graph)
    return fail(PMG_ERR_HIP, "capturing the V-cycle failed: %s", hipGetErrorString(e));
  pmg_multigrid_s::GraphEntry g{rhs, y, y_zero, (uint64_t)cfg, nullptr, mg->counts};
  const hipError_t ei = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess)
    return fail(PMG_ERR_HIP, "instantiating the V-cycle graph failed: %s", hipGetErrorString(ei));
  if (mg->graphs.size() >= 16) // callers that cycle through many vectors: keep the cache small
    drop_graphs(mg);
  mg->graphs.push_back(g);
  PMG_HIP(hipGraphLaunch(g.exec, s));
  mg->graph_replays++;
  *done = true;
  return PMG_OK;
}
} // namespace

extern "C" int pmg_multigrid_set_graph(pmg_multigrid mg, int enable)
{
  PMG_REQUIRE(mg, "pmg_multigrid_set_graph: NULL argument");
  drop_graphs(mg);
  mg->graph_mode = enable < 0 ? -1 : (enable != 0 ? 1 : 0);
  return PMG_OK;
}

extern "C" long long pmg_multigrid_graph_replays(pmg_multigrid mg) { return mg ? mg->graph_replays : -1; }

extern "C" int pmg_multigrid_apply(pmg_multigrid mg, const double* rhs, double* y, double* rnorm,
                                   pmg_stream stream)
{
  PMG_REQUIRE(mg && rhs && y, "pmg_multigrid_apply: NULL argument");
  hipStream_t s = S(stream);
  bool done = false;
  if (use_graph(mg))
    PMG_TRY(mg_apply_graph(mg, rhs, y, false, s, &done));
  if (!done)
    PMG_TRY(mg_apply(mg, rhs, y, false, s));
  if (rnorm) // src/pmg.hpp:141-150
  {
    const int L = mg->L;
    pmg_chebyshev sm = mg->smoothers[L - 1];
    PMG_TRY(laplacian_apply(mg->ops[L - 1], y, sm->q, s));
    launch_axpy(mg->layouts[L - 1]->size_local, sm->r, -1.0, sm->q, rhs, s);
    double v;
    PMG_TRY(dot_host(mg->layouts[L - 1], sm->r, sm->r, &v, s));
    *rnorm = std::sqrt(v);
  }
  return PMG_OK;
}

extern "C" int pmg_multigrid_apply_counts(pmg_multigrid mg, int* counts, int capacity)
{
  PMG_REQUIRE(mg && counts, "pmg_multigrid_apply_counts: NULL argument");
  for (int i = 0; i < mg->L && i < capacity; ++i)
    counts[i] = mg->counts[i];
  return mg->L;
}
