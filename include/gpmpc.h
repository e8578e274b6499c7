/*
 * gpmpc.h -- C ABI of the MI355X-native GP-MPC rollout library (libgpmpc_hip.so).
 *
 * Drop-in boundary for the hot path of Thiagodcv/gaussian-process-mpc.  The
 * reference has no FFI of its own (it is pure Python on torch); the entry
 * points below are what a binding for this path has to call, and each one
 * names the reference interface it replaces (paths relative to the upstream
 * repository root).  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - All matrices are row-major fp64.  "dev" = device (HBM) pointer, "host" =
 *    host pointer.  The caller owns every buffer; the library never frees or
 *    reallocates caller memory and keeps no global state besides the pack (exceptions: the opt-in timing
 *    counters of gpmpc_timing_enable, process-wide behind a mutex; GPMPC_* tuning environment variables, read once
 *    per pack at gpmpc_pack_create / gpmpc_pack_reload_tuning, never on the per-call path).
 *  - A pack lives on the HIP device that was current when it was created; every entry point that takes a pack
 *    returns GPMPC_E_DEVICE if the calling thread's current device differs (it never switches devices itself).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *    All work is enqueued asynchronously on it; nothing synchronises the
 *    device.  Calls are re-entrant across streams and host threads as long as workspaces differ: a pack is only read by a
 *    rollout, and what a pack OWNS -- its private streams / events (graph replay, the concurrent sub-batches of a mid-size
 *    batch), the captured graphs and the staging buffers of gpmpc_objective_gradient -- is guarded by a per-pack host lock
 *    (lazy creation, a stream capture from begin to end, a fork / join of a split launch, the callback entry are each one
 *    critical section).  Functions that MODIFY a pack (build, resize, enable_fullcov, reload_tuning, autotune, destroy) must
 *    not run concurrently with anything else on that pack.
 *    HOST arrays (hyper-parameters, cost parameters) are consumed before the call
 *    returns -- they travel as kernel arguments --: the caller may free or overwrite
 *    them at once, whatever `stream` is waiting for.  DEVICE buffers must stay valid
 *    until the work enqueued on `stream` has run.
 *  - Return value: 0 on success, a negative GPMPC_E_* code otherwise.  No
 *    exception crosses the ABI.  NaNs produced by the arithmetic (negative
 *    variances, log of a non-positive determinant) are passed through, as in
 *    the reference.
 *  - Dimensions: D = ds + da <= GPMPC_MAX_D, ds <= GPMPC_MAX_DS.
 */
#ifndef GPMPC_H
#define GPMPC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPMPC_MAX_D   8
#define GPMPC_MAX_DS  8

#define GPMPC_OK            0
#define GPMPC_E_ARG        -1   /* bad argument (NULL pointer, dimension out of range) */
#define GPMPC_E_ALLOC      -2   /* device allocation failed */
#define GPMPC_E_LAUNCH     -3   /* a HIP call or kernel launch failed */
#define GPMPC_E_WORKSPACE  -4   /* workspace too small */
#define GPMPC_E_STATE      -5   /* pack not built */
#define GPMPC_E_DEVICE     -6   /* the calling thread's current HIP device is not the one the pack was created on */

/* flags for gpmpc_rollout / gpmpc_moment_match */
#define GPMPC_WANT_GRAD      1u   /* also produce d cost / d U (rollout) or input Jacobians (moment_match) */
#define GPMPC_USE_GRAPH      4u   /* gpmpc_rollout: replay the launches of the rollout as one hipGraph.  The caller promises
                                     that the pointer arguments (inputs, outputs, workspace) are the same buffers on every
                                     call of one shape with this flag; a changed argument captures anew (4 shapes are kept
                                     per pack).  For launch-latency-bound small batches (B = 1 solver loops). */
#define GPMPC_FP32_ACCUM     8u   /* gpmpc_rollout, objective only: N^2 products and their sum in fp32 (exponent / exp in fp64) */
#define GPMPC_FP32_ALL      16u   /* gpmpc_rollout, objective only: transformed points, exponent, exp and sum in fp32.
                                     Both exist for the fp64-vs-fp32 tolerance sweep of BASELINE config 3: the variance is a
                                     cancelling sum and single precision FAILS the 1e-4 tolerance (profiles/r01/fp32_sweep.txt) */
#define GPMPC_COV_BUG_COMPAT 2u   /* cross-covariance with the reference's transposed cross term
                                     (src/tools/uncertainty_prop.py:446) instead of the consistent one */

typedef struct gpmpc_pack gpmpc_pack;   /* opaque: device-resident GP state */

/* Library / device probe.  Returns the number of visible HIP devices (>= 0) or a negative code. */
int gpmpc_device_count(void);
const char* gpmpc_version(void);
const char* gpmpc_last_error(void);     /* thread-local text of the last failing HIP call */

/* ---------------------------------------------------------------------------
 * GP state ("pack").  Replaces the per-call constant work of
 *   mean_prop_torch      src/tools/uncertainty_prop.py:324-327 (beta = Ky_inv @ y)
 *   variance_prop_torch  src/tools/uncertainty_prop.py:392-399 (Lambda_part, Ky_inv - beta beta^T)
 * and holds what Dynamics/GaussianProcessRegression hold on the device
 * (src/dynamics.py:33-37, src/gpr.py:24-36): X_train shared by all ds GPs,
 * per GP: beta, lambdas, sigma_f and the folded weight matrix
 *   M_a[i][j] = sym(Ky_inv_a - beta_a beta_a^T)[i][j] * sigma_f_a^4
 *               * exp(-1/4 (x_i-x_j)^T Lambda_a^-1 (x_i-x_j))
 * stored upper-triangular with off-diagonal weight 2.
 * ------------------------------------------------------------------------- */
int gpmpc_pack_create(gpmpc_pack** out, int n_train, int state_dim, int action_dim);
int gpmpc_pack_destroy(gpmpc_pack* pack);
/* Re-use the pack for n_train points when the padded size (multiple of 64) is unchanged -- GPMPC_E_ARG otherwise --; the pack
 * is "not built" until the next gpmpc_pack_build*; captured launch sequences (GPMPC_USE_GRAPH, gpmpc_objective_gradient) stay
 * valid (no rollout launch depends on the unpadded size).  For the closed loop, where Dynamics.append_train_data adds one
 * observation per step (src/simulator.py:55): no allocation per step. */
int gpmpc_pack_resize(gpmpc_pack* pack, int n_train);
/* Re-read the GPMPC_* tuning environment variables for this pack (they are otherwise read once, at
 * gpmpc_pack_create) and drop its captured graph.  For A/B runs and tests; no reference counterpart. */
int gpmpc_pack_reload_tuning(gpmpc_pack* pack);
/* Number of hipGraph captures gpmpc_rollout(GPMPC_USE_GRAPH) has done for this pack (up to 4 captured call shapes are
 * kept per pack, least recently used replaced).  Diagnostic; no reference counterpart. */
long long gpmpc_pack_graph_captures(const gpmpc_pack* pack);

/* A few host values (<= 512 bytes, a multiple of 4) -> device memory, ordered on `stream`, consumed before the call returns (they travel
 * as kernel arguments: no pageable-memory copy, no synchronisation).  The closed loop appends ONE observation per step
 * (src/simulator.py:55 -> src/gpr.py:109-119): the new input row and targets go straight into capacity-padded device buffers. */
int gpmpc_store_host(void* dst_dev, const void* src_host, size_t bytes, void* stream);

/* Build K_f, K_y = K_f + noise_var*I for one GP on the device
 * (GaussianProcessRegression.build_Ky_inv_mat, src/gpr.py:163-170; the inverse at :171 is
 * taken by the caller).  X dev [n][D]; lambdas host [D]; Kf, Ky dev [n][n] (Kf may be NULL).
 * noise_var is the value added on the diagonal (the reference adds float32(sigma_n^2)). */
int gpmpc_build_ky(int n, int D, const double* X_dev, const double* lambdas_host,
                   double sigma_f, double noise_var, double* Kf_dev, double* Ky_dev, void* stream);

/* Fill the pack.  X dev [N][D]; Y dev [N][ds] (column a = targets of GP a, as
 * Dynamics.append_train_data stores them, src/dynamics.py:51-60); Ky_inv dev [ds][N][N]
 * (src/gpr.py:171); lambdas host [ds][D]; sigma_f host [ds]. */
int gpmpc_pack_build(gpmpc_pack* pack, const double* X_dev, const double* Y_dev,
                     const double* Ky_inv_dev, const double* lambdas_host,
                     const double* sigma_f_host, void* stream);

/* Same for inverses that are NOT packed [ds][N][N]: row stride `ld` (>= N) and the stride `gp_stride` from one GP's matrix to
 * the next, both in doubles.  gp_stride = 0: ONE matrix for every GP (GPs fed the same inputs under identical hyper-parameters
 * have identical Ky_inv -- every experiment of the reference, src/experiments/pretrain_uncertainty.py:100-105 -- and a capacity-padded
 * buffer that grows by one observation per Simulator step, src/simulator.py:55, is read in place instead of being copied). */
int gpmpc_pack_build_strided(gpmpc_pack* pack, const double* X_dev, const double* Y_dev, const double* Ky_inv_dev,
                             size_t ld, size_t gp_stride, const double* lambdas_host, const double* sigma_f_host, void* stream);

/* Same, with beta given instead of the targets: beta dev [N][ds] (column a = beta_a), as the callers of
 * variance_prop_torch / covariance_prop_torch hold it (src/tools/uncertainty_prop.py:341, :402).
 * Ky_inv may be NULL: the weight matrices are then zero and only means and cross-covariances
 * (which need beta alone) are meaningful. */
int gpmpc_pack_build_beta(gpmpc_pack* pack, const double* X_dev, const double* beta_dev,
                          const double* Ky_inv_dev, const double* lambdas_host,
                          const double* sigma_f_host, void* stream);

/* Allocate and maintain the cross-covariance weight matrices (one N x N matrix per GP pair a < b): needed by
 * gpmpc_rollout_fullcov and by the analytic cross-covariance Jacobians of gpmpc_moment_match.  Without it
 * cross-covariances are evaluated by a direct N^2 kernel, forward only. */
int gpmpc_pack_enable_fullcov(gpmpc_pack* pack, void* stream);

/* Inspection for tests / bindings.  gpmpc_pack_export copies into caller buffers (either may be NULL):
 * beta_out dev [ds][n_padded] (beta_a = Ky_inv_a y_a, zero padded); weights_out dev
 * [ds][n_padded][n_padded], element (i <= j) of M_a at [a][j][i], zero elsewhere. */
int gpmpc_pack_dims(const gpmpc_pack* pack, int* n_train, int* n_padded, int* state_dim, int* action_dim);
/* 1 if the last gpmpc_pack_build* found bit-identical length-scales for every GP (the setting of all of the reference's
 * experiments, e.g. src/experiments/pretrain_uncertainty.py:100-105): gpmpc_rollout then evaluates exponent and exp once
 * per pair for a group of GPs.  0 otherwise, negative on error.  Diagnostic; no reference counterpart. */
int gpmpc_pack_shared_lambda(const gpmpc_pack* pack);
int gpmpc_pack_export(const gpmpc_pack* pack, double* beta_out, double* weights_out, void* stream);

/* ---------------------------------------------------------------------------
 * Single-step exact moment matching for nq Gaussian inputs N(u_q, S_q), all ds GPs.
 * Replaces mean_prop_torch / variance_prop_torch / covariance_prop_torch
 * (src/tools/uncertainty_prop.py:296-338, :341-399, :402-465).  S may be a full
 * symmetric positive-definite matrix.
 *   u dev [nq][D], S dev [nq][D][D]
 *   out_mean dev [nq][ds]; out_var dev [nq][ds]
 *   out_cov  dev [nq][ds][ds] or NULL: full predictive covariance (diagonal = out_var,
 *            off-diagonal = cross-covariances; flag GPMPC_COV_BUG_COMPAT selects the
 *            reference's transposed cross term)
 *   out_l    dev [nq][ds][N] or NULL: the vector l of mean_prop_torch's second return value
 *            (l_i = c_m exp(-1/2 v_i^T B v_i), src/tools/uncertainty_prop.py:335-336)
 *   with GPMPC_WANT_GRAD (the first four non-NULL):
 *     dmean_du dev [nq][ds][D], dmean_dS dev [nq][ds][D][D] (symmetrised),
 *     dvar_du  dev [nq][ds][D], dvar_dS  dev [nq][ds][D][D] (symmetrised),
 *     dcov_du  dev [nq][ds][ds][D], dcov_dS dev [nq][ds][ds][D][D] or both NULL: Jacobians of the full covariance
 *              (consistent form only; needs gpmpc_pack_enable_fullcov, else GPMPC_E_STATE)
 * ------------------------------------------------------------------------- */
size_t gpmpc_moment_match_workspace_bytes(const gpmpc_pack* pack, int nq);
int gpmpc_moment_match(const gpmpc_pack* pack, int nq, const double* u_dev, const double* S_dev,
                       unsigned flags, double* out_mean, double* out_var, double* out_cov, double* out_l,
                       double* dmean_du, double* dmean_dS, double* dvar_du, double* dvar_dS,
                       double* dcov_du, double* dcov_dS,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Risk-sensitive cost (RiskSensitiveMPC.cost_torch, src/mpc.py:156-200).
 * gamma == 0 selects the risk-neutral limit tr(Q Sigma) + e^T Q e (the reference
 * divides by gamma and cannot evaluate it).
 * ------------------------------------------------------------------------- */
typedef struct gpmpc_cost_params {
    double gamma;
    double Q[GPMPC_MAX_DS * GPMPC_MAX_DS];            /* [ds][ds] row-major in the leading ds*ds entries */
    double R[GPMPC_MAX_D * GPMPC_MAX_D];              /* [da][da] */
    double R_delta[GPMPC_MAX_D * GPMPC_MAX_D];        /* [da][da], used when has_R_delta != 0 */
    double x_ref[GPMPC_MAX_DS];
    double u_ref[GPMPC_MAX_D];
    double last_u[GPMPC_MAX_D];                       /* last_traj[0:da], src/mpc.py:192 */
    int has_R_delta;
    int reserved;
} gpmpc_cost_params;

/* Cost of B given trajectories with FULL covariance matrices (parity with cost_torch on
 * arbitrary, even non-symmetric, Sigma): means dev [B][H+1][ds], covs dev [B][H+1][ds][ds],
 * U dev [B][H][da] -> out_cost dev [B]. */
int gpmpc_cost(int B, int H, int state_dim, int action_dim, const gpmpc_cost_params* cost_host,
               const double* means_dev, const double* covs_dev, const double* U_dev,
               double* out_cost, void* stream);

/* The same cost with its derivatives: what `curr_cost.backward()` (src/mpc.py:251) leaves in the graph of
 * cost_torch (src/mpc.py:179-198) -- d_means dev [B][H+1][ds], d_covs dev [B][H+1][ds][ds] (element [k][l] =
 * d cost / d Sigma_kl of a general, possibly non-symmetric Sigma), d_U dev [B][H][da] (input and input-rate terms).
 * The three derivative outputs are given together or all NULL. */
int gpmpc_cost_grad(int B, int H, int state_dim, int action_dim, const gpmpc_cost_params* cost_host,
                    const double* means_dev, const double* covs_dev, const double* U_dev,
                    double* out_cost, double* d_means, double* d_covs, double* d_U, void* stream);

/* ---------------------------------------------------------------------------
 * The hot path: B independent shooting rollouts + cost + gradient.
 * Replaces, for each trajectory b,
 *   Dynamics.forward_propagate_torch(H, x0[b], U[b])      src/dynamics.py:126-191
 *   RiskSensitiveMPC.cost_torch(...)                        src/mpc.py:156-200
 *   RiskSensitiveMPC.objective(x) / gradient(x)             src/mpc.py:202-255
 * (the reference handles one trajectory per call; B > 1 is the batched form).
 * Semantics kept: Sigma_0 = 1e-3 I (float64), action-noise variance float32(1e-3),
 * diagonal covariance propagation, no clamp of negative variances.
 *   x0 dev [B][ds]; U dev [B][H][da]
 *   out_means dev [B][H+1][ds]; out_vars dev [B][H+1][ds]  (either may be NULL)
 *   out_cost dev [B]; out_grad dev [B][H][da] (required with GPMPC_WANT_GRAD)
 * Mid-size batches (a few to a few dozen trajectories) are run as 2-4 concurrent sub-batches (2 when launched plainly) on
 * streams owned by the pack, forked from and joined back into `stream` with events (parallel branches of the graph under
 * GPMPC_USE_GRAPH): the caller still sees ONE call ordered on ONE stream, results are bit-identical to the unsplit launch,
 * and the workspace size reported below covers the sub-batches' slices.  Small and mid-size batches (up to ~4700 tile
 * workgroups per horizon step) are one kernel launch per horizon step; larger ones two (head + pair kernel).
 * ------------------------------------------------------------------------- */
size_t gpmpc_rollout_workspace_bytes(const gpmpc_pack* pack, int B, int H, unsigned flags);
/* Diagnostic (no reference counterpart): what a rollout call of this shape launches, as one line of text --
 * "form=<fused_staged|fused_sb|fused_sb_shared|head+pair_sb|head+pair_sbs|head+pair_staged|lowprec> kernel=<dominant kernel instance>
 *  tiling=<rows>x<cols> workgroups=<per horizon step> launches_per_step=<1|2> split=<concurrent sub-batches> ..." --
 * with GPMPC_USE_GRAPH in `flags` for the split a graph replay would use.  bench.py names its dominant kernel with it, the
 * parity tests check which kernel form a shape reaches.  out_bytes >= 64; returns 0 or GPMPC_E_ARG. */
int gpmpc_plan_describe(const gpmpc_pack* pack, int B, int H, unsigned flags, char* out, size_t out_bytes);
/* Plan selection that MEASURES (no reference counterpart).  The kernel form of a rollout call -- tiling, one or two launches per
 * horizon step or the whole-horizon kernel, trajectories per wave, concurrent sub-batches -- is chosen from thresholds measured on
 * one MI355X; this call times the candidate plans of ONE call shape (B, H, objective-only or with gradient; GPMPC_USE_GRAPH in
 * `flags`: as graph replays, else as plain launches) on THIS device with the pack's own data -- each candidate a warm-up and the best
 * of three timed blocks, on scratch buffers of its own -- and makes the pack remember the fastest (up to 16 shapes; the default plan
 * stays unless beaten by more than 2 %).  Every plan sums in a fixed order: results stay bit-reproducible per plan, and differ between
 * plans by rounding only.  Synchronous (tens of milliseconds); not to be called while other host threads use the pack.
 * `report` (optional): "name:fused=..,tiling=..,...:ms;..." per candidate, the winner marked with '*', the default first.
 * Returns the number of candidates timed (> 0) or a negative code.  gpmpc_pack_autotune_clear forgets every measured plan; so do
 * gpmpc_pack_reload_tuning and a gpmpc_pack_build that changes the "all GPs share their length-scales" property. */
int gpmpc_pack_autotune(gpmpc_pack* pack, int B, int H, unsigned flags, char* report, size_t report_bytes);
int gpmpc_pack_autotune_clear(gpmpc_pack* pack);
int gpmpc_rollout(const gpmpc_pack* pack, int B, int H, const double* x0_dev, const double* U_dev,
                  const gpmpc_cost_params* cost_host, unsigned flags,
                  double* out_means, double* out_vars, double* out_cost, double* out_grad,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Differentiable propagation: Dynamics.forward_propagate_torch (src/dynamics.py:126-191) returns tensors with the autograd
 * graph attached, and RiskSensitiveMPC.gradient back-propagates through it (src/mpc.py:218, :251).  The two entry points
 * below are the forward and the backward of that graph:
 *   gpmpc_rollout_jac   means / variances of B rollouts (no cost) + the step Jacobians
 *                       out_jac dev [B][H][2ds][2ds+da]: rows (mu_t, var_t), columns (mu_{t-1}, var_{t-1}, u_{t-1})
 *   gpmpc_rollout_vjp   g_means, g_vars dev [B][H+1][ds] (upstream gradients of every step's mean / variance; either
 *                       may be NULL = zero) -> out_gU dev [B][H][da], out_gx0 dev [B][ds] or NULL.
 * ------------------------------------------------------------------------- */
size_t gpmpc_rollout_jac_workspace_bytes(const gpmpc_pack* pack, int B, int H);
int gpmpc_rollout_jac(const gpmpc_pack* pack, int B, int H, const double* x0_dev, const double* U_dev,
                      double* out_means, double* out_vars, double* out_jac,
                      void* workspace, size_t workspace_bytes, void* stream);
int gpmpc_rollout_vjp(int B, int H, int state_dim, int action_dim, const double* jac_dev,
                      const double* g_means, const double* g_vars, double* out_gU, double* out_gx0, void* stream);

/* ---------------------------------------------------------------------------
 * Full-covariance form of the hot path (BASELINE config 5): the state distribution carries the whole ds x ds
 * covariance; off-diagonal terms are the exact cross-covariances Cov[f_a, f_b] (covariance_prop_torch,
 * src/tools/uncertainty_prop.py:402-465, consistent form).  The reference's rollout propagates variances only
 * (src/dynamics.py:184-189 TODO); this is its extension to full Sigma with the same cost (src/mpc.py:156-200) and an
 * analytic gradient.  Needs gpmpc_pack_enable_fullcov (GPMPC_E_STATE otherwise).
 *   out_means dev [B][H+1][ds]; out_covs dev [B][H+1][ds][ds] (both required);
 *   out_cost dev [B]; out_grad dev [B][H][da] (with GPMPC_WANT_GRAD).
 * ------------------------------------------------------------------------- */
size_t gpmpc_rollout_fullcov_workspace_bytes(const gpmpc_pack* pack, int B, int H, unsigned flags);
int gpmpc_rollout_fullcov(const gpmpc_pack* pack, int B, int H, const double* x0_dev, const double* U_dev,
                          const gpmpc_cost_params* cost_host, unsigned flags,
                          double* out_means, double* out_covs, double* out_cost, double* out_grad,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Diagnostic (no reference counterpart): what gpmpc_rollout_fullcov launches for this call shape, as one line of text --
 * "form=<four_launch|two_launch> tiling=<rows>x<cols> workgroups=<of the pair kernel, per horizon step> columns_per_iteration=<1|2|4>
 *  head_workgroups_per_unit=<n> kernel=<pair kernel>".  Small batches run TWO launches per horizon step (round 4; fullcov.hip): a head
 * kernel that closes the previous step, assembles (u_t, S_t) and prepares every unit, and the pair kernel on narrow tiles; large batches
 * four (assemble, prepare, pair kernel on 256x256 tiles, close).  out_bytes >= 64; returns 0 or GPMPC_E_ARG. */
int gpmpc_rollout_fullcov_describe(const gpmpc_pack* pack, int B, int H, unsigned flags, char* out, size_t out_bytes);

/* The solver callback pair RiskSensitiveMPC.objective(x) / gradient(x) (src/mpc.py:202-255) for ONE candidate, host in and
 * host out like the cyipopt callbacks themselves: x0_host [ds] current state, U_host [H][da] the candidate (Ipopt's x),
 * out_host [1 + H da] = cost, then d cost / d U row-major (flags = GPMPC_WANT_GRAD; 0: cost only, out_host [1]).
 * SYNCHRONOUS: returns when out_host is filled.  The pack owns the staging buffers (pinned host + device) and one
 * captured hipGraph -- upload, the H + 1 kernels of the diagonal-covariance rollout, download -- so a callback costs
 * the host one graph launch and one stream wait; a changed horizon / cost parameter set re-captures.  `stream`: the
 * stream the pack was last built on (the call is ordered behind it).  Not re-entrant per pack. */
int gpmpc_objective_gradient(gpmpc_pack* pack, int H, const double* x0_host, const double* U_host,
                             const gpmpc_cost_params* cost_host, unsigned flags, double* out_host, void* stream);

/* Kernel-level timing of the dominant (pair) kernel for bench.py: when enabled, every
 * gpmpc_rollout brackets its pair-kernel launches with HIP events on the launch stream.
 * gpmpc_pair_kernel_time returns accumulated milliseconds and launch count since the last reset
 * (it synchronises on the recorded events). */
int gpmpc_timing_enable(int on);
int gpmpc_pair_kernel_time(double* total_ms, long long* launches, int reset);
/* The same totals per kernel class: 0 = the full pair kernel, 1 = its horizon-step-1 variant (constant state inputs:
 * fewer moments, cheaper), 2 = the fused small-batch step kernel (one launch per horizon step: mean sums, finish work and
 * pair tiles together -- NOT a pair-only time), so that a roofline figure can be quoted for the dominant kernel alone.  Reset with
 * gpmpc_pair_kernel_time(..., 1). */
int gpmpc_pair_kernel_time_class(int kernel_class, double* total_ms, long long* launches);

/* Launch geometry, host-side views (no device work; for tests without a GPU).
 * gpmpc_debug_run_list: the balanced-run work list of the one-launch form for ONE trajectory of a large training set (items
 *   {GP, first row, first column, end column}; n_padded = N rounded up to 64; slots = workgroup slots of the device minus the
 *   2 state_dim role workgroups).  *n_items = 0 when the plain 256x64 list is within 1.1 generations.  items_out holds 4 * capacity ints
 *   (capacity = 0: count only).
 * gpmpc_debug_xcd_order: (trajectory, grid column) that workgroup `linear_id` of a (grid_x, n_traj) grid with n_tile tile columns runs in
 *   the XCD-aware dispatch order. */
int gpmpc_debug_run_list(int n_padded, int state_dim, int slots, int* items_out, int capacity, int* n_items);
int gpmpc_debug_xcd_order(int linear_id, int grid_x, int n_tile, int n_traj, int* traj, int* column);

/* ---------------------------------------------------------------------------
 * GP prediction at test points (GaussianProcessRegression.compute_pred_train_covariance /
 * predict_latent_vars, src/gpr.py:253-332) for ONE GP given by raw arrays.
 *   X dev [n][D]; lambdas host [D]; beta dev [n] = Ky_inv (y - f_nom(X)) (needed for out_mean);
 *   Ky_inv dev [n][n] (needed for out_cov); X_pred dev [p][D];
 *   out_K dev [p][n] or NULL; out_mean dev [p] or NULL (nominal-model term added by the caller);
 *   out_cov dev [p][p] or NULL = K** - K* Ky_inv K*^T + noise_var * I.
 * gpmpc_matvec: out[r] = A[r] . v for a row-major [rows][cols] matrix (beta = Ky_inv y,
 * src/tools/uncertainty_prop.py:327).
 * ------------------------------------------------------------------------- */
int gpmpc_matvec(int rows, int cols, const double* A_dev, const double* v_dev, double* out_dev, void* stream);
size_t gpmpc_predict_workspace_bytes(int n, int D, int p);
int gpmpc_predict(int n, int D, const double* X_dev, const double* lambdas_host, double sigma_f,
                  const double* beta_dev, const double* Ky_inv_dev, double noise_var,
                  int p, const double* X_pred_dev, double* out_K, double* out_mean, double* out_cov,
                  void* workspace, size_t workspace_bytes, void* stream);

/* O(N^2) append of ONE observation to an explicit inverse (Schur complement; the reference's abandoned
 * update_Ky_inv_mat, src/gpr.py:137-157) instead of the O(N^3) rebuild of src/gpr.py:171 after every
 * Simulator step (src/simulator.py:55).  Ky_inv dev [n][n]; k dev [n] = K_f(X, x_new); kappa = sigma_f^2 + noise_var;
 * out dev [(n+1)][(n+1)] (must not alias Ky_inv).  Opt-in: results differ from a rebuild by rounding (~cond * eps). */
size_t gpmpc_kinv_append_workspace_bytes(int n);
int gpmpc_kinv_append(int n, const double* Ky_inv_dev, const double* k_dev, double kappa, double* out_dev,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The whole data update of ONE appended observation for a GP kept in CAPACITY-PADDED buffers (closed loop: src/simulator.py:55 ->
 * src/gpr.py:90-122 store the row, :159-171 rebuild Kf, Ky and invert): k = K_f(X, x_new) (src/gpr.py:124-135), then Kf, Ky and
 * Ky_inv of the n + 1 points -- the Schur step of gpmpc_kinv_append -- written into the OUTPUT buffers (row stride ld_out >= n + 1)
 * from the INPUT buffers (row strides ld_k_in of Kf / Ky and ld_in of Ky_inv, >= n).  Inputs and outputs must not alias: the caller ping-pongs two buffer sets, so the
 * n-point matrices stay valid for whoever still reads them.  X dev [n][D] (the n OLD rows), x_new dev [D], lambdas host [D].
 * Two kernel launches (k_append_vw2: k, v = Ky_inv k and the Schur scalar; k_append_fill2: the three (n + 1)-point matrices), no allocation,
 * no host-side concatenation. */
size_t gpmpc_gp_append_workspace_bytes(int n, int D);
int gpmpc_gp_append(int n, int D, const double* X_dev, const double* x_new_dev, const double* lambdas_host, double sigma_f,
                    double noise_var, const double* Kf_in, const double* Ky_in, size_t ld_k_in, const double* Ky_inv_in, size_t ld_in,
                    double* Kf_out, double* Ky_out, double* Ky_inv_out, size_t ld_out, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Gradient of the log marginal likelihood w.r.t. the LOG hyper-parameters in one pass over Ky_inv: replaces the autograd
 * backward through inv / det of update_hyperparams (src/gpr.py:334-338; likelihood src/gpr.py:240-251) and the dense
 * N x N x D derivative tensors of kernel_matrix_gradient / marginal_likelihood_grad (src/gpr.py:173-238).
 * X dev [n][D] row-major; Ky_inv dev [n][n]; alpha dev [n] = Ky_inv r; resid dev [n] = r = y - f_nom(X);
 * lambdas host [D]; noise_var = sigma_n^2 as it sits on the diagonal of Ky.
 * out dev [D+3]: d ml/d log lambda_k (D), d ml/d log sigma_f, d ml/d log sigma_n, and r^T alpha (the data-fit term of ml). */
size_t gpmpc_ml_grad_workspace_bytes(int n, int D);
int gpmpc_ml_grad(int n, int D, const double* X_dev, const double* Ky_inv_dev, const double* alpha_dev,
                  const double* resid_dev, const double* lambdas_host, double sigma_f, double noise_var,
                  double* out_dev, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPMPC_H */
