// Internal declarations shared by the HIP translation units of libgpmpc_hip.so.
// gfx950 (MI355X) only: 64-wide wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../include/gpmpc.h"

#define GPMPC_WAVE 64

// Variance of the initial state and of the action noise, as the reference has them
// (src/dynamics.py:148 float64 1e-3; src/dynamics.py:162 float32 1e-3 promoted to float64).
#define GPMPC_INIT_VAR   1e-3
#define GPMPC_ACTION_VAR ((double)1e-3f)

#define GPMPC_MAX_PAIRS (GPMPC_MAX_DS * (GPMPC_MAX_DS - 1) / 2)

// A list of pair-kernel work items (unit, i0, j0, j1).  Units 0..ds-1 are the variance units (upper-triangular
// tiles only), units ds.. are the cross-covariance units (a < b, all tiles).  Items of one unit are contiguous:
// unit u owns items [ustart[u], ustart[u+1]).
struct gpmpc_worklist {
    int it;         // rows per tile
    int waves;      // waves per workgroup: it / 64
    int jt;         // column extent of a tile (multiple of 64)
    int nunits, nwork;
    int contiguous;     // items of a unit are contiguous (ustart valid); 0: XCD-sorted order, look the unit up per item
    int* work_dev;      // [nwork][4]
    int* perm_dev;      // [nwork] item indices grouped by unit (XCD-sorted lists only, else null): unit u owns perm[ustart[u] .. ustart[u+1])
    int* ustart_dev;    // [nunits + 1]
    int ustart_host[GPMPC_MAX_DS + GPMPC_MAX_PAIRS + 1];
};

// Tuning overrides (GPMPC_* environment variables), read ONCE per pack -- at gpmpc_pack_create and again on
// gpmpc_pack_reload_tuning -- never on the per-call path (a solver loop issues thousands of B = 1 rollouts per second).
// -1 / 0 = not set: the measured defaults of plan_rollout / plan_mom apply.
struct gpmpc_tuning {
    int pair_sb;     // GPMPC_PAIR_SB     0 staged kernel | 1 scalar broadcast | -1 unset
    int tiling;      // GPMPC_TILING      0..6 | -1 unset
    int tb;          // GPMPC_PAIR_TB     1 | 2 | 4 | 0 unset
    int rgroup;      // GPMPC_RGROUP      1..16 | 0 unset
    int no_first;    // GPMPC_NO_FIRST    full moments at horizon step 1 too
    int fused;       // GPMPC_FUSED       0: head + staged pair kernel per step for small batches | -1 unset (fused step kernel)
    int no_xcd_sort; // GPMPC_NO_XCD_SORT natural tile order of the 256x256 work list (takes effect at pack creation only)
    int hchunks;     // GPMPC_HEAD_CHUNKS row chunks of the head kernel: 0 / 1 none | 2..16 | -1 unset (chosen per call)
    int sbf_min;     // GPMPC_SBF_MIN     workgroups from which the full-S path uses 256x256 tiles + pair_kernel_sbf.h | 0 unset
    int colunroll;   // GPMPC_SB_UNROLL   1 | 4 columns per iteration of the scalar-broadcast kernel on the 256x64 tiling | -1 unset (4)
    int fused_sb;    // GPMPC_FUSED_SB    one launch per horizon step for mid-size batches (step_fused.h, Q = 0): 0 off | 1 on | -1 unset
    int split;       // GPMPC_SPLIT       sub-batches (parallel graph branches) of a graph-replayed rollout: 1 none | 2..4 | -1 unset (2 for mid-size batches)
    int shared;      // GPMPC_SHARED      0: never use the shared-lambda kernel (pair_kernel_sbs.h) | -1 unset (used when lambdas are shared)
    int fc_form;     // GPMPC_FC_FORM     full-covariance rollout: 0 four launches per step | 1 two (fullcov.hip::k_fc_head) | -1 unset (by the size of the launch)
    int fc_rsplit;   // GPMPC_FC_RSPLIT   workgroups per (trajectory, unit) of the head kernel of the two-launch form | 0 unset
    int fc_cu;       // GPMPC_FC_CU       columns per iteration of the pair kernel there: 1 | 2 | 4 | 0 unset
    int fc_tiling;   // GPMPC_FC_TILING   pair-kernel tiles of the two-launch form: 0 256x256 | 2 256x64 | 4 256x128 | -1 unset
    int fc_shared;   // GPMPC_FC_SHARED   full-covariance rollout with one lambda for all GPs: cross units through pair_kernel_sbfx.h 0 off | 1 wherever it can run | -1 unset (by the size of the launch)
    int xcdmap;      // GPMPC_XCDMAP      XCD-aware dispatch order of the one-launch form for several trajectories (step_fused.h): 0 off | 1 on | -1 unset (by the size of the launch)
    int persist;     // GPMPC_PERSIST     whole-horizon kernel, one workgroup per trajectory (traj_persist.h): 0 off | 8 / 16 on with that many waves | -1 unset
};
void gpmpc_read_tuning(gpmpc_tuning* t);

struct gpmpc_pack {
    int N, Np, ds, da, D;
    int device;     // HIP device ordinal the pack's memory lives on (hipGetDevice at creation)
    int num_cu;     // compute units of that device
    gpmpc_tuning tune;
    int built;
    int npairs;     // ds (ds - 1) / 2 cross-covariance units (a < b, lexicographic)
    int fullcov;    // cross-covariance weight matrices are allocated and kept up to date
    int pair_a[GPMPC_MAX_PAIRS > 0 ? GPMPC_MAX_PAIRS : 1], pair_b[GPMPC_MAX_PAIRS > 0 ? GPMPC_MAX_PAIRS : 1];
    int* pair_ab_dev;   // [npairs][2]
    // one lambda for all GPs, full-covariance rollout (pair_kernel_sbfx.h): constant column rows [Np][fcs_rw] (refreshed by every build while
    // `fullcov`), and per tri-unit tiling k of wl[1][k] the unit offsets into a trajectory's partial-sum slots: tri units as in that work list,
    // cross unit pr at fcs_base[k] + pr * fcs_ntile
    double* fcs_rows; int fcs_rw, fcs_tj;
    // (two tile widths: 64 columns per wave for launches that fill the chip, 16 for small ones -- four times the waves, each a quarter as long)
    int fcs_ntile[3];                                  // tiles per trajectory: [0] 64 x 64, [1] 64 x 16, [2] 64 x (up to) 256 (listed in fcs_tiles256_dev)
    int* fcs_ustart_dev[8][3]; int fcs_base[8], fcs_total[8][3];
    int* fcs_tiles256_dev;                             // [fcs_ntile[2]][3] = {64-row block, first column, columns}: row block ti's columns from 64 ti on, in chunks of 256
    int ncol_host;      // the value last written to ncol_dev
    int* ncol_dev;      // [1] columns that carry weight: N rounded up to 8 (<= Np); written by every pack build (traj_persist.h reads it)
    double* X;      // dev [Np][D], rows >= N zero
    double* XT;     // dev [D][Np]
    double* beta;   // dev [ds][Np], zero padded
    // dev [ds (+ npairs)][Np][Np].  Variance unit a: element (i,j), i<=j, at [j*Np+i], weight 2 off the diagonal, zero
    // below.  Cross unit (a,b): element (i,j) at [j*Np+i] = beta_a[i] beta_b[j] sfa^2 sfb^2 exp(-1/2 d^2_{La+Lb}(x_i,x_j)).
    double* M;
    double* lam;    // dev [ds][D]
    double* sf;     // dev [ds]
    double lam_host[GPMPC_MAX_DS][GPMPC_MAX_D];
    double sf_host[GPMPC_MAX_DS];
    void* graph_cache;         // captured rollout (GPMPC_USE_GRAPH), owned by step.hip
    void* cb_cache;            // buffers + captured graph of gpmpc_objective_gradient (solver callbacks), owned by step.hip
    void* tuned;               // plans measured by gpmpc_pack_autotune (gpmpc_tuned_table, step.hip), or null
    void* lock;                // host lock of the pack's own streams / events / caches (std::recursive_mutex, step.hip::PackGuard)
    // [0: variance units only | 1: + cross units][0: 256x256 tiles | 1: 64x64 | 2: 256x64 | 3: 64x128 | 4: 256x128 | 5: 256x32 | 6: 256x16
    //  (4...6: mode 0 only)]
    gpmpc_worklist wl[2][8];   // [7] (diagonal rollout only, D >= 6): balanced runs of up to 256 columns, one workgroup generation per trajectory (pack.hip)
    // shared-lambda path (pair_kernel_sbs.h): every GP has bit-identical length-scales (detected at gpmpc_pack_build)
    int shared_lambda;
    int sh_ng;                 // GPs per workgroup
    gpmpc_worklist wl_sh[4];   // items {group, i0, j0, tile}: [0: 256x256 tiles, XCD-sorted | 1: 256x64 | 2: 256x128, XCD-sorted]; .nunits = groups
                               // [3: 256x64 in groups of TWO GPs (the one-launch form of smaller mid-size batches; built when sh_ng > 2 and ds is even)]
    int sh_tiles[3];           // tiles per GP
};

// Number of pair-kernel output moments per (trajectory, GP, tile).
static inline int gpmpc_num_moments(int D, bool diag, bool grad) {
    if (!grad) return 1;
    return diag ? 1 + 2 * D : 1 + D + D * (D + 1) / 2;
}

struct PairArgs {
    const double* M;      // [units][Np][Np]
    const double* XT;     // [D][Np]
    const double* pp;     // [B][nunits][pps]: row-side cvec[D] + transform, column side jside_off doubles further
    double* part;         // [B][nwork][nm]
    const int* work;      // [nwork][4] = {unit, i0, j0, j1}
    int Np, B, nunits, nwork, pps, nm;
    int jside_off;        // 0: columns use the row transform (symmetric units only)
    int ntri;             // units < ntri are symmetric / upper-triangular
    int ns2;              // leading dims whose second moments are needed (diag+grad path); D = all
    int colsplit;         // 64-row tiles run by 4 waves that split the columns (small batches)
};

// Arguments of the scalar-broadcast pair kernel (pair_kernel_sb.h): rollout hot path, variance units only.
struct PairSbArgs {
    const double* M;      // [ds][Np][Np] upper-triangular, weight 2 off the diagonal (pack.hip)
    const double* XT;     // [D][Np]
    const double* pp;     // [B][ds][pps]: cvec[D] = sc o u, sc[D]
    const double* G;      // [B][ds][Np][GW] column rows [h_j (D) | q_j | h_j^2 (NS2) | pad]
    double* part;         // [B][nwork][nm]
    const int* work;      // [nwork][4] = {unit, i0, j0, j1}
    int Np, B, ds, nwork, pps, nm;
    int rgroup;           // work items interleaved per trajectory in dispatch order (1 = item-major), see pair_kernel_sb.h
    int first_step;       // horizon step 1: derivatives w.r.t. the (constant) state inputs are not needed
    int colunroll;        // 4: four columns per loop iteration (the 256x64 tiling of mid-size batches, TB = 1); else one
    const int* ncol;      // device: columns that carry weight (N rounded up to 8, <= Np; gpmpc_pack::ncol_dev): the column loop ends there
};
static inline int gpmpc_sb_gw(int D, int ns2) { return (D + 1 + ns2 + 1) & ~1; }
int gpmpc_launch_pair_sb(int D, bool grad, int tb, int ns2, int waves, const PairSbArgs& a, hipStream_t s);
template <int D> int gpmpc_launch_pair_sb_D(bool grad, int tb, int ns2, int waves, const PairSbArgs& a, hipStream_t s);

// Arguments of the general scalar-broadcast pair kernel (pair_kernel_sbf.h): full S, variance + cross units.
struct PairSbfArgs {
    const double* M;      // [units][Np][Np]
    const double* XT;     // [D][Np]
    const double* pp;     // [B][nunits][pps]: row-side cvec[D] + T[D][D] (upper triangular)
    const double* G;      // [B][nunits][Np][GW] column rows [q_j (D) | |q_j|^2 | q_jk q_jl (k <= l < ns2) | pad]
    double* part;         // [B][nwork][nm]
    const int* work;      // [nwork][4] = {unit, i0, j0, j1}
    int Np, B, nunits, nwork, pps, nm, ntri;
    int cu;               // columns per loop iteration: 1 | 2 | 4 (pair_kernel_sbf.h)
    double* part0;        // [B][nwork]: the Z0 partial sums once more, contiguous (k_fc_head sums them for EVERY unit in every workgroup), or null
    int pstride;          // partial-sum slots per trajectory when the launch covers only the FIRST nwork of them (one lambda: the cross units' slots are
                          // written by pair_kernel_sbfx.h); 0 = nwork
};
// Arguments of the cross-unit pair kernel for packs with one lambda for all GPs (pair_kernel_sbfx.h)
struct PairSbfxArgs {
    const double* XT; const double* lam; const double* beta; const double* sf;
    const double* pp; const double* G;     // as PairSbfArgs: the records / column rows of unit `unit0` (any cross unit: they are all alike)
    const double* rows;                    // pack constant: [Np][RW] = [x_j (D) | C/4 sum_k x_jk^2 / lambda_k | beta_c,j (ds) | pad]
    double* part; double* part0;           // partial sums, slots base + pair * ntile + tile of a trajectory's pstride slots
    const int* pair_ab;
    int Np, N, B, nunits, unit0, pps, nm;
    int tj, ntile, jt;                     // 64-row blocks per side, tiles per trajectory (T (T + 1) / 2 * 64 / jt), columns per tile (64 | 16; 256: from `tiles`)
    const int* tiles;                      // jt = 256: [ntile][3] = {64-row block, first column, columns}; else null
    int base, pstride;
};
template <int D> int gpmpc_launch_pair_sbfx_D(bool grad, int ns2, const PairSbfxArgs& a, hipStream_t s);
int gpmpc_launch_pair_sbfx(int D, bool grad, int ns2, const PairSbfxArgs& a, hipStream_t s);
static inline int gpmpc_sbf_gw(int D, int ns2) { return (D + 1 + ns2 * (ns2 + 1) / 2 + 1) & ~1; }
int gpmpc_launch_pair_sbf(int D, bool grad, int ns2, int waves, const PairSbfArgs& a, hipStream_t s);
template <int D> int gpmpc_launch_pair_sbf_D(bool grad, int ns2, int waves, const PairSbfArgs& a, hipStream_t s);

// Arguments of the shared-lambda pair kernel (pair_kernel_sbs.h): all GPs of the bundle have the same length-scales.
struct PairSbsArgs {
    const double* M;      // [ds][Np][Np] as PairSbArgs
    const double* XT;     // [D][Np]
    const double* pp;     // [B][ds][pps]; the entry of GP 0 is read (the transform is the same for every GP)
    const double* G;      // [B][Np][GW] ONE set of column rows per trajectory
    double* part;         // [B][ds * tiles][nm]: tile q of GP a at item a * tiles + q
    const int* work;      // [nwork][4] = {GP group, i0, j0, tile index within a GP}
    int Np, B, ds, nwork, tiles, jt, pps, nm;
    int rgroup, first_step;
    const int* ncol;      // as PairSbArgs
};
// GPs per workgroup of the shared-lambda kernel: as many as keep the accumulators (NG x (1 + D + ds) doubles) within ~48,
// then balanced over the groups.
static inline int gpmpc_sbs_group(int ds, int D) {
    int cap = 48 / (1 + D + ds);
    if (cap > 4) cap = 4;
    if (cap < 2) cap = 2;
    const int groups = (ds + cap - 1) / cap;
    return (ds + groups - 1) / groups;
}
int gpmpc_launch_pair_sbs(int D, bool grad, int ng, int ns2, const PairSbsArgs& a, hipStream_t s);
template <int D> int gpmpc_launch_pair_sbs_D(bool grad, int ng, int ns2, const PairSbsArgs& a, hipStream_t s);

int gpmpc_launch_pair_lowprec(int D, int mode, const PairArgs& a, hipStream_t s);     // lowprec.hip (tolerance sweep)

// Arguments of the fused small-batch step kernel (step_fused.h; instantiated per D in fused_d*.o)
struct FusedArgs {
    // pack
    const double* XT; const double* beta; const double* lam; const double* sf; const double* M; const int* work;
    int Np, nwork;                         // (padded size only, see gpmpc_pack_resize) nwork = tile workgroups per trajectory (work items x column pieces per item)
    int tri64;                             // the work list is the plain 64x64 upper-triangular order: decode items arithmetically
    int ustart[GPMPC_MAX_DS + 1];          // workgroups of GP a are [ustart[a], ustart[a+1]) (64-row work lists are unit-contiguous)
    // problem
    const double* x0; const double* U; int B, H;
    // outputs / state
    double* means; double* vars; double* jac;
    double* sp; double* part; double* partz;
    int sps, nm;
    double* gscr;                          // mid-size form (q = 0 / 32 / 16): [B][ntile][columns][gw] column rows, one slot per tile workgroup
    int ntile;                             // tile workgroups per trajectory (= nwork, except with one lambda for all GPs: groups x tiles)
    int tiles;                             // one lambda for all GPs: tiles per GP (partial sums are laid out [GP][tile], nwork = ds * tiles)
    const int* ncol;                       // device: columns that carry weight (N rounded up to 8, <= Np; gpmpc_pack::ncol_dev): tile column loops end there
    int xcdmap;                            // 256-row forms, several trajectories: XCD-aware dispatch order (step_fused.h; set by the plan from B x ntile)
};
template <int D> int gpmpc_launch_step_fused_D(bool grad, int ns2, int q, int ng, const FusedArgs& a, int t, hipStream_t s);
// XCD-aware dispatch order of the one-launch form (step_fused.h): linear workgroup id L of a (gx, nB) grid whose first nt columns are tile
// workgroups -> (trajectory b, column bx).  Order: (8 tiles) x (all trajectories) x (tile within the 8), then the tiles beyond a multiple of 8 per
// trajectory, then the gx - nt role workgroups per trajectory: a bijection of [0, gx nB) onto [0, nB) x [0, gx) (tests/test_host_cpu.py).
__host__ __device__ inline void gpmpc_xcd_remap(int L, int gx, int nt, int nB, int* b, unsigned* bx) {
    const int nt8 = nt & ~7, full = nt8 * nB, allt = nt * nB;
    if (L < full) { const int grp = L / (8 * nB), r = L - grp * 8 * nB; *b = r >> 3; *bx = (unsigned)(grp * 8 + (r & 7)); }
    else if (L < allt) { const int rem = nt - nt8, L2 = L - full; *b = L2 / rem; *bx = (unsigned)(nt8 + L2 - *b * rem); }
    else { const int nr = gx - nt, L3 = L - allt; *b = L3 / nr; *bx = (unsigned)(nt + L3 - *b * nr); }
}

// Arguments of the trajectory-persistent whole-horizon kernel (traj_persist.h; instantiated per D in persist_d*.o)
struct PersistArgs {
    const double* XT; const double* beta; const double* lam; const double* sf; const double* M;
    int Np;
    const double* x0; const double* U; int B, H;
    double* means; double* vars; double* jac;      // [B][H+1][ds] x2, [B][H][2ds][2ds+da] (jac may be null: objective only)
    double* gscr;                                  // [B][ds][Np][GW] column rows, one slot per workgroup
    const int* ncol;                               // device: columns that carry weight (N rounded up to 8, <= Np); gpmpc_pack::ncol_dev
    int total;                                     // (host bookkeeping) columns of the flattened (GP, row block, column) space at ncol = Np
};
template <int D> int gpmpc_launch_persist_D(bool grad, int ns2, int waves, int ng, const PersistArgs& a, hipStream_t s);


// Implemented in pair_d*.hip (one translation unit per D so the build parallelises).
int gpmpc_launch_pair(int D, bool diag, bool grad, int tb, int waves, const PairArgs& a, hipStream_t s);
template <int D> int gpmpc_launch_pair_D(bool diag, bool grad, int tb, int waves, const PairArgs& a, hipStream_t s);

void gpmpc_set_error(const char* what, hipError_t e);
// Small host array (<= 512 bytes: hyper-parameters, index pairs) -> device memory, ordered on `s`, with the host bytes CONSUMED BEFORE
// THE CALL RETURNS: they travel as kernel arguments.  (hipMemcpyAsync from pageable memory may read the host buffer only when the
// stream gets there -- behind a wait on another stream that can be after the caller has freed it; seen as a wrong K matrix from a
// rebuild queued on a side stream.)  Returns a GPMPC_* code.
int gpmpc_upload_small(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s);
// GPMPC_OK if the calling thread's current device is the pack's, else GPMPC_E_DEVICE (kernels launched on another
// device would dereference this pack's memory without peer access: a GPU page fault, not an error code)
static inline int gpmpc_check_device(const gpmpc_pack* p) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return GPMPC_E_LAUNCH;
    return dev == p->device ? GPMPC_OK : GPMPC_E_DEVICE;
}
// pair-kernel timing classes (gpmpc_pair_kernel_time_class): the horizon-step-1 variant is cheaper than the full kernel
// (the fused small-batch step kernel -- mean sums, finish work and tiles of one horizon step in one launch -- has its own class)
enum { GPMPC_TIME_FULL = 0, GPMPC_TIME_FIRST = 1, GPMPC_TIME_FUSED = 2, GPMPC_TIME_CLASSES = 3 };
void gpmpc_tuned_free(void* table);
void gpmpc_tuned_clear(void* table);
void* gpmpc_lock_create();
void gpmpc_lock_destroy(void* lock);
void gpmpc_graph_cache_free(void* cache);
void gpmpc_cb_cache_free(void* cache);
void gpmpc_graph_cache_invalidate(void* cache);     // drop the captured graphs, keep streams / events / buffers
void gpmpc_cb_cache_invalidate(void* cache);
#define GPMPC_HIP(call)                                              \
    do {                                                             \
        hipError_t e_ = (call);                                      \
        if (e_ != hipSuccess) { gpmpc_set_error(#call, e_); return GPMPC_E_LAUNCH; } \
    } while (0)

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
// Sum over the 64 lanes of a wave, result in every lane.  DPP butterflies inside each row of 16 lanes (VALU moves: no
// LDS round trips as with __shfl_xor / ds_bpermute, ~8x lower latency), then the four row sums are read with
// v_readlane and added in a fixed order.
template <int CTRL>
__device__ __forceinline__ double gpmpc_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double gpmpc_readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += gpmpc_dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
    v += gpmpc_dpp_f64<0x4E>(v);        // quad_perm [2,3,0,1]
    v += gpmpc_dpp_f64<0x141>(v);       // row_half_mirror
    v += gpmpc_dpp_f64<0x140>(v);       // row_mirror: every lane of a row holds the row's sum
    return (gpmpc_readlane_f64(v, 0) + gpmpc_readlane_f64(v, 16)) + (gpmpc_readlane_f64(v, 32) + gpmpc_readlane_f64(v, 48));
}

// First half of wave_sum: every lane ends up with the sum of ITS ROW of 16 lanes (rows 0-15, 16-31, 32-47, 48-63).  The pair
// kernels store the four row sums and let the final cross-wave combine add 16 values instead of 4: that drops the 8
// v_readlane + 3 v_add_f64 of the second half (11 of 23 instructions per reduced value; the end-of-tile reduction of
// 22 values per wave was 2.4 % of a C3 launch).
__device__ __forceinline__ double wave_row_sum(double v) {
    v += gpmpc_dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
    v += gpmpc_dpp_f64<0x4E>(v);        // quad_perm [2,3,0,1]
    v += gpmpc_dpp_f64<0x141>(v);       // row_half_mirror
    v += gpmpc_dpp_f64<0x140>(v);       // row_mirror
    return v;
}

// The same for workgroups of exactly 4 waves (256 threads), with the cheaper row sums: scratch >= 16*NV doubles
// ([wave][row of 16 lanes][value]); fixed summation order.
template <int NV>
__device__ __forceinline__ void block_sum4_rows(const double (&v)[NV], double* scratch, double* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_row_sum(v[k]);
        if ((lane & 15) == 0) scratch[(w * 4 + (lane >> 4)) * NV + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int ww = 0; ww < 4; ++ww) {
            const double* r4 = &scratch[ww * 4 * NV + threadIdx.x];
            s += (r4[0] + r4[NV]) + (r4[2 * NV] + r4[3 * NV]);
        }
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// Sum NV per-thread values over the workgroup (<= 16 waves).  Result in out[0..NV) (LDS), visible to all
// threads after return.  scratch: LDS, >= 16*NV doubles.  Deterministic summation order.
template <int NV>
__device__ __forceinline__ void block_sum(const double (&v)[NV], double* scratch, double* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = wave_sum(v[k]);
        if (lane == 0) scratch[w * NV + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int ww = 0; ww < nw; ++ww) s += scratch[ww * NV + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}
