// One launch per horizon step for SMALL batches (the B = 1 callbacks of a solver loop, src/mpc.py:202-255, which Ipopt
// issues strictly one after the other): the per-step "head" kernel (finish step t-1, prepare step t) and the staged pair
// kernel (pair_kernel.h) fused into a single grid, so that a rollout is H + 1 launches instead of 2H + 1.
//
// Why fusion by REDUNDANCY and not by a last-arriver hand-off: at these sizes every kernel is a chain of dependent
// latencies (measured on MI355X at N = 512, B = 1: head 7.2 us, pair 6.8 us, boundary 1.9 us, all of it waiting), and an
// in-kernel hand-off (release fence + ticket + acquire fence) costs what a kernel boundary costs (MI355X: ~1.7 us per
// fence).  Instead every workgroup of launch t recomputes the cheap part of the old head itself -- the Z0 sums of step
// t-1, hence the input variances of step t -- from a compact array the previous launch wrote, and the rest of the head
// runs in 2 ds extra workgroups of the SAME launch, beside the tiles instead of before them.  Nothing inside a launch
// depends on another workgroup of that launch.
//
//   grid = (nwg + 2 ds, B) x 256 threads
//   blockIdx.x <  nwg          : piece of a tile (GP a, 64 rows x 64|128 columns) of the N^2 sum of step t (pair_kernel.h, staged
//                                form).  Q = 4 while whole tiles would leave most SIMDs without a wave: a workgroup then takes
//                                16 of a tile's 64 columns, 4 per wave (a workgroup is confined to one CU: more waves per
//                                workgroup do not spread the column loop, more workgroups do); else Q = 1, 16 columns per wave
//   nwg      <= .. < nwg + ds  : GP a: mean sums of step t over the N points                     (step.hip::prep_step)
//   nwg + ds <= ..             : GP a: outputs and Jacobian rows of step t-1                     (step.hip::finish_step);
//                                nothing in the next launch waits for these, only the tail kernel does
//
// Inside a workgroup every global load that depends on nothing computed in this launch is issued FIRST, into registers
// (tile rows and columns, weights, the previous step's Z0 partials and scalars, the exp table, whose LDS write is
// deferred), so that the chain is one memory round trip, then arithmetic.  In-kernel timeline of an earlier form
// (s_memtime, N = 512, 16-wave workgroups): 16.8 k cycles per launch, of which 3.4 k waited on the table copy, 2.6 k on
// the Z0 partials and 5.5 k in the reduction of the 16 waves of a workgroup (tools/fused_stamps.py, profiles/r02/README.md).
//
// Reference restated: Dynamics.forward_propagate_torch (src/dynamics.py:145-189), mean_prop_torch / variance_prop_torch
// (src/tools/uncertainty_prop.py:296-399), exactly as step.hip / pair_kernel.h do; the closed forms are in step.hip.
// State between launches (double-buffered by step parity): sp [2][B][ds][sps], part [2][B][nwg][nm], partz [2][B][nwg].
#pragma once
#include "gpmpc_internal.h"
#include "fast_exp.h"

// Item q of the 64x64 upper-triangular list of one unit (pack.hip::build_worklist: row tile r, then column tile c >= r;
// T tiles per side): r is the largest row with start(r) = r T - r (r - 1) / 2 <= q.  Saves the dependent load of the item.
__device__ __forceinline__ void gpmpc_tri_decode(int q, int T, int* r_out, int* c_out) {
    const float tt = 2.0f * T + 1.0f;
    int r = (int)((tt - sqrtf(tt * tt - 8.0f * (float)q)) * 0.5f);
    r = r < 0 ? 0 : (r > T - 1 ? T - 1 : r);
    while (r > 0 && r * T - r * (r - 1) / 2 > q) --r;
    while (r + 1 < T && (r + 1) * T - (r + 1) * r / 2 <= q) ++r;
    *r_out = r;
    *c_out = r + (q - (r * T - r * (r - 1) / 2));
}

// Diagnostic build only (-DGPMPC_FUSED_STAMPS): s_memtime stamps of the phases of one tile workgroup and of the mean-sum
// workgroup of GP 0 at horizon step 5, read back with gpmpc_debug_stamps (tools/fused_stamps.py).  The product build
// executes no stamp.
#ifndef GPMPC_STAMP_D
#define GPMPC_STAMP_D 4
#endif
#ifdef GPMPC_FUSED_STAMPS
static __device__ unsigned long long g_fused_stamps[64];
static __device__ unsigned long long g_fused_wg[2 * 8192];         // [start | end] of every workgroup of trajectory 0 at step 5 (100 MHz counter)
#define GPMPC_STAMP_WG(which) do { if (t == 5 && blockIdx.y == 0 && tid == 0 && blockIdx.x < 8192) g_fused_wg[(which) * 8192 + blockIdx.x] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GPMPC_STAMP(slot) do { if (t == 5 && blockIdx.y == 0 && tid == 0 && (blockIdx.x == 0 || (int)blockIdx.x == A.nwork || (int)blockIdx.x == A.nwork + NS2)) \
    g_fused_stamps[(blockIdx.x == 0 ? 0 : ((int)blockIdx.x == A.nwork ? 16 : 32)) + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define GPMPC_STAMP_REAL(slot) do { if (t == 5 && blockIdx.y == 0 && tid == 0 && blockIdx.x == 0) g_fused_stamps[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GPMPC_STAMP(slot) do { } while (0)
#define GPMPC_STAMP_REAL(slot) do { } while (0)
#define GPMPC_STAMP_WG(which) do { } while (0)
#endif

// The tile pieces of this kernel evaluate 4..16 columns per wave: copying the 16 KB exp table into LDS per workgroup (7 MB
// per launch at N = 512, most of the burst every workgroup opens with) costs more than the ~12 extra fp64 instructions per
// pair of the table-free exp.  GPMPC_FUSED_TABLE=1 restores the table (A/B).
#ifndef GPMPC_FUSED_TABLE
#define GPMPC_FUSED_TABLE 0
#endif
#ifndef GPMPC_FUSED_UNROLL
// Columns of the evaluation loop in flight per wave.  Fully unrolled (16 for a whole tile's share) three instantiations were
// compiled to 256 VGPRs; 2 / 4 / 8 avoid that and differ little, 2 gives the pendulum shape <3, 2> a fourth wave per SIMD
// (N = 300, B = 64: 0.231 ms per batch against 0.256 / 0.252; `profiles/r02/batch_size_map.txt`).
#define GPMPC_FUSED_UNROLL 2
#endif
#define GPMPC_FUSED_PZ 4        // Z0 partials of one GP prefetched per thread: covers 256 * 4 workgroups per GP
#ifndef GPMPC_FUSED_PZ_SB
#define GPMPC_FUSED_PZ_SB 6     // ... of the 256-row forms: 384 per GP in the first round trip (N = 2048 on 256x32 tiles has 288), the rest 4 at a time
#endif
#ifndef GPMPC_FUSED_XCDMAP
#define GPMPC_FUSED_XCDMAP 1      // XCD-aware dispatch order of the 256-row forms for several trajectories (A/B: profiles/r05/ab19_xcdmap.txt)
#endif
#ifndef GPMPC_FUSED_PIPE_EARLY
#define GPMPC_FUSED_PIPE_EARLY 1    // first group requested in phase 0 (1) or when the column loop starts (0)
#endif
#ifndef GPMPC_FUSED_PIPE_ILP
#define GPMPC_FUSED_PIPE_ILP 2      // columns whose dependency chains the scheduler may interleave (4 held ~20 more registers: spills at the 96 of 5 waves)
#endif
#ifndef GPMPC_FUSED_MEAN_UNROLL
#define GPMPC_FUSED_MEAN_UNROLL 2   // points per round trip of the mean-sum workgroups of the 256-row forms
#endif
#ifndef GPMPC_FUSED_STAGED
#define GPMPC_FUSED_STAGED 1      // the 256-row forms' column loop in staged batches with the next weight group in flight (round 5); 0: the four-column body
#endif
#ifndef GPMPC_FUSED_MGS
#define GPMPC_FUSED_MGS 2         // columns per group of weight loads of the staged loop (two groups in flight)
#endif
#ifndef GPMPC_FUSED_PIPE
// 256-row forms (Q = 0 / 32 / 16, one GP per tile workgroup): the weight stream M_ij is software-pipelined -- the first group of
// columns is requested in phase 0, behind every load the prologue waits for (vector-memory results return in order, so the prologue's
// counted waits do not include it), and group g + 1 is requested before group g is evaluated.  One or a few trajectories of a large
// training set stream M from HBM / Infinity Cache with a handful of waves per SIMD: a wave that issues its loads, waits, and only then
// evaluates has nothing in flight half of the time (N = 2048, B = 1: 3.2 TB/s of the 67 MB per step; N = 4096, ds = 6: 3.3 TB/s).
// Same sums in the same order: results are bit-identical to the unpipelined loop.
// MEASURED (round 4, profiles/r04/ab_fused_pipeline.txt): it LOSES 4-20 % everywhere it applies, with the first group requested in
// phase 0 or at the loop, two or four columns interleaved -- N = 2048, B = 1: 0.485 -> 0.509 ms per rollout; N = 4096, ds = 6, B = 1:
// 3.59 -> 3.90 ms; N = 1024, B = 16: 0.98 -> 1.07 ms.  The in-kernel timeline (tools/fused_stamps.py, profiles/r04/fused_stamps_*.txt)
// says why: at N = 2048, B = 1 the 4.5 waves per SIMD of the single workgroup generation already saturate VALU issue in the column loop
// (2170 cycles per four columns = 4.5 waves x 4 x 27 instructions x 4.4 cycles), the launch is prologue + loop + reduction in lock-step;
// at N = 4096, ds = 6 a column costs a wave 834 cycles against 145 of issue because its G row (28 SGPRs) is fetched by scalar loads one
// column at a time -- the SGPR file holds two of them -- and neither is a matter of weight loads in flight.  Kept as an A/B build (1).
#define GPMPC_FUSED_PIPE 0
#endif

// Q = 0: the MID-SIZE form (round 3).  The tile workgroups take 256 x 64 tiles (work list 2) and run the SCALAR-BROADCAST column
// loop of pair_kernel_sb.h (four columns in flight, table exp) instead of the staged one: a batch of B = 4...32 trajectories of
// a large N then needs ONE launch per horizon step where the two-kernel form needs head + pair kernel, and the O(N) mean sums
// and the finish work run beside the tiles instead of in front of them.  The column rows G[j] = [h_j | q_j N/ln2 | h_jk^2] the
// loop reads through scalar loads are computed by the workgroup itself for ITS 64 columns (wave 0, one column per lane),
// stored to a scratch slot of its own, and read back after s_waitcnt vmcnt(0) (the stores have reached L2), an s_dcache_inv
// (the scalar cache may hold the slot's previous contents) and the workgroup barrier.
// layout of sp (doubles), as step.hip: 0 c | 1 mu | 2 sf2 | 3 A[D] | 3+D scale[D] | 3+2D dmu_du[D] | 3+3D dmu_ds[D]
#ifndef GPMPC_FUSED_SB_W4_FROM
#define GPMPC_FUSED_SB_W4_FROM 6    /* input dimension from which the 256-row forms are compiled for 4 waves per SIMD (A/B knob) */
#endif
#ifndef GPMPC_FUSED_SB_WAVES
#define GPMPC_FUSED_SB_WAVES 5      /* waves per SIMD the mid-size form is compiled for up to D = 5 (A/B knob; 4 from D = 6: 21 spilled registers at 96) */
#endif
// NG > 1 (Q = 0 only): every GP has the SAME length-scales (pair_kernel_sbs.h): a tile workgroup takes a group of NG GPs, evaluates the
// exponent and the exp ONCE per pair and applies them to NG weight loads; work list wl_sh[1] (items {group, i0, j0, tile}),
// partial sums laid out [GP][tile], one row of column data per trajectory instead of one per GP.
template <int D, int NS2, bool GRAD, int Q, int NG = 1>
__global__ __launch_bounds__(256, (Q == 0 || Q == 32 || Q == 16 || Q == 256) ? (NG >= 3 ? 3 : ((NG == 2 || D >= GPMPC_FUSED_SB_W4_FROM) ? 4 : GPMPC_FUSED_SB_WAVES)) : 1)
void k_step_fused(FusedArgs A, int t) {
    constexpr bool SB = Q == 0 || Q == 32 || Q == 16 || Q == 256;      // (256: runs of up to 256 columns, work list 7 -- one generation of workgroups per trajectory)
    constexpr bool SH = NG > 1;
    static_assert(!SH || Q == 0, "the shared-lambda tile role runs on 256x64 tiles");
    constexpr int NC = Q == 0 ? 64 : Q;                                            // columns of a tile of the mid-size form
    constexpr int QQ = SB ? 1 : Q;
    constexpr int DS = NS2, DA = D - NS2, NM = GRAD ? 1 + 2 * D : 1, DP = (D + 1) & ~1;
    constexpr int NV = 1 + 2 * D, NT = 256, CW = 16 / QQ, NCOL = 64 / QQ;          // NCOL columns of a chunk per workgroup
    constexpr int TABN = GPMPC_EXP_N / NT; (void)TABN;
    constexpr bool TABLE = SB || GPMPC_FUSED_TABLE;
    constexpr int GW = (D + 1 + NS2 + 1) & ~1;                                     // doubles per G row, as PairSbTraits
    __shared__ double s_tab[TABLE ? GPMPC_EXP_N : 1];
    __shared__ __attribute__((aligned(16))) double s_hj[SB ? 2 : NCOL * DP];
    __shared__ double s_red[16 * NV * NG];
    __shared__ double s_out[NV];
    __shared__ double s_zw[GPMPC_MAX_DS];
    __shared__ double s_mu[GPMPC_MAX_DS], s_var[GPMPC_MAX_DS], s_z0[GPMPC_MAX_DS], s_c[GPMPC_MAX_DS];
    __shared__ double s_spp[4 * D];
    __shared__ double s_uin[D], s_sin[D], s_sck[D], s_cv[D];     // input moments of step t and the pair transform h = sc (u - x)
    const int tid = threadIdx.x, lane = tid & 63;
#if !GPMPC_FUSED_XCDMAP
    const int b = blockIdx.y; unsigned bx = blockIdx.x;
#else
    int b = blockIdx.y; unsigned bx = blockIdx.x;
    // XCD-aware order for several trajectories (256-row forms): workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), each
    // XCD with its own L2.  In the natural order trajectory b + 1 visits a tile ntile + 2 ds workgroups after trajectory b -- another XCD as
    // often as not, and long after the tile has left that L2: every trajectory streams every weight tile from the Infinity Cache again.  Here
    // the dispatch order is (8 tiles) x (all trajectories) x (tile within the 8): tile k of EVERY trajectory runs on XCD k mod 8, the B visits
    // next to each other in dispatch order; the role workgroups of all trajectories come last.
    if (SB && Q != 256 && A.xcdmap) {                    // (not on runs, work list 7: a list for ONE trajectory)
        const int gx = gridDim.x;
        gpmpc_xcd_remap((int)blockIdx.y * gx + (int)blockIdx.x, gx, A.ntile, A.B, &b, &bx);
    }
#endif
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = ((int)bx >= A.ntile) + ((int)bx >= A.ntile + DS);     // 0 tile | 1 mean sums | 2 finish  (as a sum of comparisons: the select form of the same value costs the D = 5 instances 13 VGPR spills -- block placement)
    const int Np = A.Np, pprev = (t - 1) & 1, pcur = t & 1;
    const int am = role == 1 ? (int)bx - A.ntile : (int)bx - A.ntile - DS;         // GP of a non-tile workgroup

    // All kernel arguments used below are read HERE, unconditionally (the empty asm is an unconditional use): left to
    // itself the compiler loads each field inside the branch that first needs it, one s_load + s_waitcnt after the other
    // (ten dependent scalar round trips, ~3 k cycles of the 4.8 k this phase took before).
    asm volatile("" ::"s"(A.XT), "s"(A.beta), "s"(A.lam), "s"(A.sf), "s"(A.M), "s"(A.work), "s"(A.x0), "s"(A.U));
    asm volatile("" ::"s"(A.means), "s"(A.vars), "s"(A.jac), "s"(A.sp), "s"(A.part), "s"(A.partz), "s"(A.Np), "s"(A.nwork),
                 "s"(A.tri64), "s"(A.B), "s"(A.H), "s"(A.sps), "s"(A.nm));
    // columns >= N carry zero weight: tile column loops end at N rounded up to 8 (round 5, as traj_persist.h; the count lives in the pack's
    // device memory: a captured launch stays valid while the training set grows within its padded size)
    // The count is fetched with the phase-0 loads below (a VECTOR load through an opaque zero offset: issued there, waited for at the
    // column loops).  As a scalar load the compiler either waits for it at once -- its round trip, ~2.5 k cycles after a bandwidth-bound
    // launch has swept the L2, in front of every phase-0 load: N = 2048, B = 1 0.48 -> 0.54 ms per rollout -- or sinks it to its use.
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const int ncw_v = A.ncol[zoff];
    int ust[DS + 1];
#pragma unroll
    for (int a = 0; a <= DS; ++a) { ust[a] = A.ustart[a]; asm volatile("" ::"s"(ust[a])); }
    GPMPC_STAMP(0);
    GPMPC_STAMP_WG(0);
    GPMPC_STAMP_REAL(12);       // constant 100 MHz counter beside the shader-clock one: the clock the launch ran at
#ifdef GPMPC_FUSED_STAMPS
    {   // diagnostic: latency of ONE read-only load (length-scales), ONE load of data the previous launch wrote (partz), alone
        const double probe1 = __builtin_nontemporal_load(A.lam);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(probe1));
        GPMPC_STAMP(8);
        const double probe2 = __builtin_nontemporal_load(A.partz + ((size_t)((t - 1) & 1) * A.B + blockIdx.y) * A.nwork);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(probe2));
        GPMPC_STAMP(9);
        const double probe3 = __builtin_nontemporal_load(A.M + 64 * (size_t)A.Np + tid);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(probe3));
        GPMPC_STAMP(10);
    }
#endif
    // ---- phase 0: every load that depends on nothing computed in this launch, issued back to back into registers.
    //      Per-thread conditions are turned into clamped addresses + selects: a load under a divergent branch is
    //      followed by its own s_waitcnt, which serialises the round trips. ---------------------------------------------------
    int unit = 0, i0 = 0, j0 = 0, j1 = 0, jq = 0;
    double xrow[D], xcol[D], mpre[CW];
    double tabreg[(TABLE && !SB) ? TABN : 1];
    if (role == 0) {
        const int item = (int)bx / QQ;
        jq = ((int)bx - item * QQ) * NCOL;                                    // this workgroup's columns of every chunk
        if (A.tri64) {
            const int T = Np >> 6, per = T * (T + 1) / 2;
            unit = item / per;
            int r, c;
            gpmpc_tri_decode(item - unit * per, T, &r, &c);
            i0 = r << 6; j0 = c << 6; j1 = j0 + 64;
        } else {
            const int* wk = A.work + 4 * item;                                        // uniform address: scalar loads
            unit = wk[0]; i0 = wk[1]; j0 = wk[2]; j1 = wk[3];
        }
    }
    const int a_own = role == 0 ? (SH ? (unit * NG < DS ? unit * NG : DS - 1) : unit) : am;      // the GP whose length-scales this workgroup needs
    // Load-instruction diet: a CU's vector memory path moves 64 B per clock, so every wave-wide load costs ~8 cycles of it
    // whatever it fetches; with 120 of them per workgroup (4 waves x 30) the opening burst took 3.7 k cycles (in-kernel
    // stamps), against ~500 for one load alone.  Per-dimension scalars are therefore loaded by wave 0 only, and the Z0
    // partials of GP a by wave a % 4 only (which then reduces them without a cross-wave combine).
    const int kd = lane < D ? lane : D - 1, kg = lane < DS ? lane : DS - 1;
    double lam_k = 0.0, sp_c = 0.0, sp_mu = 0.0, sp_sf2 = 0.0, u_act = 0.0, spp_v = 0.0;
    if (w == 0) {                                                   // wave-uniform branch: no exec masking, no wait
        lam_k = A.lam[a_own * D + kd];
        if (t > 1) {
            const double* sp = A.sp + (((size_t)pprev * A.B + b) * DS + kg) * A.sps;
            sp_c = sp[0]; sp_mu = sp[1]; sp_sf2 = sp[2];
        } else {
            sp_mu = A.x0[(size_t)b * DS + kg];
        }
        if (DA > 0) u_act = A.U[((size_t)b * A.H + (t - 1)) * DA + (lane >= DS && lane < D ? lane - DS : 0)];
    }
    if (role == 2 && t > 1 && GRAD && w == 1)                       // A, scale, dmu_du, dmu_ds of step t-1 (4 D <= 32 values)
        spp_v = A.sp[(((size_t)pprev * A.B + b) * DS + am) * A.sps + 3 + (lane < 4 * D ? lane : 4 * D - 1)];
    // Z0 partials of step t-1: wave a % 4 fetches those of GP a, GPMPC_FUSED_PZ per lane up front (covers 256 workgroups
    // per GP, i.e. every configuration with column pieces; more are looped over in phase 1)
    constexpr int PZ = SB ? GPMPC_FUSED_PZ_SB : GPMPC_FUSED_PZ;
    double pzr[(DS + 3) / 4][PZ];
    const double* pz = A.partz + ((size_t)pprev * A.B + b) * A.nwork;
    if (t > 1) {
#pragma unroll
        for (int g = 0; g < (DS + 3) / 4; ++g) {
            const int a = w + 4 * g;                                // wave-uniform
            if (a < DS) {
#pragma unroll
                for (int r = 0; r < PZ; ++r) {
                    const int wi = ust[a < DS ? a : 0] + lane + r * 64;
                    const int we = ust[a < DS ? a + 1 : 1];
                    const double v = pz[wi < we ? wi : ust[a < DS ? a : 0]];           // clamped: always a valid address
                    pzr[g][r] = wi < we ? v : 0.0;
                }
            }
        }
    }
    if (role == 0 && SB) {
        // 256 x 64 tile: wave w owns rows i0 + 64 w ... (clamped: the last row tile of a padded size that is no multiple of 256
        // has waves past the end, which skip the column loop), wave 0 also fetches the tile's 64 columns, one per lane
        const int ir = i0 + 64 * w + lane;
#pragma unroll
        for (int k = 0; k < D; ++k) xrow[k] = A.XT[(size_t)k * Np + (ir < Np ? ir : Np - 1)];
        if (w == 0 || NC > 64) {                                          // (runs of up to 256 columns: every wave fetches 64 of them)
            const int cidx = NC > 64 ? tid : (lane & (NC - 1));
#pragma unroll
            for (int k = 0; k < D; ++k) xcol[k] = A.XT[(size_t)k * Np + (j0 + cidx < Np ? j0 + cidx : Np - 1)];      // (clamped: a run may end within NC columns of the end)
        }
    }
    if (role == 0 && !SB) {
#pragma unroll
        for (int k = 0; k < D; ++k) xrow[k] = A.XT[(size_t)k * Np + i0 + lane];      // the 4 waves share the tile's 64 rows
        const int jfirst = (j0 + 63 < i0) ? j0 + 64 : j0;                             // first column chunk that carries weight
        if (w == 0) {                                                                 // the columns are staged by wave 0
            const int jcol = jfirst + jq + (lane < NCOL ? lane : 0);
#pragma unroll
            for (int k = 0; k < D; ++k) xcol[k] = A.XT[(size_t)k * Np + jcol];
        }
        const double* Mc = A.M + (size_t)unit * Np * Np + (size_t)(jfirst + jq) * Np + i0 + lane;
#pragma unroll
        for (int q = 0; q < CW; ++q) mpre[q] = Mc[(size_t)(w * CW + q) * Np];
    }
    if (SB && role == 0) {
        // exp table: global -> LDS directly (16 x 1 KiB wave-instructions per workgroup, no registers held while they fly);
        // the compiler drains them (vmcnt) in front of the barrier that ends phase 1
        typedef const void __attribute__((address_space(1))) gvoid;
        typedef void __attribute__((address_space(3))) lvoid;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = ((w * 4 + r) * 64 + lane) * 2;
            __builtin_amdgcn_global_load_lds((gvoid*)(gpmpc_exp2_table + e), (lvoid*)(s_tab + e), 16, 0, 0);
        }
    } else if (TABLE && role == 0) {
#pragma unroll
        for (int r = 0; r < TABN; ++r) tabreg[r] = gpmpc_exp2_table[tid + r * NT];   // LDS write deferred: see below
    }
    // 256-row forms: the first group of weight columns of this wave, requested LAST in phase 0 (see GPMPC_FUSED_PIPE above)
    constexpr bool PIPE = SB && !SH && GPMPC_FUSED_PIPE;
    constexpr bool EARLY = PIPE && GPMPC_FUSED_PIPE_EARLY;
    constexpr int MG = 4;                                           // columns per group (two groups in flight)
    double mga[PIPE ? MG : 1];
    const int iw0_t = i0 + 64 * w;
    const bool wave_on = role == 0 && iw0_t < Np && (NC > 64 ? j1 > iw0_t : j0 >= iw0_t);    // tiles left of the wave's diagonal block carry no weight (wave-uniform; runs, NC = 256: a run that ENDS left of it)
    __amdgpu_buffer_rsrc_t Mrs_t = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double*>(A.M + (size_t)unit * Np * Np + (size_t)j0 * Np + (wave_on ? iw0_t : 0)), 0, 0x7fffffff, 0x00020000);
    if constexpr (EARLY) {
        if (wave_on) {
#pragma unroll
            for (int q = 0; q < MG; ++q)
                mga[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs_t, lane * 8, q * Np * 8, 0));
        }
        __builtin_amdgcn_sched_barrier(0);                          // ... and not moved in front of the loads above
    }
    // mean sums: this thread's first points (the loop below loads the ones beyond)
    constexpr int PF = SB ? 1 : 2;                                  // (mid-size form: registers; N >= 1024 loops anyway)
    double xpt[PF][D], bpt[PF], sf_a = 0.0;
    if (role == 1) {
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            const int i = tid + r * NT, ic = i < Np ? i : 0;
#pragma unroll
            for (int k = 0; k < D; ++k) xpt[r][k] = A.XT[(size_t)k * Np + ic];
            const double bv = A.beta[(size_t)am * Np + ic];
            bpt[r] = i < Np ? bv : 0.0;
        }
        sf_a = A.sf[am];
    }
    GPMPC_STAMP(1);

    // ---- phase 1: mean and variance of step t-1 for ALL GPs, identically in every workgroup; thread k < D goes straight on
    //      to the scalars of ITS input dimension, so that one barrier publishes everything ----------------------------------
    if (role == 2 && t > 1 && GRAD && w == 1 && lane < 4 * D) s_spp[lane] = spp_v;
    if (t > 1) {
#pragma unroll
        for (int g = 0; g < (DS + 3) / 4; ++g) {
            const int a = w + 4 * g;
            if (a < DS) {                                           // wave-uniform: wave a % 4 owns GP a's Z0 sum, fixed order
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < PZ; ++r) s += pzr[g][r];
                // beyond the prefetched ones (N = 4096 on 256x64 tiles: 544 per GP): four loads per round trip, added in index order
                const int u0 = ust[a < DS ? a : 0], u1 = ust[a < DS ? a + 1 : 1];
                for (int wb = u0 + PZ * 64; wb < u1; wb += 4 * 64) {
                    double e[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int wi = wb + lane + r * 64; const double v = pz[wi < u1 ? wi : u0]; e[r] = wi < u1 ? v : 0.0; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) s += e[r];
                }
                s = wave_sum(s);
                if (lane == 0) s_zw[a] = s;
            }
        }
        __syncthreads();
    }
    GPMPC_STAMP(2);
    if (tid < D) {
        const int k = tid;
        double uk, sk;
        if (k < DS) {
            if (t == 1) { uk = sp_mu; sk = GPMPC_INIT_VAR; }
            else {
                const int kk = k < DS ? k : 0;
                const double z0 = s_zw[kk];
                uk = sp_mu;
                sk = sp_sf2 - sp_c * z0 - sp_mu * sp_mu;          // no clamp (src/tools/uncertainty_prop.py:399)
                s_z0[kk] = z0; s_c[kk] = sp_c;
            }
            s_mu[k < DS ? k : 0] = uk; s_var[k < DS ? k : 0] = sk;
        } else {
            uk = u_act; sk = GPMPC_ACTION_VAR;                    // src/dynamics.py:162
        }
        s_uin[k] = uk; s_sin[k] = sk;
        // sqrt(0.125 / x) as rsqrt(8 x): v_rsq_f64 + refinement instead of a division and a square root, which were ~900
        // cycles of dependent latency on the critical path of every launch (thread k < D sits in wave 0, which loaded
        // lam_k, sp_*, u_act); tile and mean-sum workgroups run this same expression, the finish workgroups read it from sp
        const double sc = rsqrt(8.0 * (0.5 * lam_k + sk));
        s_sck[k] = sc;
        s_cv[k] = sc * uk;
    }
    if (TABLE && !SB && role == 0) {
#pragma unroll
        for (int r = 0; r < TABN; ++r) s_tab[tid + r * NT] = tabreg[r];
    }
    __syncthreads();
    GPMPC_STAMP(3);

    if constexpr (SB) {
    if (role == 0) {
        // ---- 256 x 64 tile of the N^2 sum of step t, scalar-broadcast form (pair_kernel_sb.h: same expressions, same order) ----
        double sck[D], cv[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { sck[k] = s_sck[k]; cv[k] = s_cv[k]; }
        double* __restrict__ Gs = A.gscr + ((size_t)b * A.ntile + bx) * (size_t)(NC * GW);
        if (w == 0 || NC > 64) {                                         // column rows of this tile, one column per lane (< NC); runs: per thread
            double g[GW], qh = 0.0;
#pragma unroll
            for (int k = 0; k < GW; ++k) g[k] = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double h = fma(-sck[k], xcol[k], cv[k]);
                g[k] = h;
                qh = fma(h, h, qh);
                if (k < NS2) g[D + 1 + k] = h * h;
            }
            g[D] = GPMPC_EXP_NEG_INV_C * qh;
            double2* dst = reinterpret_cast<double2*>(Gs + (size_t)(NC > 64 ? tid : (lane & (NC - 1))) * GW);
            if (NC >= 64 || lane < NC) {
#pragma unroll
                for (int k = 0; k < GW / 2; ++k) dst[k] = make_double2(g[2 * k], g[2 * k + 1]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the rows are in L2 (this CU's vector L1 writes through)
            __builtin_amdgcn_s_dcache_inv();                             // drop what the scalar cache may hold of this slot
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const int iw0 = i0 + w * 64;
        double hi2[D], qi;
        {
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double h = fma(-sck[k], xrow[k], cv[k]);
                hi2[k] = (2.0 * GPMPC_EXP_NEG_INV_C) * h;
                q = fma(h, h, q);
            }
            qi = GPMPC_EXP_NEG_INV_C * q;
        }
        constexpr int NA = GRAD ? 1 + D + NS2 : 1;
        if constexpr (SH) {
            // ---- group of NG GPs with one lambda: exponent and exp once per pair (pair_kernel_sbs.h: same expressions, same order) ----
            const int tile = j1;                                         // 4th entry of a shared-list item: the tile's index within its GP
            double accs[NG][NA];
#pragma unroll
            for (int q = 0; q < NG; ++q)
#pragma unroll
                for (int m = 0; m < NA; ++m) accs[q][m] = 0.0;
            __syncthreads();                                             // G rows and exp table ready
            if (iw0 < Np && j0 >= iw0) {
                const double* Gl = Gs;
                asm volatile("" : "+s"(Gl) :: "memory");
                __amdgpu_buffer_rsrc_t Mrs[NG];
#pragma unroll
                for (int q = 0; q < NG; ++q) {
                    const int a = unit * NG + q < DS ? unit * NG + q : DS - 1;       // a partial last group re-reads the last GP
                    Mrs[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A.M + ((size_t)a * Np + j0) * Np + iw0), 0, 0x7fffffff, 0x00020000);
                }
                const int lane8 = lane * 8;
                constexpr int CU = NG >= 3 ? 1 : 2;
                const int ncw = __builtin_amdgcn_readfirstlane(ncw_v);
                const int ncl = ncw - j0 < NC ? ncw - j0 : NC;               // columns of this tile that carry weight (a multiple of 8, or <= 0)
                for (int jc = 0; jc < ncl; jc += CU) {
                    double mij[CU][NG];
#pragma unroll
                    for (int c = 0; c < CU; ++c)
#pragma unroll
                        for (int q = 0; q < NG; ++q)
                            mij[c][q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[q], lane8, (jc + c) * Np * 8, 0));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < CU; ++c) {
                        typedef const double __attribute__((address_space(4))) gpmpc_cdouble;
                        const gpmpc_cdouble* __restrict__ g = (const gpmpc_cdouble*)(Gl + (size_t)(jc + c) * GW);
                        double sx = qi + g[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) sx = fma(hi2[k], g[k], sx);
                        const double e = gpmpc_exp_neg_scaled(sx, s_tab);
#pragma unroll
                        for (int q = 0; q < NG; ++q) {
                            const double P = mij[c][q] * e;
                            accs[q][0] += P;
                            if (GRAD) {
#pragma unroll
                                for (int k = 0; k < D; ++k) accs[q][GRAD ? 1 + k : 0] = fma(P, g[k], accs[q][GRAD ? 1 + k : 0]);
#pragma unroll
                                for (int k = 0; k < NS2; ++k) accs[q][GRAD ? 1 + D + k : 0] = fma(P, g[D + 1 + k], accs[q][GRAD ? 1 + D + k : 0]);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                double z[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) z[m] = 0.0;
                const double rs = accs[q][0];
                z[0] = rs;
                if (GRAD) {
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double h = (0.5 / GPMPC_EXP_NEG_INV_C) * hi2[k], v = accs[q][GRAD ? 1 + k : 0];
                        z[GRAD ? 1 + k : 0] = fma(h, rs, v);
                        if (k < NS2) z[GRAD ? 1 + D + k : 0] = fma(h * h, rs, fma(2.0 * h, v, accs[q][GRAD ? 1 + D + (k < NS2 ? k : 0) : 0]));
                    }
                }
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const double sr = wave_row_sum(z[m]);
                    if ((lane & 15) == 0) s_red[((w * 4 + (lane >> 4)) * NG + q) * NM + m] = sr;
                }
            }
            __syncthreads();
            for (int idx = tid; idx < NG * NM; idx += NT) {
                const int q = idx / NM, m = idx - q * NM;
                const int a = unit * NG + q;
                if (a < DS) {
                    double sum = 0.0;                                    // fixed order: waves, each as (row 0 + row 1) + (row 2 + row 3)
                    for (int ww = 0; ww < 4; ++ww) {
                        const double* r4 = &s_red[(ww * 4 * NG + q) * NM + m];
                        sum += (r4[0] + r4[NG * NM]) + (r4[2 * NG * NM] + r4[3 * NG * NM]);
                    }
                    const size_t o = ((size_t)pcur * A.B + b) * A.nwork + (size_t)a * A.tiles + tile;
                    A.part[o * A.nm + m] = sum;
                    if (m == 0) A.partz[o] = sum;
                }
            }
            return;
        }
        double acc[NA];
#pragma unroll
        for (int m = 0; m < NA; ++m) acc[m] = 0.0;
        __syncthreads();                                                 // G rows and exp table ready
        GPMPC_STAMP(4);
        // (runs, NC = 256: a run may START left of this wave's diagonal block -- the wave enters at the block's first column, jst)
        const int jst = (NC > 64 && iw0 > j0) ? iw0 - j0 : 0;
        if (iw0 < Np && (NC > 64 ? j1 > iw0 : j0 >= iw0)) {              // tiles left of the wave's diagonal block carry no weight
            // the scratch slot was written by this workgroup a moment ago: take the address through an opaque asm so that the
            // constant-address-space loads below (scalar loads) cannot be moved in front of the stores and the barrier
            const double* Gl = Gs;
            asm volatile("" : "+s"(Gl) :: "memory");
            const __amdgpu_buffer_rsrc_t Mrs = Mrs_t;
            const int lane8 = lane * 8;
            typedef const double __attribute__((address_space(4))) gpmpc_cdouble;
            // one column: exponent from the wave-uniform row G[j] (scalar loads), table exp, weight, moments (pair_kernel_sb.h)
            auto column = [&](int j, double mij) {
                const gpmpc_cdouble* __restrict__ g = (const gpmpc_cdouble*)(Gl + (size_t)j * GW);
                double sx = qi + g[D];
#pragma unroll
                for (int k = 0; k < D; ++k) sx = fma(hi2[k], g[k], sx);
                const double P = mij * gpmpc_exp_neg_scaled(sx, s_tab);
                acc[0] += P;
                if (GRAD) {
#pragma unroll
                    for (int k = 0; k < D; ++k) acc[GRAD ? 1 + k : 0] = fma(P, g[k], acc[GRAD ? 1 + k : 0]);
#pragma unroll
                    for (int k = 0; k < NS2; ++k) acc[GRAD ? 1 + D + k : 0] = fma(P, g[D + 1 + k], acc[GRAD ? 1 + D + k : 0]);
                }
            };
            if constexpr (PIPE) {
                if constexpr (!EARLY) {
#pragma unroll
                    for (int q = 0; q < MG; ++q)
                        mga[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, q * Np * 8, 0));
                }
                static_assert(NC % (2 * MG) == 0, "two groups of columns per iteration");
                constexpr int PAIRW = GPMPC_FUSED_PIPE_ILP;
                double mgb[MG];
                for (int jc = 0; jc < NC; jc += 2 * MG) {
#pragma unroll
                    for (int q = 0; q < MG; ++q)                         // group jc + MG: in flight while group jc is evaluated
                        mgb[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jc + MG + q) * Np * 8, 0));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < MG; ++q) {
                        column(jc + q, mga[q]);
                        if (q % PAIRW == PAIRW - 1) __builtin_amdgcn_sched_barrier(0);      // PAIRW columns' dependency chains interleaved at a time
                    }
                    {   // group jc + 2 MG; UNCONDITIONAL (the last iteration re-requests its own first group, unused): under a
                        // branch the compiler's wait counts merge both paths and every wait below becomes "all loads done"
                        const int jn = jc + 2 * MG < NC ? jc + 2 * MG : jc;
#pragma unroll
                        for (int q = 0; q < MG; ++q)
                            mga[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jn + q) * Np * 8, 0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < MG; ++q) {
                        column(jc + MG + q, mgb[q]);
                        if (q % PAIRW == PAIRW - 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
            const int ncw = __builtin_amdgcn_readfirstlane(ncw_v);
            const int nct = j1 - j0 < NC ? j1 - j0 : NC;                      // (work list 7: the last tiles of the list are 16 columns wide)
            const int ncl = ncw - j0 < nct ? ncw - j0 : nct;                  // columns of this tile that carry weight (a multiple of 8, or <= 0)
#if GPMPC_FUSED_STAGED
            // Round 5 (as traj_persist.h): a wave of this loop is bound by its own latency chain, not by issue -- the compiler's schedule
            // of the four-column body waited for the G rows twice, for the exp table four times and for the weights of the SAME
            // iteration (no load in flight across iterations) per four columns; at D = 7 (C4, one trajectory: 834 cycles per column
            // against 145 of issue).  Columns are evaluated in batches of KC with the stages pinned: all G rows of the batch (one
            // scalar-load wait), all exponents and table reads (one LDS wait), then weights x exp and the moment sums, while the weight
            // loads of the next group of columns are in flight (unconditional refill, two register sets in alternation).
            constexpr int MGS = GPMPC_FUSED_MGS, KC = (D >= 8 || (MGS > 2 && D >= 7)) ? 1 : 2;     // (MGS = 4: an A/B option, more weight loads in flight)
            static_assert(8 % (2 * MGS) == 0 && MGS % KC == 0, "tiles carry multiples of 8 columns");
            double wa[MGS], wb[MGS];
#pragma unroll
            for (int q = 0; q < MGS; ++q) wa[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jst + q) * Np * 8, 0));
            auto batch = [&](int j, const double* mw) {
                double g[KC][GW];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const gpmpc_cdouble* __restrict__ gp = (const gpmpc_cdouble*)(Gl + (size_t)(j + c) * GW);
#pragma unroll
                    for (int k = 0; k < D + 1 + NS2; ++k) g[c][k] = gp[k];
                }
                __builtin_amdgcn_sched_barrier(0);
                double fr[KC], pq[KC], Tv[KC];
                int ni[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    double sx = qi + g[c][D];
#pragma unroll
                    for (int k = 0; k < D; ++k) sx = fma(hi2[k], g[c][k], sx);
                    const double ax = __builtin_fabs(sx);            // gpmpc_exp_neg_scaled (fast_exp.h), split around the table read
                    ni[c] = (int)(-ax);
                    fr[c] = __builtin_amdgcn_fract(ax);
                    Tv[c] = s_tab[ni[c] & (GPMPC_EXP_N - 1)];
                    pq[c] = fma(fr[c], fma(fr[c], GPMPC_EXP_A3, GPMPC_EXP_A2), GPMPC_EXP_A1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const double P = mw[c] * ldexp(fma(Tv[c] * fr[c], pq[c], Tv[c]), ni[c] >> GPMPC_EXP_BITS);
                    acc[0] += P;
                    if (GRAD) {
#pragma unroll
                        for (int k = 0; k < D; ++k) acc[GRAD ? 1 + k : 0] = fma(P, g[c][k], acc[GRAD ? 1 + k : 0]);
#pragma unroll
                        for (int k = 0; k < NS2; ++k) acc[GRAD ? 1 + D + k : 0] = fma(P, g[c][D + 1 + k], acc[GRAD ? 1 + D + k : 0]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            for (int jc = jst; jc < ncl; jc += 2 * MGS) {
#pragma unroll
                for (int q = 0; q < MGS; ++q)
                    wb[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jc + MGS + q) * Np * 8, 0));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < MGS; q += KC) batch(jc + q, &wa[q]);
                {
                    const int jn = jc + 2 * MGS < ncl ? jc + 2 * MGS : jc;
#pragma unroll
                    for (int q = 0; q < MGS; ++q)
                        wa[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jn + q) * Np * 8, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < MGS; q += KC) batch(jc + MGS + q, &wb[q]);
            }
#else
            constexpr int CU = 4;
            for (int jc = jst; jc < ncl; jc += CU) {
                double mij[CU];
#pragma unroll
                for (int q = 0; q < CU; ++q)
                    mij[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jc + q) * Np * 8, 0));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < CU; ++q) column(jc + q, mij[q]);
            }
#endif
            }
        }
        GPMPC_STAMP(5);
        {
            double z[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) z[m] = 0.0;
            const double rs = acc[0];
            z[0] = rs;
            if (GRAD) {
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double h = (0.5 / GPMPC_EXP_NEG_INV_C) * hi2[k], v = acc[GRAD ? 1 + k : 0];
                    z[GRAD ? 1 + k : 0] = fma(h, rs, v);
                    if (k < NS2) z[GRAD ? 1 + D + k : 0] = fma(h * h, rs, fma(2.0 * h, v, acc[GRAD ? 1 + D + (k < NS2 ? k : 0) : 0]));
                }
            }
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const double sr = wave_row_sum(z[m]);
                if ((lane & 15) == 0) s_red[(w * 4 + (lane >> 4)) * NM + m] = sr;
            }
        }
        __syncthreads();
        GPMPC_STAMP(6);
        if (tid < NM) {
            double sum = 0.0;                                            // fixed order: waves, each as (row 0 + row 1) + (row 2 + row 3)
            for (int ww = 0; ww < 4; ++ww) {
                const double* r4 = &s_red[ww * 4 * NM + tid];
                sum += (r4[0] + r4[NM]) + (r4[2 * NM] + r4[3 * NM]);
            }
            A.part[(((size_t)pcur * A.B + b) * A.nwork + bx) * A.nm + tid] = sum;
            if (tid == 0) A.partz[((size_t)pcur * A.B + b) * A.nwork + bx] = sum;
        }
        GPMPC_STAMP(7);
        GPMPC_STAMP_REAL(13);
        GPMPC_STAMP_WG(1);
        return;
    }
    }

    if (role == 0 && !SB) {
        // ---- piece of a tile of the N^2 sum of step t (staged form of pair_kernel.h: diagonal S, one trajectory) ------------
        double hi[D], sck[D], cv[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { sck[k] = s_sck[k]; cv[k] = s_cv[k]; hi[k] = fma(-sck[k], xrow[k], cv[k]); }
        double acc[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m] = 0.0;
        const double* __restrict__ Ma = A.M + (size_t)unit * Np * Np;
        bool first = true;
        for (int jc = j0; jc < j1; jc += 64) {
            if (jc + 63 < i0 || jc + jq >= __builtin_amdgcn_readfirstlane(ncw_v)) continue;      // upper-triangular M: nothing left of the diagonal chunk; nothing beyond N
            if (!first) __syncthreads();                              // the previous chunk's s_hj has been consumed
            if (tid < NCOL) {
                double x[D];
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = first ? xcol[k] : A.XT[(size_t)k * Np + jc + jq + tid];
#pragma unroll
                for (int k = 0; k < D; ++k) s_hj[tid * DP + k] = fma(-sck[k], x[k], cv[k]);
            }
            if (!first) {                                             // the first chunk's weights were fetched in phase 0
                const double* __restrict__ Mc = Ma + (size_t)(jc + jq) * Np + i0 + lane;
#pragma unroll
                for (int q = 0; q < CW; ++q) mpre[q] = Mc[(size_t)(w * CW + q) * Np];
            }
            first = false;
            __syncthreads();
            GPMPC_STAMP(4);
#pragma unroll GPMPC_FUSED_UNROLL
            for (int q = 0; q < CW; ++q) {
                const double* hj = &s_hj[(w * CW + q) * DP];
                double m[D], sq[D];
#pragma unroll
                for (int k = 0; k < D; ++k) { m[k] = hi[k] + hj[k]; sq[k] = m[k] * m[k]; }
                double s = sq[0];
#pragma unroll
                for (int k = 1; k < D; ++k) s += sq[k];
                const double P = mpre[q] * (GPMPC_FUSED_TABLE ? gpmpc_exp_neg(s, s_tab) : exp(-s));
                acc[0] += P;
                if (GRAD) {
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        acc[GRAD ? 1 + k : 0] = fma(P, m[k], acc[GRAD ? 1 + k : 0]);
                        if (k < NS2) acc[GRAD ? 1 + D + k : 0] = fma(P, sq[k], acc[GRAD ? 1 + D + k : 0]);
                    }
                }
            }
        }
        GPMPC_STAMP(5);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double s = wave_row_sum(acc[m]);                   // rows of 16 lanes; the combine below adds 16 values
            if ((lane & 15) == 0) s_red[(w * 4 + (lane >> 4)) * NM + m] = s;
        }
        __syncthreads();
        GPMPC_STAMP(6);
        if (tid < NM) {
            double s = 0.0;
            for (int ww = 0; ww < 4; ++ww) {
                const double* r4 = &s_red[ww * 4 * NM + tid];
                s += (r4[0] + r4[NM]) + (r4[2 * NM] + r4[3 * NM]);
            }
            A.part[(((size_t)pcur * A.B + b) * A.nwork + bx) * A.nm + tid] = s;
            if (tid == 0) A.partz[((size_t)pcur * A.B + b) * A.nwork + bx] = s;
        }
        GPMPC_STAMP(7);
        GPMPC_STAMP_REAL(13);
        GPMPC_STAMP_WG(1);
        return;
    }

    const int a = am;
    if (role == 2) {
        // ---- GP a: outputs and Jacobian rows of step t-1 (step.hip::finish_step) ---------------------------------------
        if (t == 1) {
            if (a == 0 && tid < DS) {
                A.means[((size_t)b * (A.H + 1)) * DS + tid] = s_mu[tid];
                A.vars[((size_t)b * (A.H + 1)) * DS + tid] = GPMPC_INIT_VAR;
            }
            return;
        }
        if (tid == 0) {
            A.means[((size_t)b * (A.H + 1) + (t - 1)) * DS + a] = s_mu[a];
            A.vars[((size_t)b * (A.H + 1) + (t - 1)) * DS + a] = s_var[a];
        }
        if (!GRAD) return;
        // moments 1..NM-1 of GP a: thread = (moment, channel), 16 channels stride the items; fixed-order combine
        const int ch = tid & 15;
        for (int m = tid >> 4; m < NM; m += NT / 16) {
            const double* p = A.part + (((size_t)pprev * A.B + b) * A.nwork) * A.nm + m;
            double s = 0.0;
            for (int wi = A.ustart[a] + ch; wi < A.ustart[a + 1]; wi += 16) s += p[(size_t)wi * A.nm];
            s_red[m * 16 + ch] = s;
        }
        GPMPC_STAMP(4);
        __syncthreads();
        if (tid < NM) {
            double s = 0.0;
            for (int c = 0; c < 16; ++c) s += s_red[tid * 16 + c];
            s_out[tid] = s;
        }
        __syncthreads();
        GPMPC_STAMP(5);
        if (tid < D) {
            const int k = tid, nc = 2 * DS + DA;
            const double c = s_c[a], mu = s_mu[a], T = c * s_z0[a];  // the Z0 sum every workgroup of this launch uses
            const double Ak = s_spp[k], sc = s_spp[D + k], dmu_du = s_spp[2 * D + k], dmu_ds = s_spp[3 * D + k];
            const double dT_du = -4.0 * sc * c * s_out[GRAD ? 1 + k : 0];
            const double dv_du = -dT_du - 2.0 * mu * dmu_du;
            double* jm = A.jac + (((size_t)b * A.H + (t - 2)) * 2 * DS + a) * nc;          // row of mu_a (step t-1)
            double* jv = A.jac + (((size_t)b * A.H + (t - 2)) * 2 * DS + DS + a) * nc;     // row of var_a
            if (k < DS) {
                const double dT_ds = Ak * (c * s_out[GRAD ? 1 + D + (k < NS2 ? k : 0) : 0] - 0.5 * T);
                const double dv_ds = -dT_ds - 2.0 * mu * dmu_ds;
                jm[k] = dmu_du; jm[DS + k] = dmu_ds;
                jv[k] = dv_du;  jv[DS + k] = dv_ds;
            } else {            // action input: its variance is a constant
                jm[2 * DS + (k - DS)] = dmu_du;
                jv[2 * DS + (k - DS)] = dv_du;
            }
        }
        GPMPC_STAMP(6);
        GPMPC_STAMP_WG(1);
        return;
    }

    // ---- GP a: mean sums of step t over the N points (step.hip::prep_step without the pair-kernel parameters) ---------
    __shared__ double s_B[D], s_A[D], s_r1[D], s_r2[D];
    if (tid < D) {
        const int k = tid;
        const double sk = s_sin[k], lam = lam_k;                    // wave 0 holds the length-scales
        s_B[k] = 1.0 / (sk + lam);
        s_A[k] = 1.0 / (0.5 * lam + sk);
        s_r1[k] = sk / lam + 1.0;
        s_r2[k] = 2.0 * sk / lam + 1.0;
    }
    __syncthreads();
    GPMPC_STAMP(4);
    double u[D], Bk[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { u[k] = s_uin[k]; Bk[k] = s_B[k]; }
    double v[NV];
#pragma unroll
    for (int m = 0; m < NV; ++m) v[m] = 0.0;
#pragma unroll
    for (int r = 0; r < PF; ++r) {                                 // prefetched points (zero weight past Np)
        double d[D], q = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { d[k] = u[k] - xpt[r][k]; q = fma(Bk[k] * d[k], d[k], q); }
        const double p = bpt[r] * exp(-0.5 * q);
        v[0] += p;
#pragma unroll
        for (int k = 0; k < D; ++k) { v[1 + k] = fma(p, d[k], v[1 + k]); v[1 + D + k] = fma(p * d[k], d[k], v[1 + D + k]); }
    }
    // the points beyond: UN per round trip (all loads first; a large training set otherwise walks Np / 256 dependent round trips while
    // the tile workgroups stream -- N = 2048: this workgroup lived as long as a tile workgroup, profiles/r03/fused_wg_timeline.txt).
    // Points past the end get zero weight: the same sums in the same order as one point at a time.
    constexpr int UN = SB ? GPMPC_FUSED_MEAN_UNROLL : 1;
    for (int ib = tid + PF * NT; ib < Np; ib += UN * NT) {
        double xq[UN][D], bq[UN];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = ib + r * NT, ic = i < Np ? i : 0;
#pragma unroll
            for (int k = 0; k < D; ++k) xq[r][k] = A.XT[(size_t)k * Np + ic];
            const double bv = A.beta[(size_t)a * Np + ic];
            bq[r] = i < Np ? bv : 0.0;
        }
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            double d[D], q = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) { d[k] = u[k] - xq[r][k]; q = fma(Bk[k] * d[k], d[k], q); }
            const double p = bq[r] * exp(-0.5 * q);
            v[0] += p;
#pragma unroll
            for (int k = 0; k < D; ++k) { v[1 + k] = fma(p, d[k], v[1 + k]); v[1 + D + k] = fma(p * d[k], d[k], v[1 + D + k]); }
        }
    }
    GPMPC_STAMP(5);
    block_sum4_rows<NV>(v, s_red, s_out);
    GPMPC_STAMP(6);
    if (tid < D) {
        const int k = tid;
        const double sf2 = sf_a * sf_a;
        double detm = 1.0, detv = 1.0;
        for (int l = 0; l < D; ++l) { detm *= s_r1[l]; detv *= s_r2[l]; }
        const double cm = sf2 / sqrt(detm), c = 1.0 / sqrt(detv);
        const double mu = cm * s_out[0];
        double* sp = A.sp + (((size_t)pcur * A.B + b) * DS + a) * A.sps;
        if (k == 0) { sp[0] = c; sp[1] = mu; sp[2] = sf2; }
        const double Bq = s_B[k];
        sp[3 + k] = s_A[k]; sp[3 + D + k] = s_sck[k];
        sp[3 + 2 * D + k] = -Bq * cm * s_out[1 + k];
        sp[3 + 3 * D + k] = -0.5 * mu * Bq + 0.5 * Bq * Bq * cm * s_out[1 + D + k];
    }
    GPMPC_STAMP(7);
    GPMPC_STAMP_WG(1);
}

template <int D, int NS2, bool GRAD, int Q, int NG = 1>
static int launch_step_fused_one(const FusedArgs& a, int t, hipStream_t s) {
    if ((Q == 0 || Q == 32 || Q == 16 || Q == 256) && !a.gscr) return GPMPC_E_ARG;
    if (a.ntile < 1 || (NG > 1 && a.tiles < 1)) return GPMPC_E_ARG;
    hipLaunchKernelGGL((k_step_fused<D, NS2, GRAD, Q, NG>), dim3(a.ntile + 2 * NS2, a.B), dim3(256), 0, s, a, t);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("fused step kernel launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

// ns2 = state_dim: D - ns2 in {0, 1, 2} action dimensions; q = 1 (whole tiles), 4 (a quarter of a tile's columns per workgroup)
// or 0 / 32 / 16 (256 x 64 / 32 / 16 tiles, scalar-broadcast column loop: the mid-size form)
template <int D>
int gpmpc_launch_step_fused_D(bool grad, int ns2, int q, int ng, const FusedArgs& a, int t, hipStream_t s) {
    if (a.nm != (grad ? 1 + 2 * D : 1) || (q != 0 && q != 1 && q != 4 && q != 16 && q != 32 && q != 256)) return GPMPC_E_ARG;
    if (ng > 1) {                                       // one lambda for all GPs: groups of ng GPs per tile workgroup (q = 0 only)
        if (q != 0) return GPMPC_E_ARG;
#define GPMPC_FUSED_SH(NGV, NSV)                                                                                   \
        if constexpr (NGV <= NSV) if (ng == NGV && ns2 == NSV)                                                     \
            return grad ? launch_step_fused_one<D, NSV, true, 0, NGV>(a, t, s) : launch_step_fused_one<D, NSV, false, 0, NGV>(a, t, s);
#define GPMPC_FUSED_SH_NG(NGV)                                                                                     \
        if constexpr (D >= 2) { GPMPC_FUSED_SH(NGV, (D >= 2 ? D - 1 : 1)) }                                        \
        if constexpr (D >= 3) { GPMPC_FUSED_SH(NGV, (D >= 3 ? D - 2 : 1)) }
        GPMPC_FUSED_SH_NG(2)
        GPMPC_FUSED_SH_NG(3)
        GPMPC_FUSED_SH_NG(4)
#undef GPMPC_FUSED_SH_NG
#undef GPMPC_FUSED_SH
        return GPMPC_E_ARG;
    }
#define GPMPC_FUSED_CASE(GR, QV)                                                                                   \
    if (grad == GR && q == QV) {                                                                                   \
        if (ns2 == D) return launch_step_fused_one<D, D, GR, QV>(a, t, s);                                         \
        if (D >= 2 && ns2 == D - 1) return launch_step_fused_one<D, (D >= 2 ? D - 1 : D), GR, QV>(a, t, s);        \
        if (D >= 3 && ns2 == D - 2) return launch_step_fused_one<D, (D >= 3 ? D - 2 : D), GR, QV>(a, t, s);        \
        return GPMPC_E_ARG;                                                                                        \
    }
    GPMPC_FUSED_CASE(true, 1)
    GPMPC_FUSED_CASE(true, 4)
    GPMPC_FUSED_CASE(false, 1)
    GPMPC_FUSED_CASE(false, 4)
    GPMPC_FUSED_CASE(true, 0)
    GPMPC_FUSED_CASE(false, 0)
    GPMPC_FUSED_CASE(true, 32)
    GPMPC_FUSED_CASE(false, 32)
    GPMPC_FUSED_CASE(true, 16)
    GPMPC_FUSED_CASE(false, 16)
    if constexpr (D >= 6) {                             // (runs: work list 7 exists from D = 6, pack.hip)
        GPMPC_FUSED_CASE(true, 256)
        GPMPC_FUSED_CASE(false, 256)
    }
#undef GPMPC_FUSED_CASE
    return GPMPC_E_ARG;
}
