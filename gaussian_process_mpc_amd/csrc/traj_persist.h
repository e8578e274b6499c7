// Whole-horizon rollout of ONE trajectory by ONE workgroup: the form for LARGE batches of a SMALL training set (the N = 200...400
// points the reference's experiments run with, evaluated for hundreds of candidate plans at once: sampling / multi-start MPC).
//
// There a horizon step of the batch is a few tens of microseconds of arithmetic, and the step-per-launch forms spend more than that
// around it: the head kernel in front of every pair launch, the tail of partly filled workgroup generations, the launch boundaries
// and the round trip of the per-tile partial sums through global memory (N = 300, ds = 4, B = 256: 124 us per step, ~57 us of it pair
// arithmetic -- profiles/r03/batch_size_map.txt).  Trajectories never depend on each other, so a workgroup can keep
// one for all H steps: every dependency of the recursion (src/dynamics.py:152-189) is then inside the workgroup and costs a
// __syncthreads, nothing is handed between workgroups, and a rollout is ONE launch (+ the tail kernel: cost and reverse sweep).
//
//   grid = B workgroups x (64 NW) threads, NW = 16 (or 8: two trajectories per CU)
//   per horizon step t, all inside the workgroup:
//     1  threads (a, k): the per-GP, per-dimension scalars of the step -- B_k, A_k, the pair transform sc_k, c_m, c
//        (src/tools/uncertainty_prop.py:329-337, :374-377; closed forms at the top of step.hip)
//     2  column rows G[a][j] = [h_j | q_j N/ln2 | h_jk^2] of every (GP, point) -> this workgroup's scratch slot in global memory (read
//        back through SCALAR loads in 4, as step_fused.h does); the O(N) mean sums of every GP (a group of NW / ds waves per GP)
//     3  the N^2 sum: the (GP, row block, column) space of the upper triangle, flattened, is cut into NW equal contiguous ranges, one
//        per wave; a wave walks its range with the scalar-broadcast column loop of pair_kernel_sb.h (software-pipelined weight loads),
//        re-deriving the row side whenever the range enters a new (GP, row block); moments are summed per lane across the row blocks of
//        a GP and reduced once per GP (a range touches at most two GPs)
//     4  fixed-order combine of the waves' partial sums; mean, variance and the (2 ds) x (2 ds + da) step Jacobian of every GP
//        (step.hip::finish_step); the outputs are the next step's input moments
//   The training inputs X live in LDS for the whole kernel, M (a few MB) is L2 / Infinity-Cache resident across the workgroups.
// Same expressions as the step-per-launch forms (step.hip, step_fused.h, pair_kernel_sb.h); summation orders differ, results agree to
// rounding.  Static ranges, partial sums written (not atomically added), fixed-order combines: bit-reproducible run to run.
#pragma once
#include "gpmpc_internal.h"
#include "fast_exp.h"
#include <type_traits>


// Diagnostic build (-DGPMPC_PERSIST_STAMPS, D = GPMPC_STAMP_D instances): s_memtime stamps of workgroup 0 at horizon step 3 -- per
// phase by thread 0, and the column-loop interval of every wave -- read back with gpmpc_debug_persist_stamps (tools/persist_stamps.py).
#if defined(GPMPC_PERSIST_STAMPS)
static __device__ unsigned long long g_persist_stamps[64];
#define GPMPC_PST(slot) do { if (t == 3 && blockIdx.x == 0 && threadIdx.x == 0) g_persist_stamps[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#define GPMPC_PSTW(slot) do { if (t == 3 && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_persist_stamps[(slot) + w] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GPMPC_PST(slot) do { } while (0)
#define GPMPC_PSTW(slot) do { } while (0)
#endif
#ifndef GPMPC_PERSIST_MG
#define GPMPC_PERSIST_MG 0        // 0: by D (below)
#endif
#ifndef GPMPC_PERSIST_ILP
#define GPMPC_PERSIST_ILP 0       // 0: by D (below)
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's GLOBAL memory operations (s_waitcnt vmcnt(0)): in
// this kernel that put the write latency of the step's Jacobian rows -- read by nobody before the kernel ends -- and of the weight /
// beta prefetches in front of every barrier (6.7 us of an 84 us step in the in-kernel timeline).  Where waves exchange data through
// GLOBAL memory (the column rows) the drain is explicit.
#ifdef GPMPC_PERSIST_SYNCTHREADS
#define GPMPC_LDS_BARRIER() __syncthreads()
#else
#define GPMPC_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
#ifndef GPMPC_PERSIST_CSEG
#define GPMPC_PERSIST_CSEG 16       // overhead of one (unit, row block) segment of a wave's range, in columns (range balancing)
#endif
#ifndef GPMPC_PERSIST_KC3
#define GPMPC_PERSIST_KC3 1         // batches of three columns where the rows fit the SGPR file (see KC)
#endif
#ifndef GPMPC_PERSIST_TOUCH
#define GPMPC_PERSIST_TOUCH 0       // 1: request the next batch's G-row cache lines with this batch's rows (scalar-cache warm-up; A/B)
#endif
#ifndef GPMPC_PERSIST_AGEW
#define GPMPC_PERSIST_AGEW 4        // range weights by wave age, per cent per age group (see the range table)
#endif
#ifndef GPMPC_PERSIST_PRIO_SHIFT
#define GPMPC_PERSIST_PRIO_SHIFT 5  // the issue priority rotates every 2^shift columns
#endif
#ifndef GPMPC_PERSIST_ROTPRIO
#define GPMPC_PERSIST_ROTPRIO 2     // 0: none; 1: by the wave's column count; 2: by the shader clock
#endif
#ifndef GPMPC_PERSIST_CLK_SHIFT
#define GPMPC_PERSIST_CLK_SHIFT 14  // the issue priority rotates every 2^shift cycles (measured: 12 x1.19, 14 x1.23-1.28 over the round-4 kernel at N = 300, ds = 4)
#endif
#define GPMPC_PERSIST_MAXDEV 64
#ifndef GPMPC_PERSIST_MAXNP
#define GPMPC_PERSIST_MAXNP 1024                   // X in LDS: Np * D doubles (57 KB at D = 7)
#endif

// NG > 1: every GP has the SAME length-scales (pair_kernel_sbs.h: the setting of all the reference's experiments).  The transformed
// points, the exponent and its exp are then common to the GPs of a (trajectory, step): a wave's range runs over UNITS of NG GPs, one
// set of column rows per trajectory, one exponent and one table exp per pair applied to NG weight loads -- (1 + D + 7) / NG + 2 + D + ds
// fp64-rate instructions per pair and GP (17.5 at D = 5, ds = 4, NG = 2, against 24).  A partial last unit (odd ds) re-reads the last GP;
// its results are not combined.
template <int D, int NS2, bool GRAD, int NG = 1>
__global__ __launch_bounds__(1024, 4) void k_traj_persist(PersistArgs A) {
    constexpr bool SH = NG > 1;
    constexpr int DS = NS2, DA = D - NS2, NM = GRAD ? 1 + 2 * D : 1, NA = GRAD ? 1 + D + NS2 : 1, NV = 1 + 2 * D;
    constexpr int GW = (D + 1 + NS2 + 1) & ~1;     // doubles per G row, as PairSbTraits
    // columns per batch of the column loop (KC (D + 1 + ds) doubles of G rows in SGPRs, KC sets of exp temporaries in VGPRs) and per group of
    // weight loads (two groups in flight): what fits the 128 registers of four waves per SIMD WITHOUT a spill (tools/spill_guard.py).
    // Range boundaries fall on multiples of GR = 2 MGc columns (the loop's step): 8 for NG = 1, down to 2 for units of four GPs, whose
    // flattened space is a quarter as long (N = 300: 55 columns per wave).
    // KC = 3 (one GP per unit, D <= 5: 3 (D + 1 + ds) <= 30 doubles of rows in SGPRs): with batches of two a wave waits ~650 cycles (rows + table)
    // per 230 cycles of its own arithmetic and its three neighbours on the SIMD have 690 to fill them with -- just enough only when
    // perfectly staggered (VALU busy ~78 % of the loop phase); batches of three leave 1035.  Six columns per iteration, a tail of batches of
    // two (segment lengths are even, not multiples of six).
    constexpr int KC = (D >= 7 || (NG > 1 && D >= 6) || NG >= 4 || (NG == 3 && D >= 5)) ? 1 : ((NG == 1 && D <= 5 && GPMPC_PERSIST_KC3) ? 3 : 2);
    constexpr int MGc = KC == 3 ? 3 : (GPMPC_PERSIST_MG ? GPMPC_PERSIST_MG : (NG >= 4 ? 1 : ((NG > 1 || D >= 6) ? 2 : 4)));
    constexpr int GR = KC == 3 ? 2 : 2 * MGc;
    constexpr bool GPAIR = KC != 3;                // column rows laid out per pair of columns (batches of two read them as one block); batches of three: row-major
    static_assert(MGc % KC == 0 && 8 % GR == 0, "segment lengths are multiples of GR columns; row blocks and N (rounded up) of 8");
    // weight columns per group (two groups in flight) and columns whose dependency chains may interleave: 4 and 2 up to D = 6; from D = 7
    // the accumulators alone (2 (1 + D + ds) + 2 (1 + 2 D) registers) leave room for 2 and 1 (D = 7: 68 -> 6 VGPR spills, D = 8: 94 -> 18).
    // Not only speed: the D = 8, ds = 7 instance with 94 VGPR + 80 SGPR spills returned means with the low mantissa word of some
    // double lost (1e-6 relative, deterministic, the objective-only instance and every lighter one exact) -- a spill-path miscompile;
    // tests/test_gpu_instances.py holds every (ds, da) instance to the C port.
    extern __shared__ double s_dyn[];              // X: [D][Np]
    __shared__ double s_tab[GPMPC_EXP_N];
    __shared__ double s_part[16 * 2 * NG * 4 * NM];    // [wave][first | second unit of the wave's range][GP of the unit][row of 16 lanes][moment]
    __shared__ double s_mred[16 * 4 * NV];         // mean sums: [wave][row of 16 lanes][value]
    __shared__ double s_z[DS * NM], s_ms[DS * NV];
    __shared__ double s_uin[D], s_sin[D];
    __shared__ double s_B[DS * D], s_Ak[DS * D], s_sc[DS * D], s_cv[DS * D], s_r1[DS * D], s_r2[DS * D];
    __shared__ double s_c[DS], s_cm[DS], s_sf2[DS], s_lam[DS * D], s_avar;
    __shared__ int s_rng[17], s_ga[16];            // range boundaries of the waves in the flattened column space; GP a range starts in
    const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), NW = nthr >> 6;
    const int Np = A.Np, T = Np >> 6;
    // Columns >= N carry zero weights: the column space of row block r ends at Nc = N rounded up to 8, not at the padded size -- for the
    // training-set sizes of the reference's experiments that is much of the loop (N = 200: Np = 256, 224 of 640 column steps per GP; N = 300: 8 %;
    // N = 400: 19 %).  Nc is read from the pack (device memory, refreshed by every pack build: a captured launch stays valid when the
    // training set grows within its padded size).  start(r) = sum_{r' < r} (Nc - 64 r') = r Nc - 32 r (r - 1).
#if defined(GPMPC_PERSIST_NO_CLIP)
    const int Nc = Np;                             // A/B: the padded column space of the first version
#else
    const int Nc = __builtin_amdgcn_readfirstlane(*A.ncol);
#endif
    const int per_gp = T * Nc - 32 * T * (T - 1);  // columns of one GP
    const int total = ((DS + NG - 1) / NG) * per_gp;
    double* __restrict__ s_X = s_dyn;
    double* __restrict__ s_U = s_dyn + (size_t)D * Np;       // the trajectory's actions, [H][da]
    double* __restrict__ s_stage = s_U + ((A.H * DA + 1) & ~1);     // row-major column rows: a staging block of 64 rows per row-writing wave (16-byte aligned)
    (void)s_stage;

    gpmpc_exp_table_to_lds(s_tab);
    for (int e = tid; e < D * Np; e += nthr) s_X[e] = A.XT[e];
    for (int e = tid; e < A.H * DA; e += nthr) s_U[e] = A.U[(size_t)b * A.H * DA + e];
    // Range boundaries of the waves, on multiples of 8 columns (row blocks start on multiples of 64); computed once.  Round 5: equal
    // COST, not equal columns -- every (unit, row block) segment a wave touches costs it a fixed overhead on top of its columns (row
    // side, first weight loads and G rows with nothing in flight, the moment flush: the staged loop's timeline had the waves with
    // three segments finish 30 % after those with one, profiles/r05/persist_stamps_staged_loop.txt), priced as CSEG columns per
    // row-block start: boundary k is where the cumulative cost pos + CSEG x (row blocks started before pos) reaches k / NW of its total.
    if (tid <= NW) {
        const int nun = (DS + NG - 1) / NG, nblk = nun * T;
        const long ctot = (long)total + (long)GPMPC_PERSIST_CSEG * nblk;
        // ... and equal TIME, not equal cost: the four waves a SIMD holds are served oldest first whatever their priority rotation --
        // with equal ranges wave w finished after (116, 120, 125, 132) k cycles for w / 4 = 0 ... 3 (profiles/r05/persist_stamps_phase2.txt)
        // -- so the older waves take proportionally more: weights 1 + AGEW (1.5 - w / 4) / 100 per wave, cumulative below.
        long target = 0;
        {
            long wsum = 0, wacc = 0;
            for (int k = 0; k < NW; ++k) {
                const long wk = 200 + GPMPC_PERSIST_AGEW * ((NW / 4 - 1) - 2 * (k >> 2));       // 2 (100 + AGEW ((NW / 4 - 1) / 2 - group)), group = k / 4: the wave's age rank on its SIMD
                wsum += wk;
                if (k < tid) wacc += wk;
            }
            target = ctot * wacc / wsum;
        }
        int lo = total;
        if (tid < NW) {
            long before = 0;                                // cost of everything before block (u, r)
            lo = 0;
            for (int k = 0; k < nblk; ++k) {
                const int u = k / T, r = k - u * T;
                const int start = u * per_gp + r * Nc - 32 * r * (r - 1), len = Nc - 64 * r;
                if (target <= before + GPMPC_PERSIST_CSEG) { lo = start; break; }
                if (target < before + GPMPC_PERSIST_CSEG + len) { lo = start + (int)((target - before - GPMPC_PERSIST_CSEG + GR / 2) & ~(long)(GR - 1)); break; }
                before += GPMPC_PERSIST_CSEG + len;
                lo = total;
            }
            if (tid == 0) lo = 0;
        }
        s_rng[tid] = lo;
        if (tid < NW) s_ga[tid] = lo < total ? lo / per_gp : nun;
    }
    if (tid == 0) s_avar = GPMPC_ACTION_VAR;                   // (a constant read from LDS per step: held in a register pair it was a spill)
    if (tid < DS * D) s_lam[tid] = A.lam[tid];             // (read per step from LDS: a register held across the step loop is a spill)
    if (tid < DS) { s_uin[tid] = A.x0[(size_t)b * DS + tid]; s_sin[tid] = GPMPC_INIT_VAR; }
    if (tid < DS) {
        A.means[((size_t)b * (A.H + 1)) * DS + tid] = s_uin[tid];
        A.vars[((size_t)b * (A.H + 1)) * DS + tid] = GPMPC_INIT_VAR;
    }
    double* __restrict__ Gs = A.gscr + (size_t)b * DS * Np * GW;

    for (int t = 1; t <= A.H; ++t) {
        // Everything below that depends only on the thread index (row / GP of a thread in each phase, addresses, an integer division) is
        // loop-invariant over t: hoisted out of the step loop it was ~60 VGPRs of live state, spilled to scratch and reloaded in every
        // phase of every step.  A zero the compiler cannot prove (t >> 30) makes the index step-dependent: recomputed (a few integer
        // instructions) instead.  (An inline-asm-laundered zero did the same but MISCOMPILED the ds = 7 instance -- means off by 1e-3 from
        // the first step on, the NO_TZ build of the same source exact --: kept out.)
#ifndef GPMPC_PERSIST_NO_TZ
        const int tz = t >> 30;                                    // 0 (1 <= t <= H < 2^30), but not provably
#else
        const int tz = 0;
#endif
        // (the lane index too: mbcnt with a base the compiler cannot hoist, so that neither tid nor lane is held across the step loop)
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, tz)), tiz = w * 64 + ln, l8 = ln * 8;
        GPMPC_PST(0);
        // ---- 1: input moments of the action dimensions, then the per-(GP, dimension) scalars ------------------------------------
        if (DA > 0 && tiz >= DS && tiz < D) { s_uin[tiz] = s_U[(t - 1) * DA + (tiz - DS)]; s_sin[tiz] = s_avar; }
        GPMPC_LDS_BARRIER();
        if (tiz < DS * D) {
            const int a = tiz / D, k = tiz - a * D;
            const double lam = s_lam[tiz], sk = s_sin[k], uk = s_uin[k];
            s_B[tiz] = 1.0 / (sk + lam);
            s_Ak[tiz] = 1.0 / (0.5 * lam + sk);
            const double sc = rsqrt(8.0 * (0.5 * lam + sk));       // the pair transform h = sc (u - x), as step_fused.h
            s_sc[tiz] = sc;
            s_cv[tiz] = sc * uk;
            s_r1[tiz] = sk / lam + 1.0;
            s_r2[tiz] = 2.0 * sk / lam + 1.0;
        }
        GPMPC_LDS_BARRIER();
        GPMPC_PST(1);
        // ---- 2: the first DS waves take the O(N) mean sums, ONE wave per GP (step.hip::prep_step); the other waves write the column rows.
        // Round 5.  (a) The mean sums used to run on all waves, wpg per GP: every wave then paid the NV wave reductions (~130 of its
        // ~330 instructions, 16 waves x 4.4 cycles x 4 per SIMD = 5.8 k cycles of VALU per step); one wave per GP strides the points
        // with 64 lanes, reduces once, and the reductions of the DS waves run on different SIMDs beside the row writers.  The weights'
        // exp is the table exp of the pair loop (fast_exp.h: 1.3 ulp).  (b) The rows are laid out per PAIR of columns in 16-byte chunks,
        // G[a][j / 2][c][j % 2] = (g_2c, g_2c+1) of column j: two neighbouring lanes write one aligned 32-byte sector per store (row-major
        // rows scattered 64 x 16 bytes at an 80-byte stride -- partial sectors, ~5 k cycles per step), and the batch of two columns the
        // loop evaluates reads its 2 GW doubles as ONE contiguous block through scalar loads (a fully chunked layout G[a][c][j] made the
        // stores one contiguous kilobyte but cost the loop five 32-byte loads from five cache lines per batch: +6 % loop time).
        GPMPC_PST(2);
        // (two waves per GP when the workgroup has the waves to spare: the mean sums are the longer of the two roles)
        const int MW = (NW >= 16 && 2 * DS <= NW / 2) ? 2 : 1;
        if (w < MW * DS) {
            const int a = w / MW, wsub = w - a * MW, lstr = 64 * MW, l0 = wsub * 64 + ln;      // this wave's points: l0, l0 + lstr, ...
            double v[NV];
#pragma unroll
            for (int m = 0; m < NV; ++m) v[m] = 0.0;
            double u[D], Bk[D];
#pragma unroll
            for (int k = 0; k < D; ++k) { u[k] = s_uin[k]; Bk[k] = s_B[a * D + k]; }
            constexpr int PT = 8;                                  // weights of up to 512 (1024) points requested before the first is used
            double bpre[PT];
#pragma unroll
            for (int q = 0; q < PT; ++q) {
                const int i = l0 + lstr * q;
                const double bv = A.beta[(size_t)a * Np + (i < Np ? i : 0)];
                bpre[q] = i < Np ? bv : 0.0;
            }
#pragma unroll
            for (int q = 0; q < PT; ++q) {
                const int i = l0 + lstr * q;
                if (wsub * 64 + lstr * q < Np) {                   // (wave-uniform)
                    double d[D], qq = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) { d[k] = u[k] - s_X[k * Np + (i < Np ? i : 0)]; qq = fma(Bk[k] * d[k], d[k], qq); }
                    const double p = bpre[q] * gpmpc_exp_neg(0.5 * qq, s_tab);
                    v[0] += p;
#pragma unroll
                    for (int k = 0; k < D; ++k) { v[1 + k] = fma(p, d[k], v[1 + k]); v[1 + D + k] = fma(p * d[k], d[k], v[1 + D + k]); }
                }
            }
            for (int i = l0 + lstr * PT; i < Np; i += lstr) {      // larger training sets
                double d[D], qq = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) { d[k] = u[k] - s_X[k * Np + i]; qq = fma(Bk[k] * d[k], d[k], qq); }
                const double p = A.beta[(size_t)a * Np + i] * gpmpc_exp_neg(0.5 * qq, s_tab);
                v[0] += p;
#pragma unroll
                for (int k = 0; k < D; ++k) { v[1 + k] = fma(p, d[k], v[1 + k]); v[1 + D + k] = fma(p * d[k], d[k], v[1 + D + k]); }
            }
#pragma unroll
            for (int m = 0; m < NV; ++m) {                         // fixed order: rows of 16 lanes, then (0 + 1) + (2 + 3)
                const double sr = wave_row_sum(v[m]);
                if ((ln & 15) == 0) s_mred[(w * 4 + (ln >> 4)) * NV + m] = sr;
            }
            if (MW == 1 && ln < NV) {                              // (a wave's LDS operations execute in order)
                const double* r4 = &s_mred[w * 4 * NV + ln];
                s_ms[a * NV + ln] = (r4[0] + r4[NV]) + (r4[2 * NV] + r4[3 * NV]);
            }
        } else if constexpr (!GPAIR) {
            // row-major rows (batches of three columns read 3 GW contiguous doubles): written THROUGH LDS -- a lane's row goes to the wave's
            // staging block, the block (64 consecutive rows = 64 GW contiguous doubles in global memory) leaves as GW / 2 fully coalesced
            // kilobyte stores.  (Stored directly, a row is five 16-byte pieces at an 80-byte stride: partial sectors, +9 k cycles per step.)
            const int nrw = (NW - MW * DS) * 64, nrows = (SH ? 1 : DS) * Np;
            double2* stg = reinterpret_cast<double2*>(s_stage) + (size_t)(w - MW * DS) * 64 * (GW / 2);
            for (int e0 = (w - MW * DS) * 64; e0 < nrows; e0 += nrw) {               // (wave-uniform)
                const int e = e0 + ln;
                if (e < nrows) {
                    const int a = SH ? 0 : e / Np, j = e - a * Np;
                    double g[GW], qh = 0.0;
#pragma unroll
                    for (int k = 0; k < GW; ++k) g[k] = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double h = fma(-s_sc[a * D + k], s_X[k * Np + j], s_cv[a * D + k]);
                        g[k] = h;
                        qh = fma(h, h, qh);
                        if (k < NS2) g[D + 1 + k] = h * h;
                    }
                    g[D] = GPMPC_EXP_NEG_INV_C * qh;
#pragma unroll
                    for (int c = 0; c < GW / 2; ++c) stg[ln * (GW / 2) + c] = make_double2(g[2 * c], g[2 * c + 1]);
                }
                const int valid = (nrows - e0 < 64 ? nrows - e0 : 64) * (GW / 2);    // 16-byte chunks of the block (a wave's LDS operations execute in order)
#pragma unroll
                for (int c = 0; c < GW / 2; ++c) {
                    const int idx = c * 64 + ln;
                    if (idx < valid) reinterpret_cast<double2*>(Gs)[(size_t)e0 * (GW / 2) + idx] = stg[idx];
                }
            }
        } else {
            const int nrw = (NW - MW * DS) * 64;                   // row-writing threads
            for (int e = tiz - MW * DS * 64; e < (SH ? 1 : DS) * Np; e += nrw) {     // one set of rows per trajectory when the GPs share lambda
                const int a = SH ? 0 : e / Np, j = e - a * Np;
                double g[GW], qh = 0.0;
#pragma unroll
                for (int k = 0; k < GW; ++k) g[k] = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double h = fma(-s_sc[a * D + k], s_X[k * Np + j], s_cv[a * D + k]);
                    g[k] = h;
                    qh = fma(h, h, qh);
                    if (k < NS2) g[D + 1 + k] = h * h;
                }
                g[D] = GPMPC_EXP_NEG_INV_C * qh;
#pragma unroll
                for (int c = 0; c < GW / 2; ++c)
                    reinterpret_cast<double2*>(Gs)[(((size_t)a * (Np >> 1) + (j >> 1)) * (GW / 2) + c) * 2 + (j & 1)] = make_double2(g[2 * c], g[2 * c + 1]);
            }
        }
        GPMPC_PST(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's G rows are in L2
        GPMPC_PST(4);
        GPMPC_LDS_BARRIER();
        GPMPC_PST(5);
        __builtin_amdgcn_s_dcache_inv();                           // the scalar cache may hold the previous step's rows
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MW == 2 && tiz < DS * NV) {                            // two waves per GP: their row sums in fixed order (read in phase 4, two barriers on)
            const int a = tiz / NV, m = tiz - a * NV;
            const double* r0 = &s_mred[(2 * a) * 4 * NV + m];
            const double* r1 = r0 + 4 * NV;
            s_ms[tiz] = ((r0[0] + r0[NV]) + (r0[2 * NV] + r0[3 * NV])) + ((r1[0] + r1[NV]) + (r1[2 * NV] + r1[3 * NV]));
        }
        GPMPC_PST(6);
        GPMPC_PSTW(16);
        // ---- 3: this wave's range of the N^2 sum, over UNITS of NG GPs (NG = 1: a unit is a GP) ------------------------------------
        // Round 5.  The loop is latency-bound per wave, not issue-bound (tools/ubench/latency_probe.hip: a scalar load of a G row that
        // is not in the scalar cache takes ~550 cycles, an LDS table read 60; four waves per SIMD hide 3 x 115 cycles per column): the
        // compiler's own schedule waited for scalar loads twice and for the table four times per two columns.  Columns are therefore
        // evaluated in BATCHES of KC with the stages pinned by sched_barriers -- all G rows of the batch (one scalar-load wait), all
        // exponents and table reads (one LDS wait), then the weights and moment sums -- and the weight loads of the next group of
        // columns are in flight meanwhile (both kinds of unit now: the shared-lambda units had no weight prefetch).  The lane sums of a
        // (unit, row block) segment are reduced and ADDED to the wave's LDS slot at the end of the segment instead of living in
        // 2 NG (1 + 2 D) registers across the whole range: that was what spilled (k_traj_persist<5,4,true,2>: 40 VGPR spills).
        {
            // this wave's range [r_lo, r_hi) and the unit it starts in (slot 0 of s_part; slot 1 = the next unit), from the table
            const int r_lo = __builtin_amdgcn_readfirstlane(s_rng[w]), r_hi = __builtin_amdgcn_readfirstlane(s_rng[w + 1]);
            int pos = r_lo, slot = 0, u = __builtin_amdgcn_readfirstlane(s_ga[w]);
            int rem = pos - u * per_gp, r = 0;                      // position within the unit; row block: the largest r with start(r) <= rem
            while (r + 1 < T && (r + 1) * Nc - 32 * (r + 1) * r <= rem) ++r;
            for (int e = ln; e < 2 * NG * 4 * NM; e += 64) s_part[w * (2 * NG * 4 * NM) + e] = 0.0;   // (a wave's LDS operations execute in order)
            const double* Gl = Gs;
            asm volatile("" : "+s"(Gl) :: "memory");               // rows written a moment ago: keep the scalar loads behind the barrier
            typedef const double __attribute__((address_space(4))) gpmpc_cdouble;
            while (pos < r_hi) {
                const int bstart = r * Nc - 32 * r * (r - 1);      // start(r) = r Nc - 32 r (r - 1)
                const int j0 = 64 * r + (rem - bstart);            // first column of the segment
                const int blen = Nc - 64 * r;                      // columns of the row block
                int n = bstart + blen - rem;                       // ... left in it
                if (n > r_hi - pos) n = r_hi - pos;
                const int i0 = 64 * r;
                __amdgpu_buffer_rsrc_t Mrs[NG];
#pragma unroll
                for (int q = 0; q < NG; ++q) {
                    const int a = u * NG + q < DS ? u * NG + q : DS - 1;       // (a partial last unit re-reads the last GP; not combined)
                    Mrs[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A.M + ((size_t)a * Np + j0) * Np + i0), 0, 0x7fffffff, 0x00020000);
                }
                double mga[MGc][NG], mgb[MGc][NG];
#pragma unroll
                for (int c = 0; c < MGc; ++c)
#pragma unroll
                    for (int q = 0; q < NG; ++q) mga[c][q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[q], l8, (c < n ? c : 0) * Np * 8, 0));
                const int at = SH ? 0 : u;                         // the transform of GP 0 is that of every GP when lambda is shared
                double hi2[D], qi;
                {
                    double q = 0.0;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double h = fma(-s_sc[at * D + k], s_X[k * Np + i0 + ln], s_cv[at * D + k]);
                        hi2[k] = (2.0 * GPMPC_EXP_NEG_INV_C) * h;
                        q = fma(h, h, q);
                    }
                    qi = GPMPC_EXP_NEG_INV_C * q;
                }
                double acc[NG][NA];
#pragma unroll
                for (int q = 0; q < NG; ++q)
#pragma unroll
                    for (int m = 0; m < NA; ++m) acc[q][m] = 0.0;
                const double* Ga = Gl + ((size_t)at * Np + j0) * GW;                 // the segment's first column pair (j0 is even)
                // KC columns: rows -> exponents + table reads -> weights x exp, moment sums
                auto batch = [&](auto kc_tag, int j, const double (*mw)[NG]) {
                    constexpr int KC = decltype(kc_tag)::value;     // (shadows the kernel's: the tail of a KC = 3 loop runs batches of two)
                    double g[KC][GW];
#if GPMPC_PERSIST_TOUCH   /* (pair layout only; measured SLOWER: profiles/r05/persist_stamps_touch.txt) */
                    // Scalar-cache warm-up: one dword of every 64-byte line of the NEXT batch's rows is requested together with this batch's
                    // rows (same wait), so that the next batch's loads hit the scalar cache (~90 cycles) instead of the L2 (~550: the
                    // latency four waves per SIMD do not hide, tools/ubench/latency_probe.hip).  Within the segment only (the last batch
                    // touches itself): nothing is read beyond the trajectory's rows.
                    unsigned touch[(KC * GW * 8 + 63) / 64];
                    {
                        const double* gn = Ga + (size_t)((j + KC < n ? j + KC : j) >> 1) * (2 * GW) + ((j + KC) & 1) * 2;
                        asm volatile("" : "+s"(gn));
#pragma unroll
                        for (int q = 0; q < (KC * GW * 8 + 63) / 64; ++q) touch[q] = ((const unsigned __attribute__((address_space(4)))*)gn)[16 * q];
                    }
#endif
                    // (the base goes through an opaque asm per batch: left to see that consecutive batches are adjacent in memory the
                    // compiler merges their loads into 64-byte ones -- 80 SGPRs of rows, the loop's other scalars spilled to VGPR lanes
                    // and read back with 18 v_readlane per iteration)
                    const double* gb = GPAIR ? Ga + (size_t)(j >> 1) * (2 * GW) + (j & 1) * 2 : Ga + (size_t)j * GW;       // (j is even with KC = 2)
                    asm volatile("" : "+s"(gb));
#pragma unroll
                    for (int k = 0; k < D + 1 + NS2; ++k)             // pair layout: chunk k / 2 of the pair is [column j | column j + 1], 16 bytes each
#pragma unroll
                        for (int c = 0; c < KC; ++c)
                            g[c][k] = GPAIR ? ((const gpmpc_cdouble*)(gb + (size_t)(k >> 1) * 4 + c * 2))[k & 1] : ((const gpmpc_cdouble*)(gb + (size_t)c * GW))[k];
                    __builtin_amdgcn_sched_barrier(0);
#if GPMPC_PERSIST_TOUCH
#pragma unroll
                    for (int q = 0; q < (KC * GW * 8 + 63) / 64; ++q) asm volatile("" :: "s"(touch[q]));
#endif
                    double fr[KC], pq[KC], Tv[KC];
                    int ni[KC];
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        double sx = qi + g[c][D];
#pragma unroll
                        for (int k = 0; k < D; ++k) sx = fma(hi2[k], g[c][k], sx);
                        const double ax = __builtin_fabs(sx);      // gpmpc_exp_neg_scaled (fast_exp.h), split around the table read
                        ni[c] = (int)(-ax);
                        fr[c] = __builtin_amdgcn_fract(ax);
                        Tv[c] = s_tab[ni[c] & (GPMPC_EXP_N - 1)];
                        if (NG == 1) pq[c] = fma(fr[c], fma(fr[c], GPMPC_EXP_A3, GPMPC_EXP_A2), GPMPC_EXP_A1);   // in the shadow of the table read
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        if (NG > 1) pq[c] = fma(fr[c], fma(fr[c], GPMPC_EXP_A3, GPMPC_EXP_A2), GPMPC_EXP_A1);    // (two GPs' sums: registers are short)
                        const double e = ldexp(fma(Tv[c] * fr[c], pq[c], Tv[c]), ni[c] >> GPMPC_EXP_BITS);
#pragma unroll
                        for (int q = 0; q < NG; ++q) {
                            const double P = mw[c][q] * e;
                            acc[q][0] += P;
                            if (GRAD) {
#pragma unroll
                                for (int k = 0; k < D; ++k) acc[q][GRAD ? 1 + k : 0] = fma(P, g[c][k], acc[q][GRAD ? 1 + k : 0]);
#pragma unroll
                                for (int k = 0; k < NS2; ++k) acc[q][GRAD ? 1 + D + k : 0] = fma(P, g[c][D + 1 + k], acc[q][GRAD ? 1 + D + k : 0]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
#if GPMPC_PERSIST_ROTPRIO == 2
                unsigned long long clk = __builtin_amdgcn_s_memtime();
#endif
                int jc = 0;
                for (; jc + 2 * MGc <= n; jc += 2 * MGc) {         // n is a multiple of GR (= 2 MGc, or 2 with a tail below)
#if GPMPC_PERSIST_ROTPRIO
                    // The four waves a SIMD holds of this workgroup are arbitrated by age: left alone the oldest runs ahead and the
                    // youngest finishes 40 % later, the SIMD half empty at the end (stamps: wave 2 109 k cycles, wave 15 165 k for
                    // equal ranges).  Rotating the issue priority lets them advance together.  Round 5: the rotation follows the
                    // shader CLOCK, not the wave's own column count -- positions within segments differ from wave to wave, priorities
                    // then coincide and ties go to the older wave again (waves 12-15 finished 15-20 % after the others); with a common
                    // clock the four waves of a SIMD hold four distinct priorities at every instant.  The counter read is issued one
                    // iteration ahead (a scalar memory instruction: consumed after waits that happen anyway).
#if GPMPC_PERSIST_ROTPRIO == 2
                    const unsigned tick = (unsigned)(clk >> GPMPC_PERSIST_CLK_SHIFT);
                    clk = __builtin_amdgcn_s_memtime();
                    switch ((tick + (w >> 2)) & 3) {
#else
                    switch (((jc >> GPMPC_PERSIST_PRIO_SHIFT) + (w >> 2)) & 3) {
#endif
                        case 0: __builtin_amdgcn_s_setprio(0); break;
                        case 1: __builtin_amdgcn_s_setprio(1); break;
                        case 2: __builtin_amdgcn_s_setprio(2); break;
                        default: __builtin_amdgcn_s_setprio(3); break;
                    }
#endif
#pragma unroll
                    for (int c = 0; c < MGc; ++c)
#pragma unroll
                        for (int q = 0; q < NG; ++q)
                            mgb[c][q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[q], l8, (jc + MGc + c) * Np * 8, 0));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < MGc; c += KC) batch(std::integral_constant<int, KC>{}, jc + c, &mga[c]);
                    {   // unconditional (the last iteration re-requests its own first group, unused): under a branch the compiler's
                        // wait counts merge both paths and every wait below becomes "all loads done"
                        const int jn = jc + 4 * MGc <= n ? jc + 2 * MGc : jc;
#pragma unroll
                        for (int c = 0; c < MGc; ++c)
#pragma unroll
                            for (int q = 0; q < NG; ++q)
                                mga[c][q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[q], l8, (jn + c) * Np * 8, 0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < MGc; c += KC) batch(std::integral_constant<int, KC>{}, jc + MGc + c, &mgb[c]);
                }
                if constexpr (KC == 3) {                           // 2 or 4 columns left (their weights are not prefetched: ~2 such tails per wave and step)
                    for (; jc < n; jc += 2) {
                        double mt[2][NG];
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int q = 0; q < NG; ++q) mt[c][q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[q], l8, (jc + c) * Np * 8, 0));
                        __builtin_amdgcn_sched_barrier(0);
                        batch(std::integral_constant<int, 2>{}, jc, mt);
                    }
                }
                // per-ln combination into the m-moments of this (unit, row block) segment (pair_kernel_sb.h), reduced over rows of 16
                // lanes and added to the wave's slot of the unit: segments in program order, rows as (0 + 1) + (2 + 3) in the combine
#pragma unroll
                for (int q = 0; q < NG; ++q) {
                    const double rs = acc[q][0];
                    double* sp = &s_part[(((w * 2 + slot) * NG + q) * 4 + (ln >> 4)) * NM];
                    // one moment at a time (nothing held): row sum over 16 lanes, then ds_add_f64 by the row's first lane -- same lane,
                    // program order: the slot sums its segments in a fixed order
                    auto add = [&](int m, double v) {
                        const double sr = wave_row_sum(v);
                        if ((ln & 15) == 0) __hip_atomic_fetch_add(&sp[m], sr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    };
                    add(0, rs);
                    if (GRAD) {
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const double h = (0.5 / GPMPC_EXP_NEG_INV_C) * hi2[k], v = acc[q][GRAD ? 1 + k : 0];
                            add(GRAD ? 1 + k : 0, fma(h, rs, v));
                            if (k < NS2) add(GRAD ? 1 + D + k : 0, fma(h * h, rs, fma(2.0 * h, v, acc[q][GRAD ? 1 + D + (k < NS2 ? k : 0) : 0])));
                        }
                    }
                }
                pos += n;
                rem += n;
                if (rem >= bstart + blen) {                        // the row block is done: next block, or the next unit's first
                    if (++r == T) { r = 0; rem = 0; ++u; slot = 1; }
                }
            }
            GPMPC_PSTW(32);
#if GPMPC_PERSIST_ROTPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            GPMPC_PSTW(48);
        }
        GPMPC_PST(7);
        GPMPC_LDS_BARRIER();
        GPMPC_PST(8);
        // ---- 4: combine, outputs and Jacobian rows of step t, input moments of step t + 1 ------------------------------------
        if (tiz >= 64 && tiz < 64 + DS) {                          // c_m, c of the step (wave 1, beside the combine of wave 0)
            const int a = tiz - 64;
            double detm = 1.0, detv = 1.0;
            for (int l = 0; l < D; ++l) { detm *= s_r1[a * D + l]; detv *= s_r2[a * D + l]; }
            const double sf = A.sf[a], sf2 = sf * sf;
            s_sf2[a] = sf2; s_cm[a] = sf2 / sqrt(detm); s_c[a] = 1.0 / sqrt(detv);
        }
        // every (GP, moment) sum by a ROW of 16 lanes, one lane per wave of the workgroup (round 5: one thread per sum walked the 16
        // waves in a chain of dependent LDS reads, ~3 k cycles of a 9.5 k phase); fixed order: rows as (0 + 1) + (2 + 3), then the
        // 16-lane row sum's tree over the waves
        int tiz4 = tiz;                                            // (opaque: index arithmetic of this phase is not computed ahead of the column loop and held across it)
        asm volatile("" : "+v"(tiz4));
        for (int o = tiz4 >> 4; o < DS * NM; o += nthr >> 4) {
            const int ww = tiz4 & 15, a = o / NM, m = o - a * NM;
            double v = 0.0;
            if (ww < NW) {
                const int lo = s_rng[ww], hi = s_rng[ww + 1];
                const int a0 = s_ga[ww];                           // the range ends in unit a0 or a0 + 1: hi <= (a0 + 2) per_gp
                const int un = a / NG, qn = a - un * NG;           // unit of GP a and its place in it
                if (lo < hi && (un == a0 || (un == a0 + 1 && hi > (a0 + 1) * per_gp))) {
                    const double* r4 = &s_part[(((ww * 2 + (un - a0)) * NG + qn) * 4) * NM + m];
                    v = (r4[0] + r4[NM]) + (r4[2 * NM] + r4[3 * NM]);
                }
            }
            v = wave_row_sum(v);
            if (ww == 0) s_z[o] = v;
        }
        GPMPC_LDS_BARRIER();
        if (tiz4 < DS * D) {
            const int a = tiz4 / D, k = tiz4 - a * D, nc = 2 * DS + DA;
            const double c = s_c[a], cm = s_cm[a], sf2 = s_sf2[a];
            const double mu = cm * s_ms[a * NV];
            const double Tt = c * s_z[a * NM];
            const double var = sf2 - Tt - mu * mu;                 // no clamp (src/tools/uncertainty_prop.py:399)
            if (k == 0) {
                A.means[((size_t)b * (A.H + 1) + t) * DS + a] = mu;
                A.vars[((size_t)b * (A.H + 1) + t) * DS + a] = var;
            }
            if (GRAD) {
                const double Bq = s_B[tiz4], Ak = s_Ak[tiz4], sc = s_sc[tiz4];
                const double dmu_du = -Bq * cm * s_ms[a * NV + 1 + k];
                const double dmu_ds = -0.5 * mu * Bq + 0.5 * Bq * Bq * cm * s_ms[a * NV + 1 + D + k];
                const double dT_du = -4.0 * sc * c * s_z[a * NM + (GRAD ? 1 + k : 0)];
                const double dv_du = -dT_du - 2.0 * mu * dmu_du;
                double* jm = A.jac + (((size_t)b * A.H + (t - 1)) * 2 * DS + a) * nc;          // row of mu_a
                double* jv = A.jac + (((size_t)b * A.H + (t - 1)) * 2 * DS + DS + a) * nc;     // row of var_a
                if (k < DS) {
                    const double dT_ds = Ak * (c * s_z[a * NM + (GRAD ? 1 + D + (k < NS2 ? k : 0) : 0)] - 0.5 * Tt);
                    const double dv_ds = -dT_ds - 2.0 * mu * dmu_ds;
                    jm[k] = dmu_du; jm[DS + k] = dmu_ds;
                    jv[k] = dv_du;  jv[DS + k] = dv_ds;
                } else {                                           // action input: its variance is a constant
                    jm[2 * DS + (k - DS)] = dmu_du;
                    jv[2 * DS + (k - DS)] = dv_du;
                }
            }
        }
        GPMPC_PST(9);
        GPMPC_LDS_BARRIER();                                           // every reader of s_uin / s_sin of step t is done
        GPMPC_PST(10);
        if (tiz < DS) {
            const double mu = s_cm[tiz] * s_ms[tiz * NV];
            s_uin[tiz] = mu;
            s_sin[tiz] = s_sf2[tiz] - s_c[tiz] * s_z[tiz * NM] - mu * mu;
        }
    }
}

template <int D, int NS2, bool GRAD, int NG = 1>
static int launch_persist_one(const PersistArgs& a, int waves, hipStream_t s) {
    // dynamic LDS: X, the actions, and -- instances whose column loop runs batches of three (row-major rows) -- the row writers' staging blocks
    constexpr int DSh = NS2, GWh = (D + 1 + NS2 + 1) & ~1;
    constexpr bool kc3 = !(D >= 7 || (NG > 1 && D >= 6) || NG >= 4 || (NG == 3 && D >= 5)) && NG == 1 && D <= 5 && GPMPC_PERSIST_KC3;
    const int mw = (waves >= 16 && 2 * DSh <= waves / 2) ? 2 : 1;
    const size_t lds = sizeof(double) * ((size_t)D * a.Np + (size_t)((a.H * (D - NS2) + 1) & ~1) + (kc3 ? (size_t)(waves - mw * DSh) * 64 * GWh : 0));
    if (a.Np > GPMPC_PERSIST_MAXNP || a.Np % 64 != 0 || (waves != 8 && waves != 16) || a.H * (D - NS2) > 2 * 512) return GPMPC_E_ARG;
    // static + dynamic LDS beyond the default 64 KB of a launch needs an opt-in (160 KB per CU on gfx950).  The attribute belongs to the
    // function object of the CURRENT device: tracked per device, and the bound uses the instance's own static LDS (NG = 2 / GRAD
    // instances carry ~36 KB of it) -- a shape that cannot fit is refused here, not at the launch.
    {
        static int static_lds[GPMPC_PERSIST_MAXDEV];           // 0: not asked yet on that device; else static LDS bytes + 1
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GPMPC_PERSIST_MAXDEV) return GPMPC_E_DEVICE;
        const void* fn = reinterpret_cast<const void*>(&k_traj_persist<D, NS2, GRAD, NG>);
        if (!static_lds[dev]) {                                // (a benign race: both writers store the same value)
            hipFuncAttributes fa;
            if (hipError_t ea = hipFuncGetAttributes(&fa, fn); ea != hipSuccess) { gpmpc_set_error("trajectory-persistent kernel: attributes", ea); return GPMPC_E_LAUNCH; }
            const int dyn_max = 160 * 1024 - (int)fa.sharedSizeBytes;
            if (hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_max);
                ea != hipSuccess) { gpmpc_set_error("trajectory-persistent kernel: LDS attribute", ea); return GPMPC_E_LAUNCH; }
            static_lds[dev] = (int)fa.sharedSizeBytes + 1;
        }
        if (lds + (size_t)(static_lds[dev] - 1) > 160 * 1024) return GPMPC_E_ARG;
    }
    hipLaunchKernelGGL((k_traj_persist<D, NS2, GRAD, NG>), dim3(a.B), dim3(64 * waves), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("trajectory-persistent rollout kernel launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

// ns2 = state_dim (D - ns2 in {1, 2} action dimensions); waves per workgroup 16 | 8; ng = 1, or 2: units of two GPs with one lambda
// (instantiated up to D = 6: beyond, two GPs' accumulators and lane sums do not fit the 128 registers of four waves per SIMD)
template <int D>
int gpmpc_launch_persist_D(bool grad, int ns2, int waves, int ng, const PersistArgs& a, hipStream_t s) {
    if (ng < 1 || ng > 4) return GPMPC_E_ARG;
    // units of THREE / FOUR GPs (round 5: all GPs of a ds = 3 / ds = 4 pack with one lambda in ONE unit -- exponent and exp once per pair for
    // all of them, 14.25 instead of 17.5 fp64 instructions per pair and GP at D = 5, ds = 4): the instances whose accumulators fit the 128
    // registers of four waves per SIMD without a spill (tools/spill_guard.py); 16-wave workgroups only (69 KB of static LDS at NG = 4)
    if constexpr (D == 5) {
        if (ng == 4 && ns2 == 4) return grad ? launch_persist_one<5, 4, true, 4>(a, waves, s) : launch_persist_one<5, 4, false, 4>(a, waves, s);
        if (ng == 3 && ns2 == 3) return grad ? launch_persist_one<5, 3, true, 3>(a, waves, s) : launch_persist_one<5, 3, false, 3>(a, waves, s);
    }
    if constexpr (D == 4) {
        if (ng == 3 && ns2 == 3) return grad ? launch_persist_one<4, 3, true, 3>(a, waves, s) : launch_persist_one<4, 3, false, 3>(a, waves, s);
    }
    if (ng > 2) return GPMPC_E_ARG;
    if constexpr (D >= 3 && D <= 6) {
        if (ng == 2 && ns2 >= 2) {
            if (ns2 == D - 1) return grad ? launch_persist_one<D, D - 1, true, 2>(a, waves, s) : launch_persist_one<D, D - 1, false, 2>(a, waves, s);
            if constexpr (D >= 4) { if (ns2 == D - 2) return grad ? launch_persist_one<D, D - 2, true, 2>(a, waves, s) : launch_persist_one<D, D - 2, false, 2>(a, waves, s); }
            return GPMPC_E_ARG;
        }
    }
    if (ng != 1) return GPMPC_E_ARG;
    if constexpr (D >= 2) {
        if (ns2 == D - 1) return grad ? launch_persist_one<D, (D >= 2 ? D - 1 : 1), true>(a, waves, s) : launch_persist_one<D, (D >= 2 ? D - 1 : 1), false>(a, waves, s);
    }
    if constexpr (D >= 3) {
        if (ns2 == D - 2) return grad ? launch_persist_one<D, (D >= 3 ? D - 2 : 1), true>(a, waves, s) : launch_persist_one<D, (D >= 3 ? D - 2 : 1), false>(a, waves, s);
    }
    return GPMPC_E_ARG;
}
