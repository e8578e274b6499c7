// Pair kernel for the rollout hot path (variance units, diagonal S, forward + gradient), "scalar broadcast" form.
//
// Same sum as pair_kernel.h (reference: src/tools/uncertainty_prop.py:372-399),
//     P_ij = M_ij exp(-|h_i + h_j|^2),   Z0 = sum P,  Z1_k = sum P m_k,  Z2_k = sum P m_k^2,   m = h_i + h_j,
// reorganised so that everything that depends on the column j is WAVE-UNIFORM and everything that depends on the row
// i is per-lane:
//     |h_i + h_j|^2 = q_i + q_j + sum_k (2 h_ik) h_jk                       (q = |h|^2)
//     Z0   = sum_i r_i                          r_i  = sum_j P_ij
//     Z1_k = sum_i ( h_ik r_i + v_ik )          v_ik = sum_j P_ij h_jk
//     Z2_k = sum_i ( h_ik^2 r_i + 2 h_ik v_ik + w_ik )      w_ik = sum_j P_ij h_jk^2
// The column rows G[j] = [h_j1..h_jD | q_j N/ln2 | h_j1^2..h_jNS2^2] are written once per (trajectory, GP, step) by the head
// kernel (step.hip) and read here through wave-uniform addresses, i.e. as SCALAR loads into SGPRs that the fp64 VALU
// instructions take as their one scalar operand: no LDS staging, no barrier in the loop, no per-lane h_j registers.
// Per pair the VALU issues 1 + D (exponent) + 7 (table exp, fast_exp.h) + 1 (M_ij e) + 1 + D + NS2 (r, v, w) fp64-rate
// instructions + 3 integer ones: 24 + 3 for D = 5, NS2 = 4 (30 + 3 for D = 7, NS2 = 6) against 35 for the staged form, whose
// adds/squares of m are gone.  At the measured 4.4 / 2.5 cycles per wave64 instruction (tools/ubench/valu_op_cost.hip)
// that is 115 (141) issue cycles per pair, and the kernel runs at ~95 % of it.  The exponent is the
// expanded form the reference itself uses (:380-389, u A u + X A X^T - ...), here centred on u (h = sc (u - x)), so
// its absolute error is ~1e-16 (q_i + q_j) instead of ~1e-16 |m|^2; still fp64 throughout.
// fp64 MFMA cannot help here: on MI355X it shares the fp64 VALU's issue capacity (profiles/r01/ubench_mfma_f64_overlap.txt).
#pragma once
#include "gpmpc_internal.h"
#include "fast_exp.h"
#include <cstdlib>

// Columns per iteration of the column loop.  ONE is fastest: the plain load / wait / evaluate loop (43 instructions,
// 45 VGPRs at D = 5) beat the 2- and 4-column unrollings by 0.9 % / 1.7 % on C3 and 1.5 % / 6 % on C4 -- with 8
// waves/SIMD the other waves cover the exposed latencies, and the short body is kinder to instruction fetch.
// NOTE: the loop nest below keeps its generic form (RI row chunks, CU columns per iteration, both 1) on purpose.  A
// hand-simplified body with the same arithmetic made the compiler rotate the loop (M_ij fetched one iteration ahead, the
// two exp-table reads serialised) and ran 2.5 % slower on C3: check the ISA and A/B on one box before touching it.
#ifndef GPMPC_SB_CU
#define GPMPC_SB_CU(gw) 1
#endif

#ifndef GPMPC_SB_TB2_WAVES
// Waves per SIMD the two-trajectory shape is compiled for.  5 (96 registers; the loop needs 85): with the 7-slot exp the
// 80-register cap of 6 waves/SIMD made the compiler re-materialise a constant and an address in every iteration and
// spill; measured on one MI355X (C3): 6 waves 5.31 k rollouts/s, one trajectory per wave at 8 waves 6.07 k, 5 waves 6.24 k
// (the 9-slot exp it replaces: 5.98-6.02 k at 6 waves) -- profiles/r02/README.md.  The interleaving of the two trajectories'
// dependency chains is what the shape is for: with a sched_barrier between them the loop fits 80 registers (6 waves)
// but runs 8 % slower, 20 % slower at 5 waves.
#define GPMPC_SB_TB2_WAVES 5
#endif

// Diagnostic build (make EXTRA=-DGPMPC_SB_STAMPS, tools/sb_stamps.py): per-workgroup timeline of the D = 5 instances -- dispatch
// time, prologue, column loop, reduction -- to see where a mid-size launch spends its time.  Not compiled otherwise.
#if defined(GPMPC_SB_STAMPS) && GPMPC_PAIR_D == 5
#define GPMPC_SB_NSTAMP 8
static __device__ unsigned long long g_sb_stamps[8192 * GPMPC_SB_NSTAMP];
#define GPMPC_SBST(slot, val) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_sb_stamps[blockIdx.x * GPMPC_SB_NSTAMP + (slot)] = (val); } while (0)
#else
#define GPMPC_SBST(slot, val) do { } while (0)
#endif

template <int D, int NS2>
struct PairSbTraits {
    static constexpr int GW = (D + 1 + NS2 + 1) & ~1;     // doubles per G row (even: rows stay 16-byte aligned)
};

// GRAD = false: objective only (Z0), NM = 1.  (A two-rows-per-lane shape was measured 15 % slower: occupancy wins.)
// FIRST: horizon step 1, whose state inputs (x0, Sigma_0) are constants: only the derivatives w.r.t. the action dimensions
// (k >= NS2) are needed, so the w and the state-dimension v accumulations are dropped (8 of 30 VALU instructions at D = 5).
// CUV: columns per iteration of the column loop.  1 for full launches (see GPMPC_SB_CU above).  4 for the 256x64 tiling of
// MID-SIZE batches: there the chip is not full for most of a launch (the last generation of workgroups runs at 1-2 waves per
// SIMD), and a lone wave of the one-column loop is bound by the latency of its M_ij load -- measured per workgroup with
// tools/sb_stamps.py (profiles/r03/sb_stamps_mid.txt): 64 columns take 23 us at 8 waves per SIMD AND on an empty chip (~360 ns
// per column either way: issue-bound when full, one L2 round trip per column when alone).  Four columns in flight give a wave
// four loads per round trip and four independent dependency chains.
template <int D, int TB, int NS2, bool GRAD, bool FIRST = false, int CUV = 1>
__global__ __launch_bounds__(256, (TB == 2 && D <= 5) ? GPMPC_SB_TB2_WAVES : 1) void gpmpc_pair_kernel_sb(PairSbArgs A) {
    constexpr int GW = PairSbTraits<D, NS2>::GW;
    constexpr int NM = GRAD ? 1 + 2 * D : 1, NA = GRAD ? 1 + D + NS2 : 1;
    __shared__ double s_red[16 * TB * NM];        // [wave][row of 16 lanes][trajectory][moment]
    __shared__ double s_tab[GPMPC_EXP_N];
    GPMPC_SBST(0, __builtin_amdgcn_s_memrealtime());
    GPMPC_SBST(1, __builtin_amdgcn_s_memtime());
    gpmpc_exp_table_to_lds(s_tab);

    // XCD-aware decode (pair_kernel.h): position p of the work list runs on XCD p % 8.  Within an XCD the dispatch
    // order is [group of R items][trajectory][item of the group]: the ~256 workgroups resident on an XCD are R row
    // tiles x 256/R trajectories, and the R tiles of a group share their column block (build_worklist sorts the XCD
    // lists that way), so each trajectory's G rows are fetched into the XCD's L2 once per R row tiles.
    int bg, wi;
    {
        const int groups = (A.B + TB - 1) / TB, items = A.nwork, R = A.rgroup;
        const int L = blockIdx.x, full = (items / (8 * R)) * (8 * R);
        if (L < full * groups) {
            const int q = L >> 3, ir = q % R, t = q / R;
            bg = t % groups;
            wi = ((t / groups) * R + ir) * 8 + (L & 7);
        } else { const int Lt = L - full * groups; wi = full + Lt / groups; bg = Lt % groups; }
    }
    const int unit = A.work[wi * 4 + 0], i0 = A.work[wi * 4 + 1], j0 = A.work[wi * 4 + 2], j1w = A.work[wi * 4 + 3];
    // columns >= N carry zero weight: the tile's column loop ends at N rounded up to 8 (round 5; read from the pack in device memory, so a
    // captured launch stays valid while the training set grows within its padded size) -- N = 200: 56 of the 256 columns of a row
    const int ncw = *(const int __attribute__((address_space(4)))*)A.ncol;
    const int j1 = j1w < ncw ? j1w : ncw;
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index is uniform across the wave: tell the compiler, so that everything indexed by the column loop
    // below is provably wave-uniform and the G rows are fetched by scalar loads
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Np = A.Np;
    constexpr int RI = 1;
    const int iw0 = i0 + w * 64;                          // first row of this wave (wave-uniform)

    const double* __restrict__ Gt[TB];
    double hi2[TB][RI][D], qi[TB][RI];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
        int b = bg * TB + tb;
        b = b < A.B ? b : A.B - 1;
        const double* __restrict__ prm = A.pp + ((size_t)b * A.ds + unit) * A.pps;
        Gt[tb] = A.G + ((size_t)b * A.ds + unit) * Np * GW;
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const int i = iw0 + 64 * r + lane;
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double x = (iw0 + 64 * r < Np) ? A.XT[(size_t)k * Np + i] : 0.0;
                const double h = fma(-prm[D + k], x, prm[k]);
                hi2[tb][r][k] = (2.0 * GPMPC_EXP_NEG_INV_C) * h;     // exponent carried as s * N/ln2 (fast_exp.h)
                q = fma(h, h, q);
            }
            qi[tb][r] = GPMPC_EXP_NEG_INV_C * q;
        }
    }

    double acc[TB][RI][NA];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int r = 0; r < RI; ++r)
#pragma unroll
            for (int m = 0; m < NA; ++m) acc[tb][r][m] = 0.0;

    const double* __restrict__ Ma = A.M + (size_t)unit * Np * Np;
    __syncthreads();                                      // exp table ready
    GPMPC_SBST(2, __builtin_amdgcn_s_memtime());

    if (iw0 < Np) {
        // columns left of every row of this wave carry zero weight (upper-triangular M): start at the wave's diagonal chunk
        const int jstart = j0 > (iw0 & ~63) ? j0 : (iw0 & ~63);
        __amdgpu_buffer_rsrc_t Mrs[RI];
#pragma unroll
        for (int r = 0; r < RI; ++r)
            Mrs[r] = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Ma + (size_t)jstart * Np + iw0 + 64 * r), 0, 0x7fffffff, 0x00020000);
        const int lane8 = lane * 8;
        constexpr int CU = CUV > 1 ? CUV : GPMPC_SB_CU(GW);      // columns per loop iteration
        for (int jc = jstart; jc < j1; jc += CU) {
            double mij[RI][CU];
#pragma unroll
            for (int r = 0; r < RI; ++r)
#pragma unroll
                for (int q = 0; q < CU; ++q) {
                    // buffer load: the column offset rides in the scalar offset, the lane offset is loop invariant,
                    // so the M_ij stream costs no VALU address arithmetic
                    const auto raw = __builtin_amdgcn_raw_buffer_load_b64(Mrs[r], lane8, (jc - jstart + q) * Np * 8, 0);
                    mij[r][q] = __builtin_bit_cast(double, raw);
                }
            // keep the M_ij load at the top of the iteration: left to itself the scheduler may sink it next to its use
            // (it did with the 7-slot exp: C3 2.18 -> 2.46 ms per launch, the load latency exposed in every iteration)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < CU; ++q) {
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) {
                    // wave-uniform address -> SGPRs.  The rows were written by the head kernel of this step and are constant for
                    // the whole launch: read through the CONSTANT address space, so that the scalar loads do not depend on the
                    // compiler proving that no earlier store of this kernel can alias them (any global store ahead of the loop
                    // -- the stamps of the diagnostic build, a persistent-loop variant -- turns them into per-lane vector loads
                    // of one address otherwise: 2.5x slower).
                    typedef const double __attribute__((address_space(4))) gpmpc_cdouble;
                    const gpmpc_cdouble* __restrict__ g = (const gpmpc_cdouble*)(Gt[tb] + (size_t)(jc + q) * GW);
#pragma unroll
                    for (int r = 0; r < RI; ++r) {
                        if (r > 0 && jc + CU - 1 < iw0 + 64 * r) continue;                  // row block r is still below the diagonal
                        double s = qi[tb][r] + g[D];
#pragma unroll
                        for (int k = 0; k < D; ++k) s = fma(hi2[tb][r][k], g[k], s);
                        const double P = mij[r][q] * gpmpc_exp_neg_scaled(s, s_tab);
                        acc[tb][r][0] += P;
                        if (GRAD) {
#pragma unroll
                            for (int k = 0; k < D; ++k) if (!FIRST || k >= NS2) acc[tb][r][1 + k] = fma(P, g[k], acc[tb][r][1 + k]);
#pragma unroll
                            for (int k = 0; k < NS2; ++k) if (!FIRST) acc[tb][r][1 + D + k] = fma(P, g[D + 1 + k], acc[tb][r][1 + D + k]);
                        }
                    }
                }
            }
        }
    }

    GPMPC_SBST(3, __builtin_amdgcn_s_memtime());
    // per-lane combination into the m-moments, then the fixed-order workgroup reduction
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
        double z[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) z[m] = 0.0;
#pragma unroll
        for (int r = 0; r < RI; ++r) {
            const double rs = acc[tb][r][0];
            z[0] += rs;
            if (GRAD) {
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double h = (0.5 / GPMPC_EXP_NEG_INV_C) * hi2[tb][r][k], v = acc[tb][r][GRAD ? 1 + k : 0];
                    if (!FIRST || k >= NS2) z[GRAD ? 1 + k : 0] += fma(h, rs, v);
                    if (k < NS2 && !FIRST) z[GRAD ? 1 + D + k : 0] += fma(h * h, rs, fma(2.0 * h, v, acc[tb][r][GRAD ? 1 + D + (k < NS2 ? k : 0) : 0]));
                }
            }
        }
        // one moment at a time (reducing all of them first keeps NM more doubles live: 60 -> 70 VGPRs at D = 7, a wave per SIMD lost)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double s = wave_row_sum(z[m]);                          // every lane: the sum of its row of 16 lanes
            if ((lane & 15) == 0) s_red[((w * 4 + (lane >> 4)) * TB + tb) * NM + m] = s;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < TB * NM; idx += blockDim.x) {
        const int tb = idx / NM, m = idx - tb * NM;
        const int b = bg * TB + tb;
        if (b < A.B) {
            double s = 0.0;                       // fixed order: waves, each as (row 0 + row 1) + (row 2 + row 3)
            for (int ww = 0; ww < (int)(blockDim.x >> 6); ++ww) {
                const double* r4 = &s_red[(ww * 4 * TB + tb) * NM + m];
                s += (r4[0] + r4[TB * NM]) + (r4[2 * TB * NM] + r4[3 * TB * NM]);
            }
            A.part[((size_t)b * A.nwork + wi) * A.nm + m] = s;
        }
    }
    GPMPC_SBST(4, __builtin_amdgcn_s_memtime());
    GPMPC_SBST(5, __builtin_amdgcn_s_memrealtime());
}

// `rows` = rows per tile of the work list (64 or 256) = threads per workgroup.
// Shapes measured on C3 (round 2, 7-slot exp): two trajectories per wave compiled for 5 waves/SIMD 2.00 ms per launch, one
// per wave at 8 waves/SIMD 2.05, two per wave squeezed into the 80 registers of 6 waves/SIMD 2.36 (profiles/r02); D = 7
// (C4) runs one trajectory per wave: 63 VGPRs, 8 waves/SIMD -- a 64th-register crossing costs 3 % there.
template <int D, int TB, int NS2, bool GRAD, bool FIRST = false>
static int launch_pair_sb_one(int rows, const PairSbArgs& a, hipStream_t s) {
    dim3 grid(((a.B + TB - 1) / TB) * a.nwork), block(rows);
    if constexpr (TB == 1) {
        if (a.colunroll == 4) hipLaunchKernelGGL((gpmpc_pair_kernel_sb<D, TB, NS2, GRAD, FIRST, 4>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gpmpc_pair_kernel_sb<D, TB, NS2, GRAD, FIRST>), grid, block, 0, s, a);
    } else
    hipLaunchKernelGGL((gpmpc_pair_kernel_sb<D, TB, NS2, GRAD, FIRST>), grid, block, 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("pair kernel (scalar broadcast) launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

// ns2 = state_dim: D - ns2 in {0, 1, 2} action dimensions are supported by this kernel.
template <int D>
int gpmpc_launch_pair_sb_D(bool grad, int tb, int ns2, int waves, const PairSbArgs& a, hipStream_t s) {
    const int rows = 64 * waves;
    if (a.nm != (grad ? 1 + 2 * D : 1)) return GPMPC_E_ARG;
#define GPMPC_SB_FIRST(TBV)                                                                                      \
    if (a.first_step && grad && tb == TBV) {                                                                     \
        if (ns2 == D) return launch_pair_sb_one<D, TBV, D, true, true>(rows, a, s);                              \
        if (D >= 2 && ns2 == D - 1) return launch_pair_sb_one<D, TBV, (D >= 2 ? D - 1 : D), true, true>(rows, a, s);  \
        if (D >= 3 && ns2 == D - 2) return launch_pair_sb_one<D, TBV, (D >= 3 ? D - 2 : D), true, true>(rows, a, s);  \
        return GPMPC_E_ARG;                                                                                      \
    }
    GPMPC_SB_FIRST(1)
    GPMPC_SB_FIRST(2)
#undef GPMPC_SB_FIRST
#define GPMPC_SB_CASE(TBV, GR)                                                                             \
    if (tb == TBV && grad == GR) {                                                                         \
        if (ns2 == D) return launch_pair_sb_one<D, TBV, D, GR>(rows, a, s);                                \
        if (D >= 2 && ns2 == D - 1) return launch_pair_sb_one<D, TBV, (D >= 2 ? D - 1 : D), GR>(rows, a, s);    \
        if (D >= 3 && ns2 == D - 2) return launch_pair_sb_one<D, TBV, (D >= 3 ? D - 2 : D), GR>(rows, a, s);    \
        return GPMPC_E_ARG;                                                                                \
    }
    GPMPC_SB_CASE(1, true)
    GPMPC_SB_CASE(2, true)
    GPMPC_SB_CASE(1, false)
    GPMPC_SB_CASE(2, false)
#undef GPMPC_SB_CASE
    return GPMPC_E_ARG;
}
