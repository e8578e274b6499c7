// The O(N^2) pair kernel, general ("staged") form.  The rollout's hot path uses the scalar-broadcast forms
// (pair_kernel_sb.h, pair_kernel_sbf.h) once the grid fills the chip; this kernel serves small batches (one-wave /
// column-split 64x64 tiles), the objective-only full-S path and is the A/B reference of the others (GPMPC_PAIR_SB=0).
//
// For one (trajectory b, unit) it evaluates
//     Z0 = sum_{i,j} M_ij exp(-|p_i + q_j|^2),      p_i = cvec_r - T_r x_i (rows),  q_j = cvec_c - T_c x_j (columns)
// and, for the input-gradients, the moments  Z1_k = sum P m_k,  Z2_kl = sum P m_k m_l  with  m = p_i + q_j,
// P = M_ij exp(-|m|^2).  Two kinds of unit share the code:
//   * variance unit a (unit < ntri): the trace term of variance_prop_torch (src/tools/uncertainty_prop.py:372-399),
//       T = c Z0,  M = sym(Ky_inv - beta beta^T) o exp(-1/4 d^2_Lambda) sf^4 (pack.hip),  rows and columns use the same
//       transform h = Cm (u - x), Cm^T Cm = (Lambda/2 + S)^-1 / 8, so exp(-1/8 (G_ii + 2 G_ij + G_jj)) = exp(-|h_i + h_j|^2)
//       (:389).  M is stored upper-triangular with weight 2 off the diagonal: only i <= j is visited.
//   * cross-covariance unit (a, b) (unit >= ntri): beta_a^T Qt beta_b of covariance_prop_torch (:402-465) written as a
//       Gaussian product, M = beta_a beta_b^T o exp(-1/2 d^2_{La+Lb}) sfa^2 sfb^2, rows use Cm diag(w_a), columns
//       Cm diag(w_b) with Cm^T Cm = (S + Lab)^-1 / 2; all N^2 ordered pairs are visited.
// step.hip / moment.hip turn (Z0, Z1, Z2) into values and Jacobians, so forward value and gradient come out of ONE
// pass and nothing N x N is ever stored per call.
//
// Mapping (CDNA4, wave64): lane = row i; a workgroup of W waves owns 64*W rows and walks its j-range in chunks of 64
// columns whose transformed points q_j (for TB trajectories) are staged in LDS and read back wave-uniformly
// (broadcast, conflict-free).  M_ij is read as [j][i], i.e. 512 contiguous bytes per wave and j; TB trajectories
// share every M_ij load.  All accumulation is per-lane fp64; the cross-lane reduction happens once per workgroup in a
// fixed order and per-tile partials are written (not atomically added): results are bit-reproducible.
// The kernel is fp64-VALU-issue bound (exp = 10 of ~35 fp64 instructions per pair): fp64 MFMA was measured NOT to
// overlap with fp64 VALU work on MI355X and is not cheaper per FMA (profiles/r01/ubench_mfma_f64_overlap.txt), so the
// moment accumulation stays on the VALU; occupancy (TB = 2 -> 3+ waves/SIMD) matters more than reuse (TB = 4).
#pragma once
#include <cstdlib>
#include "gpmpc_internal.h"
#include "fast_exp.h"

template <int D, bool DIAG, bool GRAD>
struct PairTraits {
    static constexpr int DP = (D + 1) & ~1;   // LDS row stride (doubles), keeps rows 16-byte aligned
    static constexpr int NM = !GRAD ? 1 : (DIAG ? 1 + 2 * D : 1 + D + D * (D + 1) / 2);
};

template <int D, bool DIAG>
__device__ __forceinline__ void pair_transform(const double* __restrict__ prm, const double (&x)[D], double (&h)[D]) {
    // h = cvec - T x   (DIAG: T = diag(prm[D..2D)); otherwise T upper-triangular, row-major at prm[D..D+D*D))
    if (DIAG) {
#pragma unroll
        for (int k = 0; k < D; ++k) h[k] = fma(-prm[D + k], x[k], prm[k]);
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double s = prm[k];
#pragma unroll
            for (int l = k; l < D; ++l) s = fma(-prm[D + k * D + l], x[l], s);
            h[k] = s;
        }
    }
}

// NS2: second moments are accumulated for the first NS2 dimensions only (DIAG && GRAD).  In the rollout the
//      input variance of the action dimensions is a constant, so dT/ds_k is not needed for k >= state_dim.
template <int D, bool DIAG, bool GRAD, int TB, int NS2 = D>
__global__ __launch_bounds__(256) void gpmpc_pair_kernel(PairArgs A) {
    using TR = PairTraits<D, DIAG, GRAD>;
    constexpr int DP = TR::DP, NM = TR::NM;
    __shared__ __attribute__((aligned(16))) double s_hj[TB * 64 * DP];
    __shared__ double s_red[4 * TB * NM];
    __shared__ double s_tab[GPMPC_EXP_N];
    gpmpc_exp_table_to_lds(s_tab);                        // visible after the first barrier below

    // XCD-aware decode of the flat grid.  Workgroups are dealt round-robin over the 8 XCDs (each with its own
    // L2), so the XCD label L % 8 picks the work item (unit, tile) and all trajectory groups of an item run back to
    // back on ONE XCD: every M tile is pulled through a single L2 instead of all eight.  Bijective for any
    // item count (the ragged tail falls back to the plain order).  Placement only affects speed.
    int bg, wi;
    {
        const int groups = (A.B + TB - 1) / TB, items = A.nwork;
        const int L = blockIdx.x, full = (items >> 3) << 3;
        if (L < full * groups) { const int q = L >> 3; wi = (q / groups) * 8 + (L & 7); bg = q % groups; }
        else { const int Lt = L - full * groups; wi = full + Lt / groups; bg = Lt % groups; }
    }
    const int unit = A.work[wi * 4 + 0], i0 = A.work[wi * 4 + 1], j0 = A.work[wi * 4 + 2], j1 = A.work[wi * 4 + 3];
    const bool tri = unit < A.ntri;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int Np = A.Np;
    // colsplit (small batches, 64-row tiles run by 4 waves): all waves share the 64 rows and each takes a quarter of
    // every 64-column chunk, so a tile's latency is a quarter of a one-wave tile's
    const int iw0 = A.colsplit ? i0 : i0 + w * 64;            // first row of this wave
    const bool active = iw0 < Np;           // wave-uniform (Np is a multiple of 64)
    const int i = iw0 + lane;

    const double* __restrict__ prm[TB];     // row-side parameters; the column side sits jside_off doubles further
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
        int b = bg * TB + tb;
        b = b < A.B ? b : A.B - 1;          // pad the last group with a duplicate (its result is not written)
        prm[tb] = A.pp + ((size_t)b * A.nunits + unit) * A.pps;
    }

    double hi[TB][D];
    if (active) {
        double x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = A.XT[(size_t)k * Np + i];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) pair_transform<D, DIAG>(prm[tb], x, hi[tb]);
    } else {
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int k = 0; k < D; ++k) hi[tb][k] = 0.0;
    }

    double acc[TB][NM];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[tb][m] = 0.0;

    const double* __restrict__ Ma = A.M + (size_t)unit * Np * Np;

    for (int jc = j0; jc < j1; jc += 64) {
        __syncthreads();
        // stage q_j for this chunk: 64 columns x TB trajectories
        for (int idx = tid; idx < 64 * TB; idx += blockDim.x) {
            const int jj = idx & 63, tb = idx >> 6;
            double x[D], h[D];
#pragma unroll
            for (int k = 0; k < D; ++k) x[k] = A.XT[(size_t)k * Np + jc + jj];
            // tb is not a compile-time constant here: select the parameter block dynamically
            const double* p = prm[0];
#pragma unroll
            for (int t2 = 1; t2 < TB; ++t2) p = (tb == t2) ? prm[t2] : p;
            pair_transform<D, DIAG>(p + A.jside_off, x, h);
#pragma unroll
            for (int k = 0; k < D; ++k) s_hj[(tb * 64 + jj) * DP + k] = h[k];
        }
        __syncthreads();
        // symmetric unit: rows of this wave all below the chunk's columns -> every M_ij of the chunk is zero
        if (!active || (tri && jc + 63 < iw0)) continue;

        const double* __restrict__ Mc = Ma + (size_t)jc * Np + i;
        const int jb0 = A.colsplit ? w * 16 : 0, jb1 = A.colsplit ? w * 16 + 16 : 64;
#pragma unroll 1
        for (int jb = jb0; jb < jb1; jb += 4) {
            double mij[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) mij[q] = Mc[(size_t)(jb + q) * Np];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) {
                    const double* hj = &s_hj[(tb * 64 + jb + q) * DP];
                    double m[D], sq[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) { m[k] = hi[tb][k] + hj[k]; sq[k] = m[k] * m[k]; }
                    double s = sq[0];
#pragma unroll
                    for (int k = 1; k < D; ++k) s += sq[k];
                    const double P = mij[q] * gpmpc_exp_neg(s, s_tab);
                    acc[tb][0] += P;
                    if (GRAD) {
                        if (DIAG) {
#pragma unroll
                            for (int k = 0; k < D; ++k) {
                                acc[tb][1 + k] = fma(P, m[k], acc[tb][1 + k]);
                                if (k < NS2) acc[tb][1 + D + k] = fma(P, sq[k], acc[tb][1 + D + k]);
                            }
                        } else {
                            int o = 1 + D;
#pragma unroll
                            for (int k = 0; k < D; ++k) {
                                const double pm = P * m[k];
                                acc[tb][1 + k] += pm;
#pragma unroll
                                for (int l = k; l < D; ++l) { acc[tb][o] = fma(pm, m[l], acc[tb][o]); ++o; }
                            }
                        }
                    }
                }
            }
        }
    }

    // workgroup reduction, fixed order
    __syncthreads();
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double s = wave_sum(acc[tb][m]);
            if (lane == 0) s_red[(w * TB + tb) * NM + m] = s;
        }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    for (int idx = tid; idx < TB * NM; idx += blockDim.x) {
        const int tb = idx / NM, m = idx - tb * NM;
        const int b = bg * TB + tb;
        if (b < A.B) {
            double s = 0.0;
            for (int ww = 0; ww < nw; ++ww) s += s_red[(ww * TB + tb) * NM + m];
            A.part[((size_t)b * A.nwork + wi) * A.nm + m] = s;
        }
    }
}

template <int D, bool DIAG, bool GRAD, int TB, int NS2 = D>
static int launch_pair_one(int waves, const PairArgs& a, hipStream_t s) {
    dim3 grid(((a.B + TB - 1) / TB) * a.nwork), block(64 * waves);
    hipLaunchKernelGGL((gpmpc_pair_kernel<D, DIAG, GRAD, TB, NS2>), grid, block, 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("pair kernel launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

template <int D>
int gpmpc_launch_pair_D(bool diag, bool grad, int tb, int waves, const PairArgs& a, hipStream_t s) {
    if (a.nm != gpmpc_num_moments(D, diag, grad)) return GPMPC_E_ARG;
    if (diag && grad && tb == 2 && a.ns2 < D) {   // the rollout hot path: skip the action dims' second moments
        if (D >= 2 && a.ns2 == D - 1) return launch_pair_one<D, true, true, 2, (D >= 2 ? D - 1 : D)>(waves, a, s);
        if (D >= 3 && a.ns2 == D - 2) return launch_pair_one<D, true, true, 2, (D >= 3 ? D - 2 : D)>(waves, a, s);
    }
#define GPMPC_PAIR_CASE(DG, GR)                                                  \
    if (diag == DG && grad == GR) {                                              \
        if (tb == 1) return launch_pair_one<D, DG, GR, 1>(waves, a, s);          \
        if (tb == 2) return launch_pair_one<D, DG, GR, 2>(waves, a, s);          \
        if (tb == 4) return launch_pair_one<D, DG, GR, 4>(waves, a, s);          \
        return GPMPC_E_ARG;                                                      \
    }
    GPMPC_PAIR_CASE(true, true)
    GPMPC_PAIR_CASE(true, false)
    if (!diag && grad) {   // full second moments: 1 + D + D(D+1)/2 accumulators per trajectory, keep TB <= 2
        if (tb == 1) return launch_pair_one<D, false, true, 1>(waves, a, s);
        if (tb == 2) return launch_pair_one<D, false, true, 2>(waves, a, s);
        return GPMPC_E_ARG;
    }
    GPMPC_PAIR_CASE(false, false)
#undef GPMPC_PAIR_CASE
    return GPMPC_E_ARG;
}
