// Cross-covariance units of the full-covariance rollout when ALL GPs SHARE their length-scales (round 5): every cross unit (a, b),
// a < b, of pair_kernel_sbf.h then has the same row transform p_i, the same column rows q_j and -- up to beta_a,i beta_b,j sf_a^2 sf_b^2 --
// the same weight matrix (pack.hip::k_pack_cross: E_ij = exp(-1/2 sum_k (x_ik - x_jk)^2 / (lambda_ak + lambda_bk)), here lambda_a + lambda_b =
// 2 lambda for every pair).  Instead of ds (ds - 1) / 2 streamed N x N weight matrices and as many exp per pair (201 MB and six at C5's
// sizes), ONE pass over the pairs evaluates
//     Q_ij = exp(-(1/4 sum_k (x_ik - x_jk)^2 / lambda_k + |p_i + q_j|^2))                 (one exponent: two quadratic forms, one table exp)
//     T_c,i = sum_j w_ij beta_c,j Q_ij [1 | q_j | q_jk q_jl]                               (c = 0 .. ds-1: ds accumulator sets per row i; upper triangle, below)
// with NO weight stream (x_j and beta_c,j travel in a constant column row of the pack, wave-uniform scalar loads), and the moments of
// unit (a, b) are  sf_a^2 sf_b^2 sum_i [beta_a,i zf(p_i; T_b,i) + beta_b,i zf(p_i; T_a,i)]  with zf the row-side combination of pair_kernel_sbf.h -- written into the
// partial-sum slots the head kernel (fullcov.hip::k_fc_head) reduces per unit, so nothing downstream changes.  The variance units keep
// their own weight matrices (they hold K_a^-1) and pair_kernel_sbf.h.  Reference: src/tools/uncertainty_prop.py:402-465 (covariance_prop_torch
// evaluated for every pair of GPs with the same Lambda); the setting of every experiment of the reference (pretrain_uncertainty.py:100-105).
#pragma once
#include "gpmpc_internal.h"
#include "fast_exp.h"

// One wave per 64 x 64 (or 64 x 16: small launches) tile of the UPPER TRIANGLE (the shared transform makes Q_ij f(p_i + q_j) symmetric in (i, j), so
//     S_ab = sum_ij beta_a,i beta_b,j W_ij = sum_i [beta_a,i T_b,i + beta_b,i T_a,i],   T_c,i = sum_{j >= i} w_ij beta_c,j W_ij,  w_ii = 1/2, w_ij = 1 (j > i)
// -- half the pairs of the full square, all ds accumulator sets); four tiles per workgroup: one exp table in LDS for four waves.
__device__ __forceinline__ void gpmpc_sbfx_tri_decode(int q, int T, int* r_out, int* c_out) {      // (as step_fused.h::gpmpc_tri_decode)
    const float tt = 2.0f * T + 1.0f;
    int r = (int)((tt - sqrtf(tt * tt - 8.0f * (float)q)) * 0.5f);
    r = r < 0 ? 0 : (r > T - 1 ? T - 1 : r);
    while (r > 0 && r * T - r * (r - 1) / 2 > q) --r;
    while (r + 1 < T && (r + 1) * T - (r + 1) * r / 2 <= q) ++r;
    *r_out = r;
    *c_out = r + (q - (r * T - r * (r - 1) / 2));
}

template <int D, int NS2, bool GRAD>
__global__ __launch_bounds__(256, 2) void gpmpc_pair_kernel_sbfx(PairSbfxArgs A) {
    constexpr int NW = NS2 * (NS2 + 1) / 2;
    constexpr int GW = (D + 1 + NW + 1) & ~1;                  // column rows of pair_kernel_sbf.h
    constexpr int RW = (D + 1 + NS2 + 1) & ~1;                 // constant rows of the pack: [x_j (D) | C/4 sum x_jk^2 / lambda_k | beta_c,j (ds)]
    constexpr int NM = GRAD ? 1 + D + D * (D + 1) / 2 : 1, NA = GRAD ? 1 + D + NW : 1;
    constexpr int NP = NS2 * (NS2 - 1) / 2;
    __shared__ double s_tab[GPMPC_EXP_N];
    __shared__ double s_redw[4][4 * NP * NA];                  // [wave][row of 16 lanes][pair][non-zero moment]
    gpmpc_exp_table_to_lds(s_tab);

    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_r = (int)blockIdx.x * 4 + w;
    const bool live = tile_r < A.ntile;                        // (the last workgroup of a tile count that is no multiple of 4)
    const int tile = live ? tile_r : A.ntile - 1;
    double* s_red = s_redw[w];
    const int Np = A.Np;
    // tile -> (64-row block ti, 64-column block tjx >= ti, piece of A.jt columns of that block)
    int ti, tjx, i0, j0, jn;
    if (A.tiles) {                                             // up to 256 columns of one row block, listed by the pack (wave-uniform: scalar loads)
        const int* tl = A.tiles + 3 * tile;
        ti = __builtin_amdgcn_readfirstlane(tl[0]); j0 = __builtin_amdgcn_readfirstlane(tl[1]); jn = __builtin_amdgcn_readfirstlane(tl[2]);
        i0 = ti * 64; tjx = j0 >> 6;
    } else {
    const int per = 64 / A.jt, blk = tile / per, sub = tile - blk * per;
    // (measured and dropped: a column-major block order -- the four waves of a workgroup on the SAME column rows -- and three waves per SIMD
    // (12 VGPR spills): both level at B = 2 ... 256, N = 2048: the kernel is near its issue bound there, a third of it prologue and row sums)
    gpmpc_sbfx_tri_decode(blk, Np >> 6, &ti, &tjx);
    ti = __builtin_amdgcn_readfirstlane(ti); tjx = __builtin_amdgcn_readfirstlane(tjx);
    i0 = ti * 64; j0 = tjx * 64 + sub * A.jt; jn = A.jt;
    }
    const int i = i0 + lane;                                   // (Np is a multiple of 64)
    const double* __restrict__ prm = A.pp + ((size_t)b * A.nunits + A.unit0) * A.pps;
    const double* __restrict__ G = A.G + ((size_t)b * A.nunits + A.unit0) * Np * GW;
    double p2[D], xe[D], qi = 0.0, bi[NS2];
    {
        double x[D], e = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = A.XT[(size_t)k * Np + i];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double s = prm[k];
#pragma unroll
            for (int l = k; l < D; ++l) s = fma(-prm[D + k * D + l], x[l], s);
            p2[k] = (2.0 * GPMPC_EXP_NEG_INV_C) * s;
            qi = fma(s, s, qi);
            const double il = 1.0 / A.lam[k];                                  // (one lambda for all GPs: GP 0's)
            xe[k] = (-0.5 * GPMPC_EXP_NEG_INV_C) * x[k] * il;
            e = fma(x[k] * x[k], il, e);
        }
        qi = GPMPC_EXP_NEG_INV_C * qi + (0.25 * GPMPC_EXP_NEG_INV_C) * e;
#pragma unroll
        for (int c = 0; c < NS2; ++c) bi[c] = i < A.N ? A.beta[(size_t)c * Np + i] : 0.0;
    }
    double acc[NS2][NA];
#pragma unroll
    for (int c = 0; c < NS2; ++c)
#pragma unroll
        for (int m = 0; m < NA; ++m) acc[c][m] = 0.0;
    __syncthreads();                                           // exp table ready

    typedef const double __attribute__((address_space(4))) gpmpc_cdouble;      // wave-uniform rows: scalar loads
    auto column = [&](const int jc, const double wgt, const bool weighted) {
        const gpmpc_cdouble* __restrict__ g = (const gpmpc_cdouble*)(G + (size_t)jc * GW);
        const gpmpc_cdouble* __restrict__ r = (const gpmpc_cdouble*)(A.rows + (size_t)jc * RW);
        double s = qi + g[D] + r[D];
#pragma unroll
        for (int k = 0; k < D; ++k) s = fma(p2[k], g[k], s);
#pragma unroll
        for (int k = 0; k < D; ++k) s = fma(xe[k], r[k], s);
        double E = gpmpc_exp_neg_scaled(s, s_tab);
        if (weighted) E *= wgt;
#pragma unroll
        for (int c = 0; c < NS2; ++c) {
            const double P = r[D + 1 + c] * E;
            acc[c][0] += P;
            if (GRAD) {
#pragma unroll
                for (int k = 0; k < D; ++k) acc[c][GRAD ? 1 + k : 0] = fma(P, g[k], acc[c][GRAD ? 1 + k : 0]);
#pragma unroll
                for (int k = 0; k < NW; ++k) acc[c][GRAD ? 1 + D + k : 0] = fma(P, g[D + 1 + k], acc[c][GRAD ? 1 + D + k : 0]);
            }
        }
    };
    {   // columns inside the row block's diagonal 64 x 64 block: column j of row i counts once (j > i), half (j = i) or not at all; then the rest
        const int jend = j0 + jn, jw = (tjx > ti) ? j0 : (jend < i0 + 64 ? jend : i0 + 64);      // (wave-uniform)
        for (int jc = j0; jc < jw; ++jc) { const int jl = jc - i0; column(jc, jl > lane ? 1.0 : (jl == lane ? 0.5 : 0.0), true); }
        for (int jc = jw; jc < jend; ++jc) column(jc, 1.0, false);
    }

    // row-side combination in place (pair_kernel_sbf.h, same expressions): acc[c] -> the 1 + D + NW moments of column GP c that can be non-zero
    // (Z2_kl with k or l >= NS2 is zero: the transforms are upper triangular)
    if (GRAD) {
        double p[D];
#pragma unroll
        for (int k = 0; k < D; ++k) p[k] = (0.5 / GPMPC_EXP_NEG_INV_C) * p2[k];
#pragma unroll
        for (int c = 0; c < NS2; ++c) {
            const double rr = acc[c][0];
            double v1[D];
#pragma unroll
            for (int k = 0; k < D; ++k) v1[k] = acc[c][GRAD ? 1 + k : 0];
#pragma unroll
            for (int k = 0; k < NS2; ++k)
#pragma unroll
                for (int l = k; l < NS2; ++l) {
                    const int idx = k * NS2 - k * (k - 1) / 2 + (l - k);
                    acc[c][GRAD ? 1 + D + idx : 0] = fma(p[k] * p[l], rr, fma(p[k], v1[l], fma(p[l], v1[k], acc[c][GRAD ? 1 + D + idx : 0])));
                }
#pragma unroll
            for (int k = 0; k < D; ++k) acc[c][GRAD ? 1 + k : 0] = fma(p[k], rr, v1[k]);
        }
    }
#pragma unroll
    for (int c = 1; c < NS2; ++c)
#pragma unroll
        for (int a = 0; a < c; ++a) {
            const int pr = a * NS2 - a * (a + 1) / 2 + (c - a - 1);            // lexicographic index of (a, c), a < c (pack.hip::pair_ab)
#pragma unroll
            for (int m = 0; m < NA; ++m) {
                const double sr = wave_row_sum(fma(bi[a], acc[c][m], bi[c] * acc[a][m]));
                if ((lane & 15) == 0) s_red[((lane >> 4) * NP + pr) * NA + m] = sr;
            }
        }
    __syncthreads();
    if (live)
    for (int e = lane; e < NP * NM; e += 64) {
        const int pr = e / NM, m = e - pr * NM;
        // moment m of the NM-long list -> its slot among the NA non-zero ones (or none): [Z0 | Z1_k, k < D | Z2_kl, k <= l < D row-major]
        int ma = -1;
        if (m <= D) ma = m;
        else {
            int k = 0, rem = m - 1 - D;
            while (rem >= D - k) { rem -= D - k; ++k; }
            const int l = k + rem;
            if (k < NS2 && l < NS2) ma = 1 + D + (k * NS2 - k * (k - 1) / 2 + (l - k));
        }
        double sum = 0.0;
        if (ma >= 0) { const int q = pr * NA + ma; sum = (s_red[q] + s_red[NP * NA + q]) + (s_red[2 * NP * NA + q] + s_red[3 * NP * NA + q]); }
        const int a = A.pair_ab[2 * pr], c = A.pair_ab[2 * pr + 1];
        const double s2 = A.sf[a] * A.sf[a] * A.sf[c] * A.sf[c];
        const size_t wi = (size_t)A.base + (size_t)pr * A.ntile + tile;
        const double out = s2 * sum;
        A.part[((size_t)b * A.pstride + wi) * A.nm + m] = out;
        if (m == 0 && A.part0) A.part0[(size_t)b * A.pstride + wi] = out;
    }
}

template <int D, int NS2, bool GRAD>
static int launch_pair_sbfx_one(const PairSbfxArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((gpmpc_pair_kernel_sbfx<D, NS2, GRAD>), dim3((a.ntile + 3) / 4, a.B), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("cross-unit pair kernel (one lambda, full S) launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

// the rollout's shapes only: ns2 = state_dim in {D - 1, D - 2}, 2 <= state_dim <= 4 (ds accumulator sets of 1 + D + ds (ds + 1) / 2 doubles per lane)
template <int D>
int gpmpc_launch_pair_sbfx_D(bool grad, int ns2, const PairSbfxArgs& a, hipStream_t s) {
    if (a.nm != (grad ? 1 + D + D * (D + 1) / 2 : 1) || ns2 < 2 || ns2 > 4 || a.ntile < 1 || (a.jt != 64 && a.jt != 16 && !(a.jt == 256 && a.tiles))) return GPMPC_E_ARG;
#define GPMPC_SBFX_CASE(GR, NSV)                                                                              \
    if constexpr ((NSV) >= 2 && (NSV) <= 4 && (NSV) < D) if (grad == GR && ns2 == (NSV)) return launch_pair_sbfx_one<D, ((NSV) >= 2 && (NSV) <= 4 && (NSV) < D) ? (NSV) : 2, GR>(a, s);
    GPMPC_SBFX_CASE(true, D - 1) GPMPC_SBFX_CASE(true, D - 2) GPMPC_SBFX_CASE(false, D - 1) GPMPC_SBFX_CASE(false, D - 2)
#undef GPMPC_SBFX_CASE
    return GPMPC_E_ARG;
}
