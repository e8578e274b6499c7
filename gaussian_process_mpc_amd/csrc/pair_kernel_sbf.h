// Scalar-broadcast pair kernel, general form: full input covariance S (upper-triangular transforms), variance AND
// cross-covariance units, full second moments.  Used by gpmpc_moment_match with Jacobians and by the full-covariance
// rollout (config 5).  Same regrouping as pair_kernel_sb.h, with different row / column transforms per unit:
//     P_ij = M_ij exp(-|p_i + q_j|^2),      |p_i + q_j|^2 = |p_i|^2 + |q_j|^2 + sum_k (2 p_ik) q_jk
//     r_i = sum_j P_ij,   v_ik = sum_j P_ij q_jk,   W_i,kl = sum_j P_ij q_jk q_jl      (k <= l < NS2)
//     Z0 = sum_i r_i,  Z1_k = sum_i (p_ik r_i + v_ik),  Z2_kl = sum_i (p_ik p_il r_i + p_ik v_il + p_il v_ik + W_i,kl)
// The column rows G[unit][j] = [q_j (D) | |q_j|^2 N/ln2 | q_jk q_jl (k <= l < NS2)] are written by k_mom_prep (moment.hip)
// and fetched with scalar loads; reference: src/tools/uncertainty_prop.py:372-399 (variance), :402-465 (covariance).
// NS2 < D (the rollout: NS2 = state_dim) leaves Z2_kl for k or l >= NS2 at zero: the transforms are upper
// triangular, so dT/dS_rc for r, c < NS2 only needs Z2_kl with k <= r, l <= c.
#pragma once
#include <cstdlib>
#include "gpmpc_internal.h"
#include "fast_exp.h"

// CU: columns per loop iteration (their M_ij loads issued together at the top).  1 for launches that fill the chip (C5: the other
// waves of the SIMD hide a column's round trips); 2 / 4 for the small batches of the two-launch rollout form (fullcov.hip), where a
// wave's column chain -- scalar loads of the G row, the M_ij load, the dependent exponent -- IS the run time of the launch.
#ifndef GPMPC_SBF_NT
#define GPMPC_SBF_NT 0          // cache policy of the weight loads of the multi-column (small-batch) instances: 0 default | 2 non-temporal
#endif
#ifndef GPMPC_SBF_MINWG
#define GPMPC_SBF_MINWG 1       // A/B: 6 asks the compiler for six workgroups per CU from the two-column instances at D <= 5 (80 instead of 84 VGPRs,
#endif                          // 6 instead of 5 waves per SIMD): measured SLOWER, N = 2048, B = 1 1.67 -> 1.79 ms (profiles/r04/fullcov_small_batch_ab.txt)
template <int D, int NS2, bool GRAD, int CU = 1>
__global__ __launch_bounds__(256, (CU == 2 && GRAD && D <= 5) ? GPMPC_SBF_MINWG : 1) void gpmpc_pair_kernel_sbf(PairSbfArgs A) {
    constexpr int NW = NS2 * (NS2 + 1) / 2;
    constexpr int GW = (D + 1 + NW + 1) & ~1;
    constexpr int NM = GRAD ? 1 + D + D * (D + 1) / 2 : 1, NA = GRAD ? 1 + D + NW : 1;
    __shared__ double s_red[16 * NM];             // [wave][row of 16 lanes][moment]
    __shared__ double s_tab[GPMPC_EXP_N];
    gpmpc_exp_table_to_lds(s_tab);

    int b, wi;                                            // XCD-aware decode, see pair_kernel.h (TB = 1)
    {
        const int groups = A.B, items = A.nwork;
        const int L = blockIdx.x, full = (items >> 3) << 3;
        if (L < full * groups) { const int q = L >> 3; wi = (q / groups) * 8 + (L & 7); b = q % groups; }
        else { const int Lt = L - full * groups; wi = full + Lt / groups; b = Lt % groups; }
    }
    const int unit = A.work[wi * 4 + 0], i0 = A.work[wi * 4 + 1], j0 = A.work[wi * 4 + 2], j1 = A.work[wi * 4 + 3];
    const bool tri = unit < A.ntri;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform -> scalar loads of the G rows
    const int Np = A.Np;
    const int iw0 = i0 + w * 64;
    const bool active = iw0 < Np;
    const int i = iw0 + lane;

    const double* __restrict__ prm = A.pp + ((size_t)b * A.nunits + unit) * A.pps;     // row-side transform
    const double* __restrict__ G = A.G + ((size_t)b * A.nunits + unit) * Np * GW;
    double p2[D], qi = 0.0;        // p_i is carried pre-scaled only (10 VGPRs at D = 5: 70 -> 7 waves/SIMD, 60 -> 8)
    {
        double x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = active ? A.XT[(size_t)k * Np + i] : 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double s = prm[k];
#pragma unroll
            for (int l = k; l < D; ++l) s = fma(-prm[D + k * D + l], x[l], s);
            p2[k] = (2.0 * GPMPC_EXP_NEG_INV_C) * s;               // exponent carried as s * N/ln2 (fast_exp.h)
            qi = fma(s, s, qi);
        }
        qi *= GPMPC_EXP_NEG_INV_C;
    }
    double acc[NA];
#pragma unroll
    for (int m = 0; m < NA; ++m) acc[m] = 0.0;

    const double* __restrict__ Ma = A.M + (size_t)unit * Np * Np;
    __syncthreads();                                      // exp table ready

    if (active) {
        const int jdiag = iw0 & ~63;
        const int jstart = (tri && jdiag > j0) ? jdiag : j0;   // symmetric units: zero weight left of the diagonal chunk
        const __amdgpu_buffer_rsrc_t Mrs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Ma + (size_t)jstart * Np + iw0), 0, 0x7fffffff, 0x00020000);
        const int lane8 = lane * 8;
        if (CU == 1) {
            for (int jc = jstart; jc < j1; ++jc) {           // one column per iteration: see GPMPC_SB_CU in pair_kernel_sb.h
                const double mij = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jc - jstart) * Np * 8, 0));
                const double* __restrict__ g = G + (size_t)jc * GW;              // wave-uniform address -> SGPRs
                double s = qi + g[D];
#pragma unroll
                for (int k = 0; k < D; ++k) s = fma(p2[k], g[k], s);
                const double P = mij * gpmpc_exp_neg_scaled(s, s_tab);
                acc[0] += P;
                if (GRAD) {
#pragma unroll
                    for (int k = 0; k < D; ++k) acc[1 + k] = fma(P, g[k], acc[1 + k]);
#pragma unroll
                    for (int k = 0; k < NW; ++k) acc[1 + D + k] = fma(P, g[D + 1 + k], acc[1 + D + k]);
                }
            }
        } else {
            // (column ranges of the work lists are multiples of 64 long, and so is jstart - j0)
            // Software pipeline: the M_ij loads of the NEXT group of CU columns are in flight while this group is evaluated (a wave
            // that waits for its loads at the top of every iteration spends one HBM latency per CU columns: N = 2048, B = 1 68 us per
            // launch at 3.9 TB/s).  The refill is unconditional (the last iteration re-reads its own columns): under a branch the
            // compiler waits for ALL outstanding loads instead of counting them.
            // Two register sets in alternation (a copy "this = next" would make the compiler wait for the loads it copies).
            double ma[CU], mb[CU];
            auto columns = [&](const int jc, const double (&mij)[CU]) {
#pragma unroll
                for (int q = 0; q < CU; ++q) {
                    typedef const double __attribute__((address_space(4))) gpmpc_cdouble;       // scalar loads whatever the compiler can prove (pair_kernel_sb.h)
                    const gpmpc_cdouble* __restrict__ g = (const gpmpc_cdouble*)(G + (size_t)(jc + q) * GW);
                    double s = qi + g[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) s = fma(p2[k], g[k], s);
                    const double P = mij[q] * gpmpc_exp_neg_scaled(s, s_tab);
                    acc[0] += P;
                    if (GRAD) {
#pragma unroll
                        for (int k = 0; k < D; ++k) acc[1 + k] = fma(P, g[k], acc[1 + k]);
#pragma unroll
                        for (int k = 0; k < NW; ++k) acc[1 + D + k] = fma(P, g[D + 1 + k], acc[1 + D + k]);
                    }
                }
            };
#pragma unroll
            for (int q = 0; q < CU; ++q)
                ma[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, q * Np * 8, GPMPC_SBF_NT));     // (in bounds for any jstart <= Np - 64)
            for (int jc = jstart; jc < j1; jc += 2 * CU) {
#pragma unroll
                for (int q = 0; q < CU; ++q)
                    mb[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jc - jstart + CU + q) * Np * 8, GPMPC_SBF_NT));
                __builtin_amdgcn_sched_barrier(0);               // the loads stay ahead of the arithmetic (pair_kernel_sb.h)
                columns(jc, ma);
                const int jn = jc + 2 * CU < j1 ? jc + 2 * CU : jc;
#pragma unroll
                for (int q = 0; q < CU; ++q)
                    ma[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs, lane8, (jn - jstart + q) * Np * 8, GPMPC_SBF_NT));
                __builtin_amdgcn_sched_barrier(0);
                columns(jc + CU, mb);
            }
        }
    }

    double z[NM];
    z[0] = acc[0];
    if (GRAD) {
        const double r = acc[0];
        double p[D];
#pragma unroll
        for (int k = 0; k < D; ++k) p[k] = (0.5 / GPMPC_EXP_NEG_INV_C) * p2[k];
#pragma unroll
        for (int k = 0; k < D; ++k) z[1 + k] = fma(p[k], r, acc[GRAD ? 1 + k : 0]);
        int o = 1 + D, ow = 0;
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int l = k; l < D; ++l) {
                double v = 0.0;
                if (k < NS2 && l < NS2) {
                    // position of (k,l), k <= l < NS2, in the packed W list (row-major upper triangle of NS2 x NS2)
                    const int idx = k * NS2 - k * (k - 1) / 2 + (l - k);
                    v = fma(p[k] * p[l], r, fma(p[k], acc[GRAD ? 1 + l : 0], fma(p[l], acc[GRAD ? 1 + k : 0], acc[GRAD ? 1 + D + idx : 0])));
                }
                z[GRAD ? o : 0] = v;
                ++o; (void)ow;
            }
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const double s = wave_row_sum(z[m]);                              // see gpmpc_internal.h
        if ((lane & 15) == 0) s_red[(w * 4 + (lane >> 4)) * NM + m] = s;
    }
    __syncthreads();
    for (int m = tid; m < NM; m += blockDim.x) {
        double s = 0.0;
        for (int ww = 0; ww < (int)(blockDim.x >> 6); ++ww) {
            const double* r4 = &s_red[ww * 4 * NM + m];
            s += (r4[0] + r4[NM]) + (r4[2 * NM] + r4[3 * NM]);
        }
        const size_t stride = A.pstride > 0 ? A.pstride : A.nwork;        // (slots per trajectory: more than this launch's items with one lambda)
        A.part[((size_t)b * stride + wi) * A.nm + m] = s;
        if (m == 0 && A.part0) A.part0[(size_t)b * stride + wi] = s;
    }
}

template <int D, int NS2, bool GRAD>
static int launch_pair_sbf_one(int waves, const PairSbfArgs& a, hipStream_t s) {
    dim3 grid(a.B * a.nwork), block(64 * waves);
    // the multi-column instances exist for the rollout's shapes (NS2 = state_dim = D - 1, D - 2)
    if (a.cu == 2 && NS2 < D) hipLaunchKernelGGL((gpmpc_pair_kernel_sbf<D, NS2, GRAD, (NS2 < D ? 2 : 1)>), grid, block, 0, s, a);
    else if (a.cu == 4 && NS2 < D) hipLaunchKernelGGL((gpmpc_pair_kernel_sbf<D, NS2, GRAD, (NS2 < D ? 4 : 1)>), grid, block, 0, s, a);
    else
    hipLaunchKernelGGL((gpmpc_pair_kernel_sbf<D, NS2, GRAD>), grid, block, 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("pair kernel (scalar broadcast, full S) launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

template <int D>
int gpmpc_launch_pair_sbf_D(bool grad, int ns2, int waves, const PairSbfArgs& a, hipStream_t s) {
    if (a.nm != (grad ? 1 + D + D * (D + 1) / 2 : 1)) return GPMPC_E_ARG;
#define GPMPC_SBF_CASE(GR)                                                                                   \
    if (grad == GR) {                                                                                        \
        if (ns2 == D) return launch_pair_sbf_one<D, D, GR>(waves, a, s);                                     \
        if (D >= 2 && ns2 == D - 1) return launch_pair_sbf_one<D, (D >= 2 ? D - 1 : D), GR>(waves, a, s);   \
        if (D >= 3 && ns2 == D - 2) return launch_pair_sbf_one<D, (D >= 3 ? D - 2 : D), GR>(waves, a, s);   \
        return GPMPC_E_ARG;                                                                                  \
    }
    GPMPC_SBF_CASE(true)
    GPMPC_SBF_CASE(false)
#undef GPMPC_SBF_CASE
    return GPMPC_E_ARG;
}
