// Pair kernel v2 for the rollout hot path (diagonal S, forward + gradient): the moment
// accumulation runs on the fp64 matrix cores, the VALU keeps only the exponent and exp().
//
// Same sum as pair_kernel.h (reference: src/tools/uncertainty_prop.py:372-399):
//     P_ij = M_ij exp(-|h_i + h_j|^2),   Z0 = sum P,  Z1_k = sum P m_k,  Z2_k = sum P m_k^2,   m = h_i + h_j
// Expanding m and grouping by row i:
//     with O[i][n] = sum_j P_ij G[j][n],   G[j] = [ h_j1..h_jD | 1 | h_j1^2..h_jD^2 | 0.. ]   (16 columns)
//     Z0   = sum_i O[i][D]
//     Z1_k = sum_i ( h_ik O[i][D] + O[i][k] )
//     Z2_k = sum_i ( h_ik^2 O[i][D] + O[i][D+1+k] + 2 h_ik O[i][k] )
// O = P (16 rows x 4 columns per step) x G (4 x 16) is exactly one v_mfma_f64_16x16x4_f64 per 64 pairs.
//
// Mapping (wave64): lane l -> A-operand element (row li = l & 15, k = lk = l >> 4), i.e. the lane evaluates the pair
// (i = ibase + 16 ri + li, j = jstep + lk): P_ij is produced in the register the MFMA reads it from, no data movement.
// B operand = G[jstep + lk][li], one ds_read_b64 per MFMA.  Each lane keeps RI rows (h_i resident) x TB trajectories,
// so one h_j read serves RI pairs and one M_ij load serves TB pairs.  Per pair the VALU issues
// 3D-1 (m, squares, sum) + 12 (table exp) + 1 (M_ij * e) fp64 slots; the 2D+1 moment FMAs per pair of v1 are gone.
// G is staged per 64-column chunk, double buffered (one barrier per chunk); M is prefetched one step ahead.
#pragma once
#include "gpmpc_internal.h"
#include "fast_exp.h"

typedef double gpmpc_v4f64 __attribute__((ext_vector_type(4)));

template <int D, int TB>
struct PairMfmaTraits {
    static constexpr int GW = (2 * D + 2) & ~1;
    // two G staging buffers; the epilogue reuses the area as 4 x (16 x 16) scratch tiles
    static constexpr int GAREA = (2 * TB * 64 * GW > 4 * 256) ? 2 * TB * 64 * GW : 4 * 256;
    static constexpr size_t LDS_BYTES = sizeof(double) * (GAREA + 4 * TB * (1 + 2 * D) + GPMPC_EXP_N);
};

template <int D, int TB, int RI>
__global__ __launch_bounds__(256) void gpmpc_pair_kernel_mfma(PairArgs A) {
    static_assert(2 * D + 1 <= 16, "G row must fit the 16 MFMA columns");
    constexpr int NM = 1 + 2 * D;
    constexpr int GW = PairMfmaTraits<D, TB>::GW;        // doubles per G row (even, >= 2D+1)
    constexpr int GBUF = TB * 64 * GW;                   // one staging buffer
    // all LDS in ONE dynamic array: [2][TB][64][GW] G | [4][TB][NM] reduction | [GPMPC_EXP_N] exp table
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    double* const s_G0 = s_dyn;
    double* const s_red = s_dyn + PairMfmaTraits<D, TB>::GAREA;
    double* const s_tab = s_red + 4 * TB * NM;

    const int bg = blockIdx.x, tile = blockIdx.y, a = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int i0 = A.tiles[tile * 3 + 0], j0 = A.tiles[tile * 3 + 1], j1 = A.tiles[tile * 3 + 2];
    const int Np = A.Np;
    const int ibase = i0 + w * 16 * RI;                  // first row of this wave
    const bool active = ibase < Np;                      // wave-uniform; Np is a multiple of 64 and 16*RI divides 64, so a wave's rows are all inside or all outside
    gpmpc_exp_table_to_lds(s_tab);

    const double* __restrict__ prm[TB];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
        int b = bg * TB + tb;
        b = b < A.B ? b : A.B - 1;
        prm[tb] = A.pp + ((size_t)b * A.ds + a) * A.pps;
    }

    // resident rows: h_i for RI row blocks x TB trajectories
    double hi[TB][RI][D];
#pragma unroll
    for (int ri = 0; ri < RI; ++ri) {
        const int i = ibase + ri * 16 + li;
        double x[D];
#pragma unroll
        for (int k = 0; k < D; ++k) x[k] = (active && i < Np) ? A.XT[(size_t)k * Np + i] : 0.0;
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int k = 0; k < D; ++k) hi[tb][ri][k] = fma(-prm[tb][D + k], x[k], prm[tb][k]);
    }

    gpmpc_v4f64 O[TB][RI];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) O[tb][ri] = (gpmpc_v4f64){0.0, 0.0, 0.0, 0.0};

    const double* __restrict__ Ma = A.M + (size_t)a * Np * Np;

    // stage G for a 64-column chunk into buffer `buf`
    auto stage = [&](int jc, int buf) {
        for (int idx = tid; idx < 64 * TB; idx += 256) {
            const int jj = idx & 63, tb = idx >> 6;
            const double* p = prm[0];
#pragma unroll
            for (int t2 = 1; t2 < TB; ++t2) p = (tb == t2) ? prm[t2] : p;
            double g[GW];
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double h = fma(-p[D + k], A.XT[(size_t)k * Np + jc + jj], p[k]);
                g[k] = h;
                g[D + 1 + k] = h * h;
            }
            g[D] = 1.0;
#pragma unroll
            for (int k = 2 * D + 1; k < GW; ++k) g[k] = 0.0;
            double* dst = s_G0 + buf * GBUF + (tb * 64 + jj) * GW;
#pragma unroll
            for (int k = 0; k < GW; ++k) dst[k] = g[k];
        }
    };

    stage(j0, 0);
    int buf = 0;
    for (int jc = j0; jc < j1; jc += 64, buf ^= 1) {
        __syncthreads();                                 // G[buf] staged; everyone is done reading G[buf ^ 1]
        if (jc + 64 < j1) stage(jc + 64, buf ^ 1);
        if (!active || jc + 63 < ibase) continue;        // every column of the chunk is left of every row of the wave

        double mcur[RI], mnext[RI];
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) mcur[ri] = Ma[(size_t)(jc + lk) * Np + ibase + ri * 16 + li];
#pragma unroll 2
        for (int jb = 0; jb < 16; ++jb) {
            const int jn = jc + 4 * (jb + 1 < 16 ? jb + 1 : jb) + lk;          // prefetch next step (clamped)
#pragma unroll
            for (int ri = 0; ri < RI; ++ri) mnext[ri] = Ma[(size_t)jn * Np + ibase + ri * 16 + li];
            if (jc + 4 * jb + 3 >= ibase) {              // wave-uniform: skip steps wholly below the diagonal
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) {
                    const double* grow = s_G0 + buf * GBUF + (tb * 64 + 4 * jb + lk) * GW;
                    double hj[D];
#pragma unroll
                    for (int k = 0; k < D; ++k) hj[k] = grow[k];
                    const double bop = (li < 2 * D + 1) ? grow[li < GW ? li : 0] : 0.0;
#pragma unroll
                    for (int ri = 0; ri < RI; ++ri) {
                        double s = 0.0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const double m = hi[tb][ri][k] + hj[k];
                            s = (k == 0) ? m * m : fma(m, m, s);
                        }
                        const double P = mcur[ri] * gpmpc_exp_neg(s, s_tab);
                        O[tb][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(P, bop, O[tb][ri], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int ri = 0; ri < RI; ++ri) mcur[ri] = mnext[ri];
        }
    }

    // epilogue: O tiles -> moments.  C layout of v_mfma_f64_16x16x4_f64: lane holds rows lk + 4r, column li.
    __syncthreads();
    double* scratch = s_G0 + w * 256;                    // 16 x 16 doubles per wave (G is dead now)
    double zacc[TB][NM];
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int m = 0; m < NM; ++m) zacc[tb][m] = 0.0;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int ri = 0; ri < RI; ++ri) {
#pragma unroll
            for (int r = 0; r < 4; ++r) scratch[(lk + 4 * r) * 16 + li] = O[tb][ri][r];
            __syncthreads();
            if (lane < 16) {                              // lane = row li, holds h_i of that row
                const double* orow = &scratch[lane * 16];
                const double o1 = orow[D];
                zacc[tb][0] += o1;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double h = hi[tb][ri][k];
                    zacc[tb][1 + k] += fma(h, o1, orow[k]);
                    zacc[tb][1 + D + k] += fma(h * h, o1, fma(2.0 * h, orow[k], orow[D + 1 + k]));
                }
            }
            __syncthreads();
        }
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double s = wave_sum(zacc[tb][m]);
            if (lane == 0) s_red[(w * TB + tb) * NM + m] = s;
        }
    __syncthreads();
    for (int idx = tid; idx < TB * NM; idx += 256) {
        const int tb = idx / NM, m = idx - tb * NM;
        const int b = bg * TB + tb;
        if (b < A.B) {
            double s = 0.0;
            for (int ww = 0; ww < 4; ++ww) s += s_red[(ww * TB + tb) * NM + m];
            A.part[(((size_t)b * A.ds + a) * A.ntiles + tile) * A.nm + m] = s;
        }
    }
}

template <int D, int TB, int RI>
static int launch_pair_mfma_one(const PairArgs& a, hipStream_t s) {
    dim3 grid((a.B + TB - 1) / TB, a.ntiles, a.ds), block(256);
    constexpr size_t lds = PairMfmaTraits<D, TB>::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipError_t ea = hipFuncSetAttribute((const void*)gpmpc_pair_kernel_mfma<D, TB, RI>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (ea != hipSuccess) { gpmpc_set_error("hipFuncSetAttribute(pair kernel mfma)", ea); return GPMPC_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gpmpc_pair_kernel_mfma<D, TB, RI>), grid, block, lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("pair kernel (mfma) launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

// variant: 0 -> TB=4, RI=2 (128-row tiles); 1 -> TB=2, RI=4 (256-row tiles); 2 -> TB=1, RI=4
template <int D>
int gpmpc_launch_pair_mfma_D(int variant, const PairArgs& a, hipStream_t s) {
    if (a.nm != 1 + 2 * D) return GPMPC_E_ARG;
    if constexpr (2 * D + 1 <= 16) {
        if (variant == 0) return launch_pair_mfma_one<D, 4, 2>(a, s);
        if (variant == 1) return launch_pair_mfma_one<D, 2, 4>(a, s);
        if (variant == 2) return launch_pair_mfma_one<D, 1, 4>(a, s);
    }
    return GPMPC_E_ARG;
}
