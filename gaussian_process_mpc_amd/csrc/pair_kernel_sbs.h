// Pair kernel for the rollout hot path when ALL GPs SHARE their length-scales ("shared lambda"): the regime every
// experiment of the reference runs in (src/experiments/pretrain_uncertainty.py:100-105, pretrain_pendulum.py:54-55,
// pretrain_cts_cartpole.py:42-43 set one lambda for every GP of the bundle).
//
// With lambda_a = lambda for all a, the transformed points h = sc o (u - x), the exponent |h_i + h_j|^2 and its exp are the
// same for the ds GPs of a (trajectory, step): only the folded weight M_a,ij differs (src/tools/uncertainty_prop.py:372-399
// evaluated ds times by src/dynamics.py:166-183 with identical A_part / Lambda_part).  The kernel of pair_kernel_sb.h is
// therefore re-grouped: ONE exponent and ONE table exp per pair, applied to NG weight loads,
//     e_ij = exp(-|h_i + h_j|^2),   P_a,ij = M_a,ij e_ij,   r_a,i += P_a,ij,  v_a,ik += P_a,ij h_jk,  w_a,ik += P_a,ij h_jk^2
// i.e. (1 + D) + 7 fp64-rate instructions per pair shared by NG GPs + (2 + D + ds) per pair and GP:
//     D = 5, ds = 4, NG = 4:  13/4 + 11 = 14.25 per pair-GP   (24 in pair_kernel_sb.h)
//     D = 3, ds = 2, NG = 2:  11/2 +  7 = 12.5                (18)
//     D = 7, ds = 6, NG = 3:  15/3 + 15 = 20                  (30)
// The NG independent accumulation chains per lane replace the two-trajectories-per-wave interleave of pair_kernel_sb.h
// as the source of instruction-level parallelism; NG x (1 + D + ds) fp64 accumulators set the occupancy (NG = 4 at C3:
// 40 accumulators, 4 waves/SIMD).  The column rows G are written ONCE per trajectory (not per GP) by the head kernel.
// Everything else -- wave-uniform column data through scalar loads, M_ij as [j][i] buffer loads with the column offset in
// the scalar offset, XCD-aware flat grid, fixed-order reduction, per-tile partials written not added -- is as in
// pair_kernel_sb.h, and the per-GP sums are the same sums in the same order: results agree with the distinct-lambda
// kernel to rounding of the exponent only.
#pragma once
#include "gpmpc_internal.h"
#include "fast_exp.h"

template <int D, int NS2>
struct PairSbsTraits {
    static constexpr int GW = (D + 1 + NS2 + 1) & ~1;     // doubles per G row, as PairSbTraits
};

// waves per SIMD the instance is compiled for: 2 VGPRs per accumulator + ~48 for everything else, out of 512
constexpr int gpmpc_sbs_waves(int NG, int NA) {
    // (row transform and exp temporaries alone take ~50; the two-column batches of round 5 hold 2 NG weights + two sets of exp temporaries,
    // groups of four GPs the next batch's weights too -- they run at three waves per SIMD: at four the row transform spilled into the loop)
    const int staged = NA > 1 ? 2 * NG + 12 + (NG >= 4 ? 4 * NG : 0) : 0;
    const int r0 = 2 * NG * NA + 48 + staged, regs = r0 < 80 ? 80 : r0;
    int w = 8;
    while (w > 1 && ((512 / w) / 8) * 8 < regs) --w;              // registers come in blocks of 8: the cap at w waves per SIMD is 8 floor(64 / w)
    return w;
}

template <int D, int NG, int NS2, bool GRAD, bool FIRST = false>
__global__ __launch_bounds__(256, gpmpc_sbs_waves(NG, GRAD ? (FIRST ? 1 + D - NS2 : 1 + D + NS2) : 1))
void gpmpc_pair_kernel_sbs(PairSbsArgs A) {
    constexpr int GW = PairSbsTraits<D, NS2>::GW;
    constexpr int NM = GRAD ? 1 + 2 * D : 1, NA = GRAD ? 1 + D + NS2 : 1;
    __shared__ double s_red[16 * NG * NM];        // [wave][row of 16 lanes][GP of the group][moment]
    __shared__ double s_tab[GPMPC_EXP_N];
    gpmpc_exp_table_to_lds(s_tab);

    // XCD-aware decode, as pair_kernel_sb.h with one trajectory per workgroup
    int b, wi;
    {
        const int groups = A.B, items = A.nwork, R = A.rgroup;
        const int L = blockIdx.x, full = (items / (8 * R)) * (8 * R);
        if (L < full * groups) {
            const int q = L >> 3, ir = q % R, t = q / R;
            b = t % groups;
            wi = ((t / groups) * R + ir) * 8 + (L & 7);
        } else { const int Lt = L - full * groups; wi = full + Lt / groups; b = Lt % groups; }
    }
    const int grp = A.work[wi * 4 + 0], i0 = A.work[wi * 4 + 1], j0 = A.work[wi * 4 + 2], tile = A.work[wi * 4 + 3];
    const int Np = A.Np;
    const int ncw = *(const int __attribute__((address_space(4)))*)A.ncol;      // columns that carry weight (N rounded up to 8): pair_kernel_sb.h
    const int j1 = j0 + A.jt < ncw ? j0 + A.jt : ncw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int iw0 = i0 + w * 64;                          // first row of this wave (wave-uniform)

    const double* __restrict__ prm = A.pp + (size_t)b * A.ds * A.pps;          // the transform is the same for every GP
    const double* __restrict__ Gt = A.G + (size_t)b * Np * GW;
    double hi2[D], qi;
    {
        const int i = iw0 + lane;
        double q = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double x = (iw0 < Np) ? A.XT[(size_t)k * Np + i] : 0.0;
            const double h = fma(-prm[D + k], x, prm[k]);
            hi2[k] = (2.0 * GPMPC_EXP_NEG_INV_C) * h;     // exponent carried as s * N/ln2 (fast_exp.h)
            q = fma(h, h, q);
        }
        qi = GPMPC_EXP_NEG_INV_C * q;
    }

    double acc[NG][NA];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int m = 0; m < NA; ++m) acc[g][m] = 0.0;

    __syncthreads();                                      // exp table ready

    if (iw0 < Np) {
        const int jstart = j0 > (iw0 & ~63) ? j0 : (iw0 & ~63);
        __amdgpu_buffer_rsrc_t Mrs[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // a partial last group (ds % NG != 0) re-reads the last GP; its results are not written
            const int a = grp * NG + g < A.ds ? grp * NG + g : A.ds - 1;
            Mrs[g] = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A.M + ((size_t)a * Np + jstart) * Np + iw0), 0, 0x7fffffff, 0x00020000);
        }
        const int lane8 = lane * 8;
#ifndef GPMPC_SBS_CU
#define GPMPC_SBS_CU 1       /* columns per loop iteration (A/B knob; 1 measured best, see profiles/r03/ab_shared_columns.txt) */
#endif
        constexpr int CU = GPMPC_SBS_CU;
#ifndef GPMPC_SBS_PREFETCH
#define GPMPC_SBS_PREFETCH 0     /* 1: the NG weights of column jc + 1 are requested before column jc is evaluated (register double buffer, 124 instead of
                                    114 VGPRs, still 4 waves per SIMD).  Measured -2...-4 % (C3 sizes with one lambda, B = 256: 26.5 -> 27.4 ms per batch;
                                    N = 1024, B = 256: 7.31 -> 7.60 ms; profiles/r04/ab_shared_prefetch.txt): A/B build only */
#endif
#ifndef GPMPC_SBS_STAGED
#define GPMPC_SBS_STAGED 1       /* round 5: columns in batches of two with the stages pinned (see below); 0: the one-column loop */
#endif
        if constexpr (GPMPC_SBS_STAGED && GRAD && !FIRST) {
            // Round 5 (as traj_persist.h / step_fused.h).  At 3-4 waves per SIMD a wave of the one-column loop -- weights requested, G row
            // requested, both waited for, exponent, table read, waited for, ~57 instructions of arithmetic -- leaves its SIMD to the others
            // for ~900 cycles per column, and they have 750 of arithmetic to fill them with: VALU issue 0.61 of its bound at C3 sizes with
            // one lambda (frac 0.49).  Two columns per batch: the 2 NG weights and both G rows requested together (one vector-memory and one
            // scalar-memory wait, overlapping), both exponents and table reads (one LDS wait), then 2 NG (2 + D + ds) accumulations.
            constexpr int KC = 2;
            typedef const double __attribute__((address_space(4))) gpmpc_cdouble;
            // one batch of KC columns: G rows -> exponents + table reads -> weights x exp, moment sums
            auto batch = [&](int jc, const double (*mij)[NG]) {
                double gr[KC][GW];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const gpmpc_cdouble* __restrict__ gp = (const gpmpc_cdouble*)(Gt + (size_t)(jc + c) * GW);
#pragma unroll
                    for (int k = 0; k < D + 1 + NS2; ++k) gr[c][k] = gp[k];
                }
                __builtin_amdgcn_sched_barrier(0);
                double fr[KC], Tv[KC];
                int ni[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    double sx = qi + gr[c][D];
#pragma unroll
                    for (int k = 0; k < D; ++k) sx = fma(hi2[k], gr[c][k], sx);
                    const double ax = __builtin_fabs(sx);          // gpmpc_exp_neg_scaled (fast_exp.h), split around the table read
                    ni[c] = (int)(-ax);
                    fr[c] = __builtin_amdgcn_fract(ax);
                    Tv[c] = s_tab[ni[c] & (GPMPC_EXP_N - 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    const double pq = fma(fr[c], fma(fr[c], GPMPC_EXP_A3, GPMPC_EXP_A2), GPMPC_EXP_A1);
                    const double e = ldexp(fma(Tv[c] * fr[c], pq, Tv[c]), ni[c] >> GPMPC_EXP_BITS);
#pragma unroll
                    for (int q = 0; q < NG; ++q) {
                        const double P = mij[c][q] * e;
                        acc[q][0] += P;
#pragma unroll
                        for (int k = 0; k < D; ++k) acc[q][1 + k] = fma(P, gr[c][k], acc[q][1 + k]);
#pragma unroll
                        for (int k = 0; k < NS2; ++k) acc[q][1 + D + k] = fma(P, gr[c][D + 1 + k], acc[q][1 + D + k]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto wload = [&](double (*m)[NG], int jrel) {
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int g = 0; g < NG; ++g) m[c][g] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[g], lane8, (jrel + c) * Np * 8, 0));
            };
            // groups of four GPs (three waves per SIMD: registers to spare): the weights of the NEXT batch are requested before this one is
            // evaluated -- two register sets in alternation, unconditional refill (the last batch re-requests itself)
            constexpr bool MPRE = NG >= 4;
            if constexpr (MPRE) {
                double ma[KC][NG], mb[KC][NG];
                wload(ma, 0);
                for (int jc = jstart; jc < j1; jc += 2 * KC) {    // (j1 - jstart is a multiple of 4: tile widths and N rounded up to 8)
                    wload(mb, jc - jstart + KC);
                    __builtin_amdgcn_sched_barrier(0);
                    batch(jc, ma);
                    wload(ma, (jc + 2 * KC < j1 ? jc + 2 * KC : jc) - jstart);
                    __builtin_amdgcn_sched_barrier(0);
                    batch(jc + KC, mb);
                }
            } else {
                for (int jc = jstart; jc < j1; jc += KC) {
                    double mij[KC][NG];
                    wload(mij, jc - jstart);
                    batch(jc, mij);
                }
            }
        } else {
        constexpr bool PRE = GPMPC_SBS_PREFETCH && CU == 1;
        double mnext[NG];
        if (PRE) {
#pragma unroll
            for (int g = 0; g < NG; ++g) mnext[g] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[g], lane8, 0, 0));
        }
        for (int jc = jstart; jc < j1; jc += CU) {
            double mij[CU][NG];
            if (PRE) {
                // this kernel runs 3-4 waves per SIMD (NG x (1 + D + ds) accumulators): a wave that issues its NG loads, waits for
                // them and only then evaluates the column leaves its SIMD to 2-3 others for a whole L2 round trip -- VALU-busy 0.81
                // against 0.88-0.95 of the distinct-lambda kernels (profiles/r03/pmc_C3_shared.json).  One column ahead costs 2 NG
                // registers.  The last iteration re-requests its own column (in bounds, unused).
                const int jn = jc + 1 < j1 ? jc + 1 : jc;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    mij[0][g] = mnext[g];
                    mnext[g] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[g], lane8, (jn - jstart) * Np * 8, 0));
                }
            } else {
#pragma unroll
            for (int c = 0; c < CU; ++c)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    mij[c][g] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(Mrs[g], lane8, (jc - jstart + c) * Np * 8, 0));
            }
            __builtin_amdgcn_sched_barrier(0);            // loads stay at the top of the iteration (pair_kernel_sb.h)
#pragma unroll
            for (int c = 0; c < CU; ++c) {
                typedef const double __attribute__((address_space(4))) gpmpc_cdouble;      // constant for the launch: scalar loads
                const gpmpc_cdouble* __restrict__ g = (const gpmpc_cdouble*)(Gt + (size_t)(jc + c) * GW);
                double s = qi + g[D];
#pragma unroll
                for (int k = 0; k < D; ++k) s = fma(hi2[k], g[k], s);
                const double e = gpmpc_exp_neg_scaled(s, s_tab);
#pragma unroll
                for (int q = 0; q < NG; ++q) {
                    const double P = mij[c][q] * e;
                    acc[q][0] += P;
                    if (GRAD) {
#pragma unroll
                        for (int k = 0; k < D; ++k) if (!FIRST || k >= NS2) acc[q][1 + k] = fma(P, g[k], acc[q][1 + k]);
#pragma unroll
                        for (int k = 0; k < NS2; ++k) if (!FIRST) acc[q][1 + D + k] = fma(P, g[D + 1 + k], acc[q][1 + D + k]);
                    }
                }
            }
        }
        }
    }

    // per-lane combination into the m-moments (as pair_kernel_sb.h), then the fixed-order workgroup reduction
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        double z[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) z[m] = 0.0;
        const double rs = acc[q][0];
        z[0] += rs;
        if (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const double h = (0.5 / GPMPC_EXP_NEG_INV_C) * hi2[k], v = acc[q][GRAD ? 1 + k : 0];
                if (!FIRST || k >= NS2) z[GRAD ? 1 + k : 0] += fma(h, rs, v);
                if (k < NS2 && !FIRST) z[GRAD ? 1 + D + k : 0] += fma(h * h, rs, fma(2.0 * h, v, acc[q][GRAD ? 1 + D + (k < NS2 ? k : 0) : 0]));
            }
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double s = wave_row_sum(z[m]);
            if ((lane & 15) == 0) s_red[((w * 4 + (lane >> 4)) * NG + q) * NM + m] = s;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NG * NM; idx += blockDim.x) {
        const int q = idx / NM, m = idx - q * NM;
        const int a = grp * NG + q;
        if (a < A.ds) {
            double s = 0.0;                       // fixed order: waves, each as (row 0 + row 1) + (row 2 + row 3)
            for (int ww = 0; ww < (int)(blockDim.x >> 6); ++ww) {
                const double* r4 = &s_red[(ww * 4 * NG + q) * NM + m];
                s += (r4[0] + r4[NG * NM]) + (r4[2 * NG * NM] + r4[3 * NG * NM]);
            }
            A.part[((size_t)b * A.ds * A.tiles + (size_t)a * A.tiles + tile) * A.nm + m] = s;
        }
    }
}

template <int D, int NG, int NS2, bool GRAD, bool FIRST = false>
static int launch_pair_sbs_one(const PairSbsArgs& a, hipStream_t s) {
    dim3 grid((unsigned)a.B * a.nwork), block(256);
    hipLaunchKernelGGL((gpmpc_pair_kernel_sbs<D, NG, NS2, GRAD, FIRST>), grid, block, 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("pair kernel (shared lambda) launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

// ns2 = state_dim (D - ns2 in {1, 2} action dimensions); ng = GPs per workgroup (gpmpc_sbs_group).
template <int D>
int gpmpc_launch_pair_sbs_D(bool grad, int ng, int ns2, const PairSbsArgs& a, hipStream_t s) {
    if (a.nm != (grad ? 1 + 2 * D : 1)) return GPMPC_E_ARG;
#define GPMPC_SBS_NS(NGV, NSV)                                                                              \
    if constexpr (NGV <= NSV) if (ng == NGV && ns2 == NSV) {                                                \
        if (!grad) return launch_pair_sbs_one<D, NGV, NSV, false>(a, s);                                    \
        if (a.first_step) return launch_pair_sbs_one<D, NGV, NSV, true, true>(a, s);                        \
        return launch_pair_sbs_one<D, NGV, NSV, true>(a, s);                                                \
    }
#define GPMPC_SBS_NG(NGV)                                                                                   \
    if constexpr (D >= 2) { GPMPC_SBS_NS(NGV, (D >= 2 ? D - 1 : 1)) }                                       \
    if constexpr (D >= 3) { GPMPC_SBS_NS(NGV, (D >= 3 ? D - 2 : 1)) }
    GPMPC_SBS_NG(2)
    GPMPC_SBS_NG(3)
    GPMPC_SBS_NG(4)
#undef GPMPC_SBS_NG
#undef GPMPC_SBS_NS
    return GPMPC_E_ARG;
}
