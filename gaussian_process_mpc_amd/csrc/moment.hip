// Single-step exact moment matching for Gaussian inputs with a FULL covariance S, all ds GPs:
// the general-S form of mean_prop_torch / variance_prop_torch / covariance_prop_torch
// (src/tools/uncertainty_prop.py:296-338, :341-399, :402-465).
//
//   B = (S + Lambda)^-1,  c_m = sf^2 det(Lambda^-1 S + I)^-1/2,  mu = c_m sum_i beta_i exp(-1/2 v_i^T B v_i)
//   A = (Lambda/2 + S)^-1, c = det(2 Lambda^-1 S + I)^-1/2,  A/8 = Cm^T Cm (Cholesky),  h_i = Cm v_i
//   T = c Z0, var = sf^2 - T - mu^2
//   dmu/du = -B c_m S1,            dmu/dS = -1/2 mu B + 1/2 B (c_m S2) B
//   dT/du  = -4 c Cm^T Z1,         dT/dS  = -1/2 T A + 8 c Cm^T Z2 Cm        (since A Cm^-1 = 8 Cm^T)
// with S1 = sum p_i v_i, S2 = sum p_i v_i v_i^T and Z* the pair-kernel moments in h-space.
#include "gpmpc_internal.h"
#include "fast_exp.h"
#include <cstdlib>

int gpmpc_timed_pair(int D, bool diag, bool grad, int tb, int waves, const PairArgs& a, hipStream_t s);
int gpmpc_timed_pair_sbf(int D, bool grad, int ns2, int waves, const PairSbfArgs& a, hipStream_t s);

#include "moment_dev.h"

template <int D>
__global__ __launch_bounds__(256) void k_mom_prep(MomArgs A) {
    __shared__ MomPrepLds<D> sh;
    // one workgroup per (query, unit): the per-unit set-up (D x D inverse, Cholesky factor, D^2 products) is the work of ONE
    // thread, and looping over the units inside a workgroup put ds of those latencies in a row (C5: 400 us per launch)
    const int q = blockIdx.x, unit = blockIdx.y;
#if defined(GPMPC_FC_STAMPS)
    if (threadIdx.x == 0) sh.stamp = -1;
#endif
    if (threadIdx.x < D) sh.u[threadIdx.x] = A.u[(size_t)q * D + threadIdx.x];
    if (threadIdx.x < D * D) sh.S[threadIdx.x] = A.S[(size_t)q * D * D + threadIdx.x];
    __syncthreads();
    const double none[D + 1] = {};
    mom_prep_body<D>(A, q, unit, sh, 0, 1, 3, false, none);
}

template <int D>
__global__ __launch_bounds__(256) void k_mom_finish(MomArgs A) {
    constexpr int NMX = 1 + D + D * (D + 1) / 2;
    __shared__ double s_z[(GPMPC_MAX_DS + GPMPC_MAX_PAIRS) * NMX];
    const int q = blockIdx.x, nm = A.nm, nunits = A.nunits;
    for (int idx = threadIdx.x; idx < nunits * nm; idx += blockDim.x) {
        const int u = idx / nm, m = idx - u * nm;
        const double* p = A.part + (size_t)q * A.nwork * nm + m;
        // four independent partial sums (fixed order): the loads of a unit's work items are independent, a single running sum
        // left one load in flight per thread (C5: 94 us per launch for 210 sums of 36-64 values)
        const int w0 = A.ustart[u], w1 = A.ustart[u + 1];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int wi = w0;
        for (; wi + 3 < w1; wi += 4) {
            s0 += p[(size_t)wi * nm]; s1 += p[(size_t)(wi + 1) * nm]; s2 += p[(size_t)(wi + 2) * nm]; s3 += p[(size_t)(wi + 3) * nm];
        }
        for (; wi < w1; ++wi) s0 += p[(size_t)wi * nm];
        s_z[idx] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    const int u = threadIdx.x;
    if (u >= nunits) return;
    mom_finish_unit<D>(A, A.sp, q, u, s_z + u * nm, true);
}


// ---------------------------------------------------------------------------
// Cross-covariance Cov[f_a, f_b] for a != b  (src/tools/uncertainty_prop.py:402-465):
//   beta_a^T Qt beta_b - mu_a mu_b,
//   Qt_ij = sfa^2 sfb^2 det(R)^-1/2 exp( -1/2 qa_i - 1/2 qb_j + 1/2 z_ij^T Am z_ij ),
//   R = S(La^-1 + Lb^-1) + I, Am = R^-1 S, z_ij = La^-1 (x_i - u) + Lb^-1 (x_j - u).
// One exponent per pair:  arg_ij = alpha_i + gamma_j + w_i . r_j  with
//   alpha_i = -1/2 qa_i + 1/2 p_i^T Am p_i, gamma_j = -1/2 qb_j + 1/2 r_j^T Am r_j, w_i = Am' p_i
// (consistent form: p = La^-1 (x-u) on the i side, r = Lb^-1 (x-u) on the j side, w_i = (Am + Am^T)/2 ... p_i).
// GPMPC_COV_BUG_COMPAT reproduces the reference's cross term z2_i^T Am z1_j (:446).
// One workgroup per (query, ordered GP pair a<b): lane = i, j broadcast from LDS in chunks.
// ---------------------------------------------------------------------------
// GRAD: also d Cov / d u and d Cov / d S (A.dcov_du [nq][ds][ds][D], A.dcov_dS [nq][ds][ds][D][D]; needs A.dmean_du / A.dmean_dS of
// this call), the derivatives autograd takes of the reference's expression (:402-465) -- in either mode: with
//   E_ij = 1/2 d_i^T P1 d_i + 1/2 d_j^T P2 d_j + d_i^T C d_j,  d = x - u,  P1 = -La^-1 + La^-1 Am La^-1,  P2 likewise with Lb,
//   C = Lpc Am Lrc  (Lpc, Lrc = La^-1, Lb^-1; swapped in the bug-compatible form),   W0 = sum_ij beta_a,i beta_b,j e^E_ij,
//   dW0/du = -(P1 m_i + P2 m_j + C m_j + C^T m_i),           m_i = sum w d_i,  m_j = sum w d_j
//   dW0/dS = R^-T sym(G) R^-1 (symmetrised),                  G = sum w (1/2 p p^T + 1/2 r r^T + pc rc^T),  dAm = R^-1 dS R^-T
//   K = sfa^2 sfb^2 det(R)^-1/2,  dK/dS = -1/2 K sym(L R^-1),  L = La^-1 + Lb^-1;      Cov = K W0 - mu_a mu_b.
// The moments are grouped by row (lane = i): per lane r = sum_j e, v = sum_j e r_j, W = sum_j e r_j r_j^T in the column loop.
template <int D, bool GRAD = false>
__global__ __launch_bounds__(256) void k_cross_cov(MomArgs A, const int* __restrict__ pairs, int npairs) {
    constexpr int NS = D * (D + 1) / 2, NV = GRAD ? 1 + 2 * D + NS : 1;
    __shared__ double s_u[D], s_S[D * D], s_Am[D * D], s_Ri[D * D];
    __shared__ double s_det;
    __shared__ __attribute__((aligned(16))) double s_j[64 * (2 * D + 2)];
    __shared__ double s_scr[16 * NV], s_out[NV];
    const int q = blockIdx.x / npairs, pr = blockIdx.x - q * npairs;
    const int a = pairs[2 * pr], b = pairs[2 * pr + 1];
    const bool bug = (A.flags & GPMPC_COV_BUG_COMPAT) != 0;
    if (threadIdx.x < D) s_u[threadIdx.x] = A.u[(size_t)q * D + threadIdx.x];
    if (threadIdx.x < D * D) { const int r = threadIdx.x / D, c = threadIdx.x - r * D; s_S[threadIdx.x] = 0.5 * (A.S[(size_t)q * D * D + r * D + c] + A.S[(size_t)q * D * D + c * D + r]); }
    __syncthreads();
    const double* la = A.lam + a * D; const double* lb = A.lam + b * D;
    if (threadIdx.x == 0) {
        double R[D * D], Ri[D * D];
        for (int r = 0; r < D; ++r) for (int c = 0; c < D; ++c) R[r * D + c] = s_S[r * D + c] * (1.0 / la[c] + 1.0 / lb[c]) + (r == c ? 1.0 : 0.0);
        s_det = small_inverse(D, R, Ri);
        for (int r = 0; r < D; ++r) for (int c = 0; c < D; ++c) { double s = 0.0; for (int l = 0; l < D; ++l) s += Ri[r * D + l] * s_S[l * D + c]; s_Am[r * D + c] = s; }
        if (GRAD) for (int e = 0; e < D * D; ++e) s_Ri[e] = Ri[e];
    }
    __syncthreads();
    double Am[D * D], u[D], ila[D], ilb[D];
#pragma unroll
    for (int e = 0; e < D * D; ++e) Am[e] = s_Am[e];
#pragma unroll
    for (int k = 0; k < D; ++k) { u[k] = s_u[k]; ila[k] = 1.0 / la[k]; ilb[k] = 1.0 / lb[k]; }

    double total = 0.0;
    double gm1[GRAD ? D : 1], gm2[GRAD ? D : 1], gG[GRAD ? NS : 1];       // sum w d_i | sum w r2_j | sym part of G (upper triangle)
    if (GRAD) {
#pragma unroll
        for (int k = 0; k < D; ++k) { gm1[k] = 0.0; gm2[k] = 0.0; }
#pragma unroll
        for (int e = 0; e < NS; ++e) gG[e] = 0.0;
    }
    constexpr int SJ = GRAD ? 2 * D + 2 : D + 2;                           // doubles per staged column
    for (int i0 = 0; i0 < A.Np; i0 += blockDim.x) {
        const int i = i0 + threadIdx.x;
        // i side
        double d[D], p1[D], pc[D], w[D], alpha = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { d[k] = (i < A.Np ? A.XT[(size_t)k * A.Np + i] : 0.0) - u[k]; p1[k] = ila[k] * d[k]; alpha = fma(-0.5 * d[k], p1[k], alpha); }
        // quadratic term uses z1 = La^-1 d on the i side in both modes; the cross term uses z2_i in bug mode
#pragma unroll
        for (int k = 0; k < D; ++k) pc[k] = bug ? ilb[k] * d[k] : p1[k];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            double s = 0.0, sc = 0.0;
#pragma unroll
            for (int l = 0; l < D; ++l) { s = fma(Am[k * D + l], p1[l], s); sc = fma(Am[l * D + k], pc[l], sc); }
            alpha = fma(0.5 * p1[k], s, alpha);
            w[k] = sc;                                     // (pc^T Am)_k : cross term = 2 * 1/2 * pc_i^T Am r_j
        }
        const double bi = i < A.Np ? A.beta[(size_t)a * A.Np + i] : 0.0;
        double rowsum = 0.0;
        double rv[GRAD ? D : 1], rW[GRAD ? NS : 1];
        if (GRAD) {
#pragma unroll
            for (int k = 0; k < D; ++k) rv[k] = 0.0;
#pragma unroll
            for (int e = 0; e < NS; ++e) rW[e] = 0.0;
        }
        for (int jc = 0; jc < A.Np; jc += 64) {
            __syncthreads();
            if (threadIdx.x < 64) {
                const int j = jc + threadIdx.x;
                double dj[D], r2[D], rc[D], gam = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) { dj[k] = A.XT[(size_t)k * A.Np + j] - u[k]; r2[k] = ilb[k] * dj[k]; gam = fma(-0.5 * dj[k], r2[k], gam); rc[k] = bug ? ila[k] * dj[k] : r2[k]; }
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double s = 0.0;
#pragma unroll
                    for (int l = 0; l < D; ++l) s = fma(Am[k * D + l], r2[l], s);
                    gam = fma(0.5 * r2[k], s, gam);
                    s_j[threadIdx.x * SJ + k] = rc[k];
                    if (GRAD) s_j[threadIdx.x * SJ + D + 2 + k] = r2[k];
                }
                s_j[threadIdx.x * SJ + D] = gam;
                s_j[threadIdx.x * SJ + D + 1] = A.beta[(size_t)b * A.Np + j];
            }
            __syncthreads();
            for (int jj = 0; jj < 64; ++jj) {
                const double* sj = &s_j[jj * SJ];
                double arg = alpha + sj[D];
#pragma unroll
                for (int k = 0; k < D; ++k) arg = fma(w[k], sj[k], arg);
                const double e = sj[D + 1] * exp(arg);
                rowsum += e;
                if (GRAD) {
                    int o = 0;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double er = e * sj[D + 2 + k];
                        rv[k] += er;
#pragma unroll
                        for (int l = k; l < D; ++l) { rW[o] = fma(er, sj[D + 2 + l], rW[o]); ++o; }
                    }
                }
            }
        }
        if (i < A.Np) {
            total = fma(bi, rowsum, total);
            if (GRAD) {
                // rc_j = (Lrc Lb) r2_j elementwise: sum_j e rc_j follows from rv
                int o = 0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    gm1[k] = fma(bi * rowsum, d[k], gm1[k]);
                    gm2[k] = fma(bi, rv[k], gm2[k]);
                }
#pragma unroll
                for (int k = 0; k < D; ++k)
#pragma unroll
                    for (int l = k; l < D; ++l) {
                        const double vck = bug ? ila[k] / ilb[k] * rv[k] : rv[k], vcl = bug ? ila[l] / ilb[l] * rv[l] : rv[l];
                        const double g = 0.5 * rowsum * p1[k] * p1[l] + 0.5 * rW[o] + 0.5 * (pc[k] * vcl + pc[l] * vck);
                        gG[o] = fma(bi, g, gG[o]);
                        ++o;
                    }
            }
        }
    }
    double v[NV];
    v[0] = total;
    if (GRAD) {
#pragma unroll
        for (int k = 0; k < D; ++k) { v[1 + k] = gm1[k]; v[1 + D + k] = gm2[k]; }
#pragma unroll
        for (int e = 0; e < NS; ++e) v[1 + 2 * D + e] = gG[e];
    }
    block_sum<NV>(v, s_scr, s_out);
    if (threadIdx.x == 0) {
        const double sfa = A.sf[a], sfb = A.sf[b];
        const double mua = A.out_mean[(size_t)q * A.ds + a], mub = A.out_mean[(size_t)q * A.ds + b];
        const double K = sfa * sfa * sfb * sfb / sqrt(s_det);
        const double cov = K * s_out[0] - mua * mub;
        A.out_cov[((size_t)q * A.ds + a) * A.ds + b] = cov;
        if (!bug) A.out_cov[((size_t)q * A.ds + b) * A.ds + a] = cov;
        if (GRAD) {
            const int ds = A.ds;
            double mi[D], mj[D], G[D * D], T1[D * D];
            for (int k = 0; k < D; ++k) { mi[k] = s_out[1 + k]; mj[k] = s_out[1 + D + k] / ilb[k]; }      // sum w d_i | sum w d_j (d_j = Lb r2_j)
            { int o = 0; for (int k = 0; k < D; ++k) for (int l = k; l < D; ++l) { G[k * D + l] = s_out[1 + 2 * D + o]; G[l * D + k] = s_out[1 + 2 * D + o]; ++o; } }
            const double* dma_u = A.dmean_du + ((size_t)q * ds + a) * D; const double* dmb_u = A.dmean_du + ((size_t)q * ds + b) * D;
            const double* dma_S = A.dmean_dS + ((size_t)q * ds + a) * D * D; const double* dmb_S = A.dmean_dS + ((size_t)q * ds + b) * D * D;
            // d/du
            for (int k = 0; k < D; ++k) {
                double s1 = 0.0;                                          // (P1 m_i + P2 m_j + C m_j + C^T m_i)_k
                for (int l = 0; l < D; ++l) {
                    const double am = Am[k * D + l];
                    const double P1 = ila[k] * am * ila[l] - (k == l ? ila[k] : 0.0), P2 = ilb[k] * am * ilb[l] - (k == l ? ilb[k] : 0.0);
                    const double lpk = bug ? ilb[k] : ila[k], lrl = bug ? ila[l] : ilb[l];      // C_kl = Lpc_k Am_kl Lrc_l
                    const double lpl = bug ? ilb[l] : ila[l], lrk = bug ? ila[k] : ilb[k];      // (C^T)_kl = C_lk = Lpc_l Am_lk Lrc_k
                    s1 += P1 * mi[l] + P2 * mj[l] + lpk * am * lrl * mj[l] + lpl * Am[l * D + k] * lrk * mi[l];
                }
                const double dc = -K * s1 - dma_u[k] * mub - mua * dmb_u[k];
                A.dcov_du[(((size_t)q * ds + a) * ds + b) * D + k] = dc;
                if (!bug) A.dcov_du[(((size_t)q * ds + b) * ds + a) * D + k] = dc;
            }
            // d/dS: K R^-T G R^-1 - 1/2 K W0 sym(L R^-1), symmetrised, minus the mean terms
            for (int m = 0; m < D; ++m) for (int l = 0; l < D; ++l) { double s2 = 0.0; for (int k = 0; k < D; ++k) s2 += s_Ri[k * D + m] * G[k * D + l]; T1[m * D + l] = s2; }
            for (int m = 0; m < D; ++m)
                for (int n = m; n < D; ++n) {
                    double g1 = 0.0, g2 = 0.0;
                    for (int l = 0; l < D; ++l) { g1 += T1[m * D + l] * s_Ri[l * D + n]; g2 += T1[n * D + l] * s_Ri[l * D + m]; }
                    const double lm = ila[m] + ilb[m], ln = ila[n] + ilb[n];
                    const double dK = -0.25 * K * (ln * s_Ri[n * D + m] + lm * s_Ri[m * D + n]);            // sym(-1/2 K L R^-1)
                    const double base = 0.5 * K * (g1 + g2) + dK * s_out[0];
                    const double d_mn = base - dma_S[m * D + n] * mub - mua * dmb_S[m * D + n];
                    const double d_nm = base - dma_S[n * D + m] * mub - mua * dmb_S[n * D + m];
                    double* o1 = A.dcov_dS + (((size_t)q * ds + a) * ds + b) * D * D;
                    o1[m * D + n] = d_mn; o1[n * D + m] = d_nm;
                    if (!bug) { double* o2 = A.dcov_dS + (((size_t)q * ds + b) * ds + a) * D * D; o2[m * D + n] = d_mn; o2[n * D + m] = d_nm; }
                }
        }
    }
}

struct MomPlan { int mode, tiling, tb, waves, nwork, nunits, nm, pps, sps, sbf, gw; size_t off_pp, off_sp, off_part, off_pairs, off_G, total; };

static void plan_mom(const gpmpc_pack* p, int nq, bool grad, bool pair_cov, int ns2, MomPlan* r) {
    const int D = p->D;
    r->mode = pair_cov ? 1 : 0;
    r->tb = nq >= 2 ? 2 : 1;
    const long groups = (nq + r->tb - 1) / r->tb;
    // 256x256 tiles + scalar-broadcast kernel from ~256 workgroups on (640 for training sets of at most 512 points, whose few
    // large tiles are mostly padding): the staged kernel costs 2.5 x as much per pair, so it only wins while the grid is
    // tiny -- full covariance at N = 1024, B = 8: 7.3 -> 3.4 ms per rollout; N = 2048, B = 2: 9.4 -> 4.5 ms; N = 700, B = 16:
    // 3.4 -> 1.7 ms (the threshold used to be 1024 workgroups).
    // (small training sets: 640, and 1536 when the padded size is not a multiple of 256 -- N = 300: the 256-row tiles are mostly padding)
    const long thr = p->tune.sbf_min > 0 ? p->tune.sbf_min : (p->Np >= 640 ? 256 : (p->Np % 256 == 0 ? 640 : 1536));
    r->tiling = (groups * p->wl[r->mode][0].nwork >= thr || p->tune.pair_sb == 1) ? 0 : 1;     // GPMPC_PAIR_SB=1 forces the large tiles
    // scalar-broadcast kernel (pair_kernel_sbf.h) once the grid fills the chip; GPMPC_PAIR_SB=0 (read at pack creation) keeps the staged kernel
    r->sbf = (r->tiling == 0 && D - ns2 <= 2) ? 1 : 0;
    if (p->tune.pair_sb == 0) r->sbf = 0;
    r->gw = gpmpc_sbf_gw(D, ns2);
    const gpmpc_worklist& w = p->wl[r->mode][r->tiling];
    r->waves = w.waves; r->nwork = w.nwork; r->nunits = w.nunits;
    r->nm = gpmpc_num_moments(D, false, grad);
    r->pps = 2 * (D + D * D);
    r->sps = msps_of(D);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    r->off_pp = take(sizeof(double) * (size_t)nq * r->nunits * r->pps);
    r->off_sp = take(sizeof(double) * (size_t)nq * r->nunits * r->sps);
    r->off_part = take(sizeof(double) * (size_t)nq * r->nwork * r->nm);
    r->off_pairs = take(sizeof(int) * 2 * GPMPC_MAX_DS * GPMPC_MAX_DS);
    r->off_G = take(r->sbf ? sizeof(double) * (size_t)nq * r->nunits * p->Np * r->gw : 0);
    r->total = off;
}

extern "C" size_t gpmpc_moment_match_workspace_bytes(const gpmpc_pack* p, int nq) {
    if (!p || nq < 1) return 0;
    size_t best = 0;
    for (int g = 0; g < 2; ++g)
        for (int pc = 0; pc < 2; ++pc) {
            for (int ns2 = p->ds; ns2 <= p->D; ns2 += (p->D > p->ds ? p->D - p->ds : 1)) {
                MomPlan r;
                plan_mom(p, nq, g != 0, pc != 0, ns2, &r);
                if (r.total > best) best = r.total;
            }
        }
    return best;
}

template <int D>
static int run_mom(const gpmpc_pack* p, MomArgs& A, const MomPlan& r, bool grad, hipStream_t s, int* pairs_dev) {
    hipLaunchKernelGGL(k_mom_prep<D>, dim3(A.nq, A.nunits), dim3(256), 0, s, A);
    PairArgs P;
    P.M = p->M; P.XT = p->XT; P.pp = A.pp; P.part = A.part; P.work = p->wl[r.mode][r.tiling].work_dev;
    P.Np = p->Np; P.B = A.nq; P.nunits = r.nunits; P.nwork = r.nwork; P.pps = r.pps; P.nm = r.nm;
    P.jside_off = D + D * D; P.ntri = p->ds; P.ns2 = D; P.colsplit = (r.tiling == 1) ? 1 : 0;
    int rc;
    if (r.sbf) {
        PairSbfArgs Q;
        Q.M = p->M; Q.XT = p->XT; Q.pp = A.pp; Q.G = A.G; Q.part = A.part; Q.work = P.work;
        Q.Np = p->Np; Q.B = A.nq; Q.nunits = r.nunits; Q.nwork = r.nwork; Q.pps = r.pps; Q.nm = r.nm; Q.ntri = p->ds; Q.cu = 1; Q.part0 = nullptr; Q.pstride = 0;
        rc = gpmpc_timed_pair_sbf(D, grad, A.ns2, r.waves, Q, s);
    } else {
        rc = gpmpc_timed_pair(D, false, grad, r.tb, P.colsplit ? 4 : r.waves, P, s);
    }
    if (rc != GPMPC_OK) return rc;
    hipLaunchKernelGGL(k_mom_finish<D>, dim3(A.nq), dim3(256), 0, s, A);
    if (A.out_cov && p->ds > 1 && A.npairs == 0) {          // direct N^2 kernel (also the bug-compatible form)
        int h[2 * GPMPC_MAX_DS * GPMPC_MAX_DS], n = 0;
        const bool bug = (A.flags & GPMPC_COV_BUG_COMPAT) != 0;
        for (int a = 0; a < p->ds; ++a)
            for (int b = 0; b < p->ds; ++b) {
                if (a == b || (!bug && b < a)) continue;   // the bug-compatible form is not symmetric in (a, b)
                h[2 * n] = a; h[2 * n + 1] = b; ++n;
            }
        if (int rcu = gpmpc_upload_small(pairs_dev, h, sizeof(int) * 2 * n, s)) return rcu;      // (h is on this stack frame)
        if (A.dcov_du) hipLaunchKernelGGL((k_cross_cov<D, true>), dim3(A.nq * n), dim3(256), 0, s, A, pairs_dev, n);
        else hipLaunchKernelGGL((k_cross_cov<D, false>), dim3(A.nq * n), dim3(256), 0, s, A, pairs_dev, n);
    }
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}

// ns2: leading input dimensions whose S-derivatives are needed (D for the public entry point; state_dim for the
// full-covariance rollout, whose action block of S is a constant).
int gpmpc_moment_match_ex(const gpmpc_pack* p, int nq, const double* u, const double* S, unsigned flags,
                          double* out_mean, double* out_var, double* out_cov, double* out_l, double* dmean_du,
                          double* dmean_dS, double* dvar_du, double* dvar_dS, double* dcov_du, double* dcov_dS,
                          void* workspace, size_t workspace_bytes, void* stream, int ns2) {
    if (!p || !u || !S || !out_mean || !out_var || !workspace || nq < 1) return GPMPC_E_ARG;
    if (!p->built) return GPMPC_E_STATE;
    const bool grad = (flags & GPMPC_WANT_GRAD) != 0;
    const bool bug = (flags & GPMPC_COV_BUG_COMPAT) != 0;
    if (grad && (!dmean_du || !dmean_dS || !dvar_du || !dvar_dS)) return GPMPC_E_ARG;
    if ((dcov_du == nullptr) != (dcov_dS == nullptr)) return GPMPC_E_ARG;
    // cross-covariances through the pair kernel (with Jacobians) need the cross weight matrices of the pack
    const bool pair_cov = out_cov && !bug && p->fullcov && p->npairs > 0;
    if (dcov_du && (!grad || !out_cov)) return GPMPC_E_STATE;       // (bug-compatible form / no cross weights: the direct kernel k_cross_cov<D, true>)
    if (ns2 < 1 || ns2 > p->D) return GPMPC_E_ARG;
    MomPlan r;
    plan_mom(p, nq, grad, pair_cov, ns2, &r);
    if (workspace_bytes < r.total) return GPMPC_E_WORKSPACE;
    char* ws = (char*)workspace;
    MomArgs A;
    memset(&A, 0, sizeof(A));
    A.XT = p->XT; A.beta = p->beta; A.lam = p->lam; A.sf = p->sf;
    A.N = p->N; A.Np = p->Np; A.ds = p->ds; A.D = p->D;
    A.u = u; A.S = S; A.nq = nq;
    A.pp = (double*)(ws + r.off_pp); A.sp = (double*)(ws + r.off_sp); A.part = (double*)(ws + r.off_part);
    A.pps = r.pps; A.sps = r.sps; A.nwork = r.nwork; A.nunits = r.nunits; A.nm = r.nm; A.grad = grad ? 1 : 0;
    A.ustart = p->wl[r.mode][r.tiling].ustart_dev;
    A.pair_ab = p->pair_ab_dev; A.npairs = pair_cov ? p->npairs : 0;
    A.out_mean = out_mean; A.out_var = out_var; A.out_cov = out_cov; A.out_l = out_l;
    A.dmean_du = dmean_du; A.dmean_dS = dmean_dS; A.dvar_du = dvar_du; A.dvar_dS = dvar_dS;
    A.dcov_du = dcov_du; A.dcov_dS = dcov_dS;
    A.G = r.sbf ? (double*)(ws + r.off_G) : nullptr; A.gw = r.gw; A.ns2 = ns2;
    A.flags = flags;
    hipStream_t s = (hipStream_t)stream;
    int* pairs_dev = (int*)(ws + r.off_pairs);
    switch (p->D) {
        case 1: return run_mom<1>(p, A, r, grad, s, pairs_dev);
        case 2: return run_mom<2>(p, A, r, grad, s, pairs_dev);
        case 3: return run_mom<3>(p, A, r, grad, s, pairs_dev);
        case 4: return run_mom<4>(p, A, r, grad, s, pairs_dev);
        case 5: return run_mom<5>(p, A, r, grad, s, pairs_dev);
        case 6: return run_mom<6>(p, A, r, grad, s, pairs_dev);
        case 7: return run_mom<7>(p, A, r, grad, s, pairs_dev);
        case 8: return run_mom<8>(p, A, r, grad, s, pairs_dev);
    }
    return GPMPC_E_ARG;
}

extern "C" int gpmpc_moment_match(const gpmpc_pack* p, int nq, const double* u, const double* S, unsigned flags,
                                  double* out_mean, double* out_var, double* out_cov, double* out_l, double* dmean_du,
                                  double* dmean_dS, double* dvar_du, double* dvar_dS, double* dcov_du, double* dcov_dS,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!p) return GPMPC_E_ARG;
    if (int rc_dev = gpmpc_check_device(p)) return rc_dev;
    if (!p) return GPMPC_E_ARG;
    return gpmpc_moment_match_ex(p, nq, u, S, flags, out_mean, out_var, out_cov, out_l, dmean_du, dmean_dS, dvar_du, dvar_dS,
                                 dcov_du, dcov_dS, workspace, workspace_bytes, stream, p->D);
}
