// Device-side pieces of the single-step moment matching with a full input covariance (moment.hip) that the full-covariance rollout's
// fused head kernel (fullcov.hip) runs too: the per-(query, unit) set-up + mean sums + column rows ("prep") and the per-unit
// closing algebra ("finish").  Reference: src/tools/uncertainty_prop.py:296-338, :341-399, :402-465.
#pragma once
#include "gpmpc_internal.h"
#include "fast_exp.h"

struct MomArgs {
    const double* XT; const double* beta; const double* lam; const double* sf;
    int N, Np, ds, D;
    const double* u; const double* S; int nq;
    double* pp; double* sp; double* part;
    int pps, sps, nwork, nunits, nm, grad;
    const int* ustart;            // work items of unit u: [ustart[u], ustart[u+1])
    const int* pair_ab; int npairs;   // cross units evaluated by the pair kernel (0: none)
    double* G; int gw, ns2;           // column rows of the scalar-broadcast kernel (pair_kernel_sbf.h), or null
    double* out_mean; double* out_var; double* out_cov; double* out_l;
    double* dmean_du; double* dmean_dS; double* dvar_du; double* dvar_dS; double* dcov_du; double* dcov_dS;
    unsigned flags;
};

// sp layout, variance unit a: 0 c | 1 mu | 2 sf2 | 3 Am[D*D] | 3+DD Cm[D*D] | 3+2DD dmu_du[D] | 3+2DD+D dmu_dS[D*D]
//            cross unit (a,b): 0 c_ab | 1,2 unused | 3 Bab[D*D] | 3+DD Cm[D*D]
// pp layout per unit: rows [cvec(D) | T(D*D)], columns the same D + D*D doubles further
__host__ __device__ static inline int msps_of(int D) { return 3 + 3 * D * D + D; }

// In-place inverse and determinant of a small general matrix (Gauss-Jordan, partial pivoting).
__device__ static double small_inverse(int n, double* a /*[n][n]*/, double* inv /*[n][n]*/) {
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) inv[r * n + c] = (r == c) ? 1.0 : 0.0;
    double det = 1.0;
    for (int k = 0; k < n; ++k) {
        int piv = k; double best = fabs(a[k * n + k]);
        for (int r = k + 1; r < n; ++r) { const double v = fabs(a[r * n + k]); if (v > best) { best = v; piv = r; } }
        if (piv != k) {
            for (int c = 0; c < n; ++c) {
                double t = a[k * n + c]; a[k * n + c] = a[piv * n + c]; a[piv * n + c] = t;
                t = inv[k * n + c]; inv[k * n + c] = inv[piv * n + c]; inv[piv * n + c] = t;
            }
            det = -det;
        }
        const double pv = a[k * n + k];
        det *= pv;
        const double ip = 1.0 / pv;
        for (int c = 0; c < n; ++c) { a[k * n + c] *= ip; inv[k * n + c] *= ip; }
        for (int r = 0; r < n; ++r) {
            if (r == k) continue;
            const double f = a[r * n + k];
            for (int c = 0; c < n; ++c) { a[r * n + c] = fma(-f, a[k * n + c], a[r * n + c]); inv[r * n + c] = fma(-f, inv[k * n + c], inv[r * n + c]); }
        }
    }
    return det;
}
// D x D algebra by the whole workgroup, in LDS (round 4: as single-lane chains of dependent LDS round trips the set-up of a unit cost
// 20 - 30 us -- most of a small-batch step of the full-covariance rollout).  Every thread of the workgroup must call these (barriers).
//
// Inverse and determinant of one or two independent D x D matrices (threads [0, D^2): a0 -> i0; [64, 64 + D^2): a1 -> i1; `which`), Gauss-Jordan
// WITHOUT pivoting: the callers pass sym(S) + diag(lambda) with S a covariance.  a*: in, destroyed; det[0 / 1]: the determinants.
template <int D>
__device__ __forceinline__ void lds_inverse_pair(double* a0, double* i0, double* a1, double* i1, double* det, const int which /* bit 0: a0, bit 1: a1 */) {
    // Each matrix is the work of ONE wave (D^2 <= 64 lanes): its LDS operations execute in program order, so the steps need no
    // workgroup barrier between them -- only the compiler must not move them (wave_barrier) --; one barrier at the end publishes the results.
    const int w = threadIdx.x >> 6, e = threadIdx.x & 63;
    const bool on = e < D * D && ((w == 0 && (which & 1)) || (w == 1 && (which & 2)));
    if (on) {
        double* const a = w == 0 ? a0 : a1;
        double* const iv = w == 0 ? i0 : i1;
        const int r = e / D, c = e - r * D;
        double arc = a[e], irc = (r == c) ? 1.0 : 0.0, dt = 1.0;
        for (int k = 0; k < D; ++k) {
            a[e] = arc; iv[e] = irc;
            __builtin_amdgcn_wave_barrier();
            const double pv = a[k * D + k], ip = 1.0 / pv, f = a[r * D + k];
            const double akc = a[k * D + c] * ip, ikc = iv[k * D + c] * ip;
            dt *= pv;
            if (r == k) { arc = akc; irc = ikc; }
            else { arc = fma(-f, akc, arc); irc = fma(-f, ikc, irc); }
            __builtin_amdgcn_wave_barrier();
        }
        iv[e] = irc;
        if (e == 0) det[w] = dt;
    }
    __syncthreads();
}

// Upper-triangular Cm with Cm^T Cm = scale * sym(Am) (Cholesky by columns; L: D^2 doubles of LDS scratch); wave 0 alone, as above.
template <int D>
__device__ __forceinline__ void lds_cholesky_upper(const double* Am, const double scale, double* L, double* Cm) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        if (tid < D * D) L[tid] = 0.0;
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < D; ++c) {
            if (tid == c) {
                double s = scale * 0.5 * (Am[c * D + c] + Am[c * D + c]);
                for (int l = 0; l < c; ++l) s -= L[c * D + l] * L[c * D + l];
                L[c * D + c] = sqrt(s);
            }
            __builtin_amdgcn_wave_barrier();
            if (tid > c && tid < D) {
                const int r = tid;
                double s = scale * 0.5 * (Am[r * D + c] + Am[c * D + r]);
                for (int l = 0; l < c; ++l) s -= L[r * D + l] * L[c * D + l];
                L[r * D + c] = s / L[c * D + c];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (tid < D * D) { const int r = tid / D, c = tid - r * D; Cm[tid] = (c >= r) ? L[c * D + r] : 0.0; }
    }
    __syncthreads();
}

// Diagnostic build (-DGPMPC_FC_STAMPS): s_memtime stamps of thread 0 of two workgroups of k_fc_head (fullcov.hip) at horizon step 3,
// trajectory 0: the first variance unit (slots 0..) and the last cross unit (slots 32..); tools/fc_stamps.py.
#if defined(GPMPC_FC_STAMPS)
static __device__ unsigned long long g_fc_stamps[64];
#define GPMPC_FST(slot) do { if (sh.stamp >= 0 && threadIdx.x == 0) g_fc_stamps[sh.stamp + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GPMPC_FST(slot) do { } while (0)
#endif

// LDS of one (query, unit) workgroup of the prep phase
template <int D>
struct MomPrepLds {
    static constexpr int NV = 1 + D + D * (D + 1) / 2;
    static constexpr int GWT = (D + 1 + D * (D + 1) / 2 + 1) & ~1;     // widest G row at this D
    static constexpr bool STAGE = GWT <= 28;                          // 256 rows of it within the 64 KB of static LDS
    double u[D], S[D * D], B[D * D];
    double scr[16 * NV], out[NV];
    double tmp[2 * D * D], tmpA[D * D], Cm[D * D], L[D * D], cm, det[2], cv[D];
    double g[STAGE ? 256 * GWT : 1];
    int stamp;          // diagnostic builds: first stamp slot of this workgroup, or -1
};

// Everything of k_mom_prep after the input distribution (sh.u, sh.S) is in LDS: one workgroup of 256 threads per (query q, unit).
template <int D>
// role: bit 0 -- the variance-side set-up (A = (Lambda/2 + S)^-1, its Cholesky factor; the cross units' whole set-up) and the column rows
// of row chunk rs of RS, the first chunk also the set-up records; bit 1 -- the mean side of a variance unit (B = (S + Lambda)^-1, the O(N)
// mean sums, mean and mean Jacobians).  k_mom_prep: one workgroup does both (role 3); k_fc_head (fullcov.hip) splits them over
// workgroups, which all run their part of the (cheap, bit-identical) set-up themselves.
// xpre (optional): the D coordinates of this thread's first row of the chunk, loaded by the caller ahead of time.
__device__ __forceinline__ void mom_prep_body(const MomArgs& A, const int q, const int unit, MomPrepLds<D>& sh, const int rs, const int RS,
                                              const int role, const bool use_pre, const double (&xpre)[D + 1]) {
    const bool need_g = (role & 1) != 0, need_mean = (role & 2) != 0;
    const bool lead = need_g && rs == 0;
    constexpr int NV = MomPrepLds<D>::NV;
    constexpr bool STAGE = MomPrepLds<D>::STAGE;
    double* const s_u = sh.u; double* const s_S = sh.S; double* const s_B = sh.B; double* const s_scr = sh.scr; double* const s_out = sh.out;
    double* const s_tmp = sh.tmp; double* const s_tmpA = sh.tmpA; double* const s_Cm = sh.Cm; double* const s_L = sh.L; double& s_cm = sh.cm;
    double* const s_g = sh.g;
    const int ds = A.ds;
    GPMPC_FST(8);
    if (unit < ds) {
        const int a = unit;
        double* sp = A.sp + ((size_t)q * A.nunits + a) * A.sps;
        double* pp = A.pp + ((size_t)q * A.nunits + a) * A.pps;
        // Set-up of the unit: the mean side (B = (S + Lambda)^-1) and the variance side (A = (Lambda/2 + S)^-1, its Cholesky
        // factor), both inverses at once by the workgroup (see lds_inverse_pair); the results go to global memory afterwards, one
        // element per thread.
        {
            const double* lam = A.lam + a * D;
            if (threadIdx.x < D * D) {
                const int r = threadIdx.x / D, c = threadIdx.x - r * D;
                const double sym = 0.5 * (s_S[r * D + c] + s_S[c * D + r]);
                s_tmpA[threadIdx.x] = sym + (r == c ? lam[r] : 0.0);
                s_tmp[threadIdx.x] = sym + (r == c ? 0.5 * lam[r] : 0.0);
            }
            __syncthreads();
            // B = (S + Lambda)^-1, det(Lambda^-1 S + I) = det(S + Lambda) / det(Lambda);  A likewise with Lambda / 2
            lds_inverse_pair<D>(s_tmpA, s_B, s_tmp, s_tmp + D * D, sh.det, (need_mean ? 1 : 0) | (need_g ? 2 : 0));
            if (threadIdx.x == 0) {
                double detlam = 1.0, dethalf = 1.0;
                for (int k = 0; k < D; ++k) { detlam *= lam[k]; dethalf *= 0.5 * lam[k]; }
                const double sf = A.sf[a];
                if (need_mean) s_cm = sf * sf / sqrt(sh.det[0] / detlam);
                if (lead) { sp[0] = 1.0 / sqrt(sh.det[1] / dethalf); sp[2] = sf * sf; }
            }
            GPMPC_FST(9);
            if (need_g) lds_cholesky_upper<D>(s_tmp + D * D, 0.125, s_L, s_Cm);        // Cholesky A/8 = L L^T, Cm = L^T (upper)
            GPMPC_FST(10);
        }
        if (lead && threadIdx.x < D * D) {
            const int e = threadIdx.x;
            sp[3 + e] = s_tmp[D * D + e];                       // Am
            sp[3 + D * D + e] = s_Cm[e];
            pp[D + e] = s_Cm[e]; pp[D + D * D + D + e] = s_Cm[e];
        }
        if (need_g && threadIdx.x >= 64 && threadIdx.x < 64 + D) {
            const int k = threadIdx.x - 64;
            double s = 0.0;
            for (int l = k; l < D; ++l) s += s_Cm[k * D + l] * s_u[l];
            sh.cv[k] = s;
            if (lead) { pp[k] = s; pp[D + D * D + k] = s; }
        }
        __syncthreads();
        if (need_mean) {
        double u[D];
        const double* Bm = s_B;      // read from LDS in the loop (wave-uniform addresses: broadcast reads): 2 D^2 registers less
#pragma unroll
        for (int k = 0; k < D; ++k) u[k] = s_u[k];
        double v[NV];
#pragma unroll
        for (int m = 0; m < NV; ++m) v[m] = 0.0;
        // ONE workgroup sums all rows of the unit, one wave per SIMD: nothing hides a load or a dependent chain but the loop itself, and
        // between two head kernels the pair kernel has swept the training set out of the L2 (a load costs ~2.5 k cycles).  So: batches
        // of RB rows per thread, the loads of TWO batches in flight (two register sets in alternation), the rows of a batch evaluated
        // in interleaved pairs; the accumulations keep the order of the rows.  N = 2048: 12.4 k -> see profiles/r04/fc_head_stamps.txt.
        constexpr int RB = D <= 5 ? 4 : 2;
        const int BD = blockDim.x;
        double xa[RB][D], ba[RB], xb[RB][D], bb[RB];
        auto fetch = [&](double (&x)[RB][D], double (&b)[RB], const int base) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int i = base + r * BD;
#pragma unroll
                for (int k = 0; k < D; ++k) x[r][k] = i < A.Np ? A.XT[(size_t)k * A.Np + i] : 0.0;
                b[r] = i < A.Np ? A.beta[(size_t)a * A.Np + i] : 0.0;         // past the end: weight zero
            }
        };
        auto rows = [&](const double (&x)[RB][D], const double (&b)[RB], const int base) {
#pragma unroll
            for (int r0 = 0; r0 < RB; r0 += 2) {
                if (base + r0 * BD >= A.Np) continue;               // (wave-uniform: Np is a multiple of 64) nothing but padding left
                double d[2][D], qf[2] = {0.0, 0.0};
                int boff = 0;
                asm volatile("" : "+v"(boff));      // opaque offset: keeps the D^2 reads of B in the loop instead of 2 D^2 hoisted registers
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < D; ++k) d[r][k] = u[k] - x[r0 + r][k];
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double bd0 = 0.0, bd1 = 0.0;
#pragma unroll
                    for (int l = 0; l < D; ++l) { const double bm = Bm[boff + k * D + l]; bd0 = fma(bm, d[0][l], bd0); bd1 = fma(bm, d[1][l], bd1); }
                    qf[0] = fma(bd0, d[0][k], qf[0]); qf[1] = fma(bd1, d[1][k], qf[1]);
                }
                const double ex0 = exp(-0.5 * qf[0]), ex1 = exp(-0.5 * qf[1]);
                const double pr[2] = {b[r0] * ex0, b[r0 + 1] * ex1};
                if (A.out_l) {
                    const int i = base + r0 * BD;
                    if (i < A.N) A.out_l[((size_t)q * ds + a) * A.N + i] = s_cm * ex0;
                    if (i + BD < A.N) A.out_l[((size_t)q * ds + a) * A.N + i + BD] = s_cm * ex1;
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const double p = pr[r];
                    v[0] += p;
                    int o = 1 + D;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const double pd = p * d[r][k];
                        v[1 + k] += pd;
#pragma unroll
                        for (int l = k; l < D; ++l) { v[o] = fma(pd, d[r][l], v[o]); ++o; }
                    }
                }
            }
        };
        fetch(xa, ba, threadIdx.x);
        for (int base = threadIdx.x; base < A.Np; base += 2 * RB * BD) {
            fetch(xb, bb, base + RB * BD);
            rows(xa, ba, base);
            fetch(xa, ba, base + 2 * RB * BD);
            rows(xb, bb, base + RB * BD);
        }
        const double cm = s_cm;
        GPMPC_FST(11);
        block_sum4_rows<NV>(v, s_scr, s_out);
        GPMPC_FST(12);
        {   // mean and its Jacobians: dmu/du = -cm B S1, dmu/dS = -1/2 mu B + 1/2 cm B S2 B -- one element per thread
            const double mu = cm * s_out[0];
            double* dmu_du = sp + 3 + 2 * D * D;
            double* dmu_dS = dmu_du + D;
            double* S2 = s_tmp;                 // LDS scratch of the set-up phase, free by now
            double* BS = s_tmp + D * D;
            const int e = threadIdx.x, r = e / D, c = e - r * D;
            if (e == 0) sp[1] = mu;
            if (e < D * D) {
                const int k = r < c ? r : c, l = r < c ? c : r;
                S2[e] = s_out[1 + D + k * D - k * (k - 1) / 2 + (l - k)];
            }
            if (e >= 64 && e < 64 + D) {
                const int k = e - 64;
                double s = 0.0;
                for (int l = 0; l < D; ++l) s += s_B[k * D + l] * s_out[1 + l];
                dmu_du[k] = -cm * s;
            }
            __syncthreads();
            if (e < D * D) { double s = 0.0; for (int l = 0; l < D; ++l) s += s_B[r * D + l] * S2[l * D + c]; BS[e] = s; }
            __syncthreads();
            if (e < D * D) {
                double s = 0.0;
                for (int l = 0; l < D; ++l) s += BS[r * D + l] * s_B[l * D + c];
                dmu_dS[e] = -0.5 * mu * s_B[e] + 0.5 * cm * s;
            }
        }
        }       // need_mean
        __syncthreads();
    }
    // cross-covariance units (a < b): Gaussian-product form of covariance_prop_torch (:402-465)
    //   Lab = (La^-1 + Lb^-1)^-1, w_a = Lab La^-1, w_b = Lab Lb^-1, Bab = (S + Lab)^-1, c = det(Lab^-1 S + I)^-1/2,
    //   Cm^T Cm = Bab / 2, rows p_i = Cm (w_a o (u - x_i)), columns q_j = Cm (w_b o (u - x_j)).
    if (unit >= ds && !need_g) return;                         // (cross units have no mean side)
    if (unit >= ds) {
        const int pr = unit - ds;
        const int a = A.pair_ab[2 * pr], b = A.pair_ab[2 * pr + 1];
        const double* la = A.lam + a * D; const double* lb = A.lam + b * D;
        double* sp = A.sp + ((size_t)q * A.nunits + ds + pr) * A.sps;
        double* pp = A.pp + ((size_t)q * A.nunits + ds + pr) * A.pps;
        double* Mt = s_tmp; double* Bab = s_tmp + D * D; double* Cm = s_Cm;
        const int e = threadIdx.x, r = e / D, c = e - r * D;
        if (e < D * D) {
            const double lab = la[r] * lb[r] / (la[r] + lb[r]);
            Mt[e] = 0.5 * (s_S[r * D + c] + s_S[c * D + r]) + (r == c ? lab : 0.0);
        }
        __syncthreads();
        lds_inverse_pair<D>(Mt, Bab, nullptr, nullptr, sh.det, 1);
        if (lead && e == 0) {
            double detlab = 1.0;
            for (int k = 0; k < D; ++k) detlab *= la[k] * lb[k] / (la[k] + lb[k]);
            sp[0] = sqrt(detlab / sh.det[0]);
            sp[1] = 0.0; sp[2] = 0.0;
        }
        if (lead && e < D * D) sp[3 + e] = Bab[e];
        lds_cholesky_upper<D>(Bab, 0.5, s_tmpA, Cm);            // Cm^T Cm = Bab / 2
        double* pr_r = pp; double* pr_c = pp + D + D * D;
        double* Tr = Mt; double* Tc = s_tmpA;                    // (free by now)
        if (e < D * D) {
            const double wa = lb[c] / (la[c] + lb[c]), wb = la[c] / (la[c] + lb[c]);
            const double tr_ = Cm[e] * wa, tc_ = Cm[e] * wb;
            if (lead) { sp[3 + D * D + e] = Cm[e]; pr_r[D + e] = tr_; pr_c[D + e] = tc_; }
            Tr[e] = tr_; Tc[e] = tc_;
        }
        __syncthreads();
        if (e < D) {
            double sr = 0.0;
            for (int l = 0; l < D; ++l) sr += Tr[e * D + l] * s_u[l];
            if (lead) pr_r[e] = sr;
        } else if (e >= 64 && e < 64 + D) {
            const int k = e - 64;
            double sc = 0.0;
            for (int l = 0; l < D; ++l) sc += Tc[k * D + l] * s_u[l];
            sh.cv[k] = sc;
            if (lead) pr_c[k] = sc;
        }
    }
    GPMPC_FST(13);
    if (!A.G || !need_g) return;
    // Column rows of every unit for the scalar-broadcast pair kernel: [q_j (D) | |q_j|^2 | q_jk q_jl (k <= l < ns2) | pad]
    // with q_j = cvec_c - T_c x_j, the unit's COLUMN-side transform written above by this workgroup.
    __syncthreads();
    {
        const int u = unit;
        double* Gu = A.G + ((size_t)q * A.nunits + u) * A.Np * A.gw;
        double cv[D];
        const double* T = unit < ds ? s_Cm : s_tmpA;    // the column-side transform (LDS copies of what went to pp), read from LDS in the loop like B above
#pragma unroll
        for (int k = 0; k < D; ++k) cv[k] = sh.cv[k];
        const int rchunk = (((A.Np + 255) / 256 + RS - 1) / RS) * 256;      // rows of this workgroup: [rs * rchunk, + rchunk)
        const int row0 = rs * rchunk, row1 = row0 + rchunk < A.Np ? row0 + rchunk : A.Np;
        // The rows of 256 consecutive points are contiguous in G: staged through LDS and written lane-contiguously (a lane
        // writing its own 8 gw-byte row stores 8 bytes into 64 different cache lines per instruction: C5 376 us per launch
        // for 671 MB).  D >= 7 (rows of up to 46 doubles) keeps the direct stores: the staging buffer would not fit.
        for (int i0 = row0; i0 < row1; i0 += blockDim.x) {
            const int i = i0 + threadIdx.x;
            if (i < row1) {
                double x[D], qv[D], qq = 0.0;
                int toff = 0;
                asm volatile("" : "+v"(toff));      // as for B above: T stays in LDS
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = (use_pre && need_g && i0 == row0) ? xpre[k] : A.XT[(size_t)k * A.Np + i];
                double* g = STAGE ? s_g + (size_t)threadIdx.x * A.gw : Gu + (size_t)i * A.gw;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    double sacc = cv[k];
#pragma unroll
                    for (int l = k; l < D; ++l) sacc = fma(-T[toff + k * D + l], x[l], sacc);
                    qv[k] = sacc; g[k] = sacc;
                    qq = fma(sacc, sacc, qq);
                }
                g[D] = GPMPC_EXP_NEG_INV_C * qq;                  // pre-scaled for gpmpc_exp_neg_scaled (fast_exp.h)
                int o = D + 1;
#pragma unroll
                for (int k = 0; k < D; ++k)
#pragma unroll
                    for (int l = k; l < D; ++l)
                        if (k < A.ns2 && l < A.ns2) { g[o] = qv[k] * qv[l]; ++o; }
                for (; o < A.gw; ++o) g[o] = 0.0;
            }
            if (STAGE) {
                __syncthreads();
                const int rows = (row1 - i0 < (int)blockDim.x) ? row1 - i0 : (int)blockDim.x;
                double* dst = Gu + (size_t)i0 * A.gw;
#pragma unroll 4
                for (int e = threadIdx.x; e < rows * A.gw; e += blockDim.x) dst[e] = s_g[e];
                __syncthreads();
            }
        }
    }
    GPMPC_FST(14);
}

// Closing algebra of ONE unit u of query q (one thread): z = the unit's nm moment sums, spbase = the set-up record array
// ([nq][nunits][sps]) of the step being closed.  moments: also write mean / variance / covariance (the Jacobians always, with A.grad).
template <int D>
__device__ __forceinline__ void mom_finish_unit(const MomArgs& A, const double* __restrict__ spbase, const int q, const int u,
                                                const double* z, const bool moments) {
    const int ds = A.ds, nunits = A.nunits;
    const double* sp = spbase + ((size_t)q * nunits + u) * A.sps;
    const double c = sp[0];
    const double* Bm = sp + 3;                 // Am (variance unit) or Bab (cross unit)
    const double* Cm = sp + 3 + D * D;
    double Z2[D * D], CtZ1[D], CZC[D * D];
    if (A.grad) {
        int o = 1 + D;
        for (int k = 0; k < D; ++k) for (int l = k; l < D; ++l) { Z2[k * D + l] = Z2[l * D + k] = z[o]; ++o; }
        for (int k = 0; k < D; ++k) {
            double s = 0.0;                                   // (Cm^T Z1)_k
            for (int l = 0; l <= k; ++l) s += Cm[l * D + k] * z[1 + l];
            CtZ1[k] = s;
        }
        double ZC[D * D];                                     // Z2 Cm
        for (int r = 0; r < D; ++r) for (int cc = 0; cc < D; ++cc) { double s = 0.0; for (int l = 0; l <= cc; ++l) s += Z2[r * D + l] * Cm[l * D + cc]; ZC[r * D + cc] = s; }
        for (int r = 0; r < D; ++r) for (int cc = 0; cc < D; ++cc) { double s = 0.0; for (int l = 0; l <= r; ++l) s += Cm[l * D + r] * ZC[l * D + cc]; CZC[r * D + cc] = s; }
    }
    if (u < ds) {                                             // variance unit
        const int a = u;
        const double mu = sp[1], sf2 = sp[2];
        const double T = c * z[0];
        const double var = sf2 - T - mu * mu;
        if (moments) {
            A.out_mean[(size_t)q * ds + a] = mu;
            A.out_var[(size_t)q * ds + a] = var;
            if (A.out_cov) A.out_cov[((size_t)q * ds + a) * ds + a] = var;
        }
        if (!A.grad) return;
        const double* dmu_du = sp + 3 + 2 * D * D;
        const double* dmu_dS = dmu_du + D;
        for (int k = 0; k < D; ++k) {
            const double dT_du = -4.0 * c * CtZ1[k];
            const double dv = -dT_du - 2.0 * mu * dmu_du[k];
            A.dmean_du[((size_t)q * ds + a) * D + k] = dmu_du[k];
            if (A.dvar_du) A.dvar_du[((size_t)q * ds + a) * D + k] = dv;
            if (A.dcov_du) A.dcov_du[(((size_t)q * ds + a) * ds + a) * D + k] = dv;
        }
        for (int e = 0; e < D * D; ++e) {
            const double dT_dS = -0.5 * T * Bm[e] + 8.0 * c * CZC[e];
            const double dv = -dT_dS - 2.0 * mu * dmu_dS[e];
            A.dmean_dS[((size_t)q * ds + a) * D * D + e] = dmu_dS[e];
            if (A.dvar_dS) A.dvar_dS[((size_t)q * ds + a) * D * D + e] = dv;
            if (A.dcov_dS) A.dcov_dS[(((size_t)q * ds + a) * ds + a) * D * D + e] = dv;
        }
    } else {                                                  // cross unit (a, b): F = c Z0, Cov = F - mu_a mu_b
        const int pr = u - ds, a = A.pair_ab[2 * pr], b = A.pair_ab[2 * pr + 1];
        const double* spa = spbase + ((size_t)q * nunits + a) * A.sps;
        const double* spb = spbase + ((size_t)q * nunits + b) * A.sps;
        const double mua = spa[1], mub = spb[1];
        const double F = c * z[0];
        const double cov = F - mua * mub;
        if (moments) {
            A.out_cov[((size_t)q * ds + a) * ds + b] = cov;
            A.out_cov[((size_t)q * ds + b) * ds + a] = cov;
        }
        if (!A.grad || !A.dcov_du) return;
        const double* da_du = spa + 3 + 2 * D * D; const double* da_dS = da_du + D;
        const double* db_du = spb + 3 + 2 * D * D; const double* db_dS = db_du + D;
        for (int k = 0; k < D; ++k) {
            const double d = -2.0 * c * CtZ1[k] - mub * da_du[k] - mua * db_du[k];
            A.dcov_du[(((size_t)q * ds + a) * ds + b) * D + k] = d;
            A.dcov_du[(((size_t)q * ds + b) * ds + a) * D + k] = d;
        }
        for (int e = 0; e < D * D; ++e) {
            const double d = -0.5 * F * Bm[e] + 2.0 * c * CZC[e] - mub * da_dS[e] - mua * db_dS[e];
            A.dcov_dS[(((size_t)q * ds + a) * ds + b) * D * D + e] = d;
            A.dcov_dS[(((size_t)q * ds + b) * ds + a) * D * D + e] = d;
        }
    }
}

// The same by the whole workgroup for ONE unit (k_fc_head): the D x D products one element per thread, LDS scratch from the prep
// phase's (sh.tmp, sh.tmpA, sh.L: free until mom_prep_body).  Jacobians only (A.grad set); every thread of the workgroup calls it.
template <int D>
__device__ __forceinline__ void mom_finish_unit_wg(const MomArgs& A, const double* __restrict__ spbase, const int q, const int u,
                                                   const double* z, MomPrepLds<D>& sh) {
    const int ds = A.ds, nunits = A.nunits, e = threadIdx.x, r = e / D, cc = e - r * D;
    const double* sp = spbase + ((size_t)q * nunits + u) * A.sps;
    const double* Bm = sp + 3;
    double* CmL = sh.tmp; double* Z2 = sh.tmp + D * D; double* ZC = sh.tmpA; double* CtZ1 = sh.L;
    double bme = 0.0;
    if (e < D * D) {
        CmL[e] = sp[3 + D * D + e];
        bme = Bm[e];
        const int k = r < cc ? r : cc, l = r < cc ? cc : r;
        Z2[e] = z[1 + D + k * D - k * (k - 1) / 2 + (l - k)];
    }
    const double c = sp[0];
    __syncthreads();
    if (e < D * D) { double s = 0.0; for (int l = 0; l <= cc; ++l) s += Z2[r * D + l] * CmL[l * D + cc]; ZC[e] = s; }      // Z2 Cm
    if (e >= 64 && e < 64 + D) {
        const int k = e - 64;
        double s = 0.0;                                       // (Cm^T Z1)_k
        for (int l = 0; l <= k; ++l) s += CmL[l * D + k] * z[1 + l];
        CtZ1[k] = s;
    }
    __syncthreads();
    double czc = 0.0;
    if (e < D * D) for (int l = 0; l <= r; ++l) czc += CmL[l * D + r] * ZC[l * D + cc];                                        // Cm^T Z2 Cm
    if (u < ds) {
        const int a = u;
        const double mu = sp[1];
        const double T = c * z[0];
        const double* dmu_du = sp + 3 + 2 * D * D;
        const double* dmu_dS = dmu_du + D;
        if (e < D) {
            const double dT_du = -4.0 * c * CtZ1[e];
            const double dv = -dT_du - 2.0 * mu * dmu_du[e];
            A.dmean_du[((size_t)q * ds + a) * D + e] = dmu_du[e];
            if (A.dcov_du) A.dcov_du[(((size_t)q * ds + a) * ds + a) * D + e] = dv;
        }
        if (e < D * D) {
            const double dT_dS = -0.5 * T * bme + 8.0 * c * czc;
            const double dv = -dT_dS - 2.0 * mu * dmu_dS[e];
            A.dmean_dS[((size_t)q * ds + a) * D * D + e] = dmu_dS[e];
            if (A.dcov_dS) A.dcov_dS[(((size_t)q * ds + a) * ds + a) * D * D + e] = dv;
        }
    } else if (A.dcov_du) {
        const int pr = u - ds, a = A.pair_ab[2 * pr], b = A.pair_ab[2 * pr + 1];
        const double* spa = spbase + ((size_t)q * nunits + a) * A.sps;
        const double* spb = spbase + ((size_t)q * nunits + b) * A.sps;
        const double mua = spa[1], mub = spb[1];
        const double F = c * z[0];
        const double* da_du = spa + 3 + 2 * D * D; const double* da_dS = da_du + D;
        const double* db_du = spb + 3 + 2 * D * D; const double* db_dS = db_du + D;
        if (e < D) {
            const double d = -2.0 * c * CtZ1[e] - mub * da_du[e] - mua * db_du[e];
            A.dcov_du[(((size_t)q * ds + a) * ds + b) * D + e] = d;
            A.dcov_du[(((size_t)q * ds + b) * ds + a) * D + e] = d;
        }
        if (e < D * D) {
            const double d = -0.5 * F * bme + 2.0 * c * czc - mub * da_dS[e] - mua * db_dS[e];
            A.dcov_dS[(((size_t)q * ds + a) * ds + b) * D * D + e] = d;
            A.dcov_dS[(((size_t)q * ds + b) * ds + a) * D * D + e] = d;
        }
    }
    __syncthreads();                 // the scratch goes back to the prep phase
}
