/* TEST INFRASTRUCTURE ONLY -- plain-C / OpenMP CPU port of the FULL-covariance rollout (BASELINE config 5), used as the
 * full-size checker of gpmpc_rollout_fullcov (the torch extension oracle cannot afford N = 2048 over a whole horizon).
 * Never linked into the product.  Pinned on the CPU to oracle/gpmpc_oracle.py::forward_propagate_fullcov /
 * objective_and_gradient_fullcov (themselves composed of functions pinned to the reference's golden vectors) by
 * tests/test_oracle_golden.py::test_cport_fullcov_matches_torch_oracle.
 *
 * Restates, per trajectory, in the DIRECT forms of the reference (no folded weights, no expanded exponents -- a
 * different evaluation order from the HIP kernels on purpose):
 *   Dynamics.forward_propagate_torch      src/dynamics.py:126-191, with the off-diagonal covariances it leaves as a TODO (:184)
 *   mean_prop_torch                       src/tools/uncertainty_prop.py:296-338   (full S: B = (S + Lambda)^-1)
 *   variance_prop_torch                   src/tools/uncertainty_prop.py:341-399   (elementwise O(N^2) trace)
 *   covariance_prop (consistent form)     src/tools/uncertainty_prop.py:187-237 / :402-465 with z1^T A z2
 *   cost_torch                            src/mpc.py:156-200 (full Sigma)
 *
 * The file is compiled twice: as is (double) and with -DCPLX (double complex).  The complex build makes the
 * complex-step derivative available: with U + i h d as input, Im(cost) / h is the directional derivative d . dcost/dU to
 * machine precision (no subtraction, h = 1e-20) -- an independent check of the analytic adjoint of the HIP path that
 * needs no Jacobian code here.  Pivoting compares moduli; every other operation is analytic.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <complex.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 8

#ifdef CPLX
typedef double complex num;
typedef long double complex lnum;
#define N_EXP cexp
#define N_SQRT csqrt
#define N_LOG clog
#define N_ABS cabs
#define FN(name) name##_cplx
#else
typedef double num;
typedef long double lnum;
#define N_EXP exp
#define N_SQRT sqrt
#define N_LOG log
#define N_ABS fabs
#define FN(name) name##_real
#endif

/* In-place Gauss-Jordan on the n x (n + m) augmented matrix a (row stride ld): returns det of the left block, leaves
 * [I | left^-1 right]. */
static num FN(gauss_jordan)(int n, int m, num* a, int ld) {
    num det = 1.0;
    for (int k = 0; k < n; ++k) {
        int piv = k; double best = N_ABS(a[k * ld + k]);
        for (int r = k + 1; r < n; ++r) if (N_ABS(a[r * ld + k]) > best) { best = N_ABS(a[r * ld + k]); piv = r; }
        if (piv != k) { for (int c = 0; c < n + m; ++c) { num t = a[k * ld + c]; a[k * ld + c] = a[piv * ld + c]; a[piv * ld + c] = t; } det = -det; }
        const num pv = a[k * ld + k]; det *= pv;
        for (int c = 0; c < n + m; ++c) a[k * ld + c] /= pv;
        for (int r = 0; r < n; ++r) if (r != k) { const num f = a[r * ld + k]; for (int c = 0; c < n + m; ++c) a[r * ld + c] -= f * a[k * ld + c]; }
    }
    return det;
}

/* inv (n x n, row-major, stride n) and determinant of Min */
static num FN(inv_det)(int n, const num* Min, num* inv) {
    num a[MAXD * 2 * MAXD];
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) { a[r * 2 * n + c] = Min[r * n + c]; a[r * 2 * n + n + c] = (r == c) ? 1.0 : 0.0; }
    const num det = FN(gauss_jordan)(n, n, a, 2 * n);
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) inv[r * n + c] = a[r * 2 * n + n + c];
    return det;
}

typedef struct {
    int N, ds, da, D, H;
    const double *X, *lam, *sf;
    const double* beta;      /* [ds][N] */
    const double* Wv;        /* [ds][N][N]  sym(Kinv - beta beta^T) o exp(-1/4 d2) sf^4 */
    double gamma; const double *Q, *R, *xref, *uref;
} ctx_t;

/* One moment-matching step for the Gaussian input N(u, S), S a full D x D covariance: mean mt [ds] and covariance
 * ct [ds][ds] of the ds GP outputs.  V, AV, Z2: N x D scratch; gq: 2N scratch. */
static void FN(step)(const ctx_t* c, const num* u, const num* S, num* mt, num* ct, num* V, num* AV, num* gq, num* Z2) {
    const int N = c->N, ds = c->ds, D = c->D;
    num* rowsum = (num*)malloc(sizeof(num) * (size_t)N);
    for (int i = 0; i < N; ++i) for (int k = 0; k < D; ++k) V[(size_t)i * D + k] = u[k] - c->X[(size_t)i * D + k];
    num mu[MAXD];
    {
        for (int a = 0; a < ds; ++a) {
            const double* la = c->lam + a * D; const double sf2 = c->sf[a] * c->sf[a];
            num Mx[MAXD * MAXD], Bm[MAXD * MAXD], Am[MAXD * MAXD], tmp[MAXD * MAXD];
            /* mean_prop_torch :329-338 */
            for (int r = 0; r < D; ++r) for (int q = 0; q < D; ++q) { Mx[r * D + q] = S[r * D + q] + (r == q ? la[r] : 0.0); tmp[r * D + q] = S[r * D + q] / la[r] + (r == q ? 1.0 : 0.0); }
            FN(inv_det)(D, Mx, Bm);
            num dummy[MAXD * MAXD];
            const num detm = FN(inv_det)(D, tmp, dummy);
            num S0 = 0.0;
            for (int i = 0; i < N; ++i) {
                const num* v = V + (size_t)i * D; num quad = 0.0;
                for (int r = 0; r < D; ++r) { num s = 0.0; for (int q = 0; q < D; ++q) s += Bm[r * D + q] * v[q]; quad += v[r] * s; }
                S0 += c->beta[(size_t)a * N + i] * N_EXP(-0.5 * quad);
            }
            mu[a] = sf2 / N_SQRT(detm) * S0;
            /* variance_prop_torch :374-399, trace as the elementwise sum */
            for (int r = 0; r < D; ++r) for (int q = 0; q < D; ++q) { Mx[r * D + q] = S[r * D + q] + (r == q ? 0.5 * la[r] : 0.0); tmp[r * D + q] = 2.0 * S[r * D + q] / la[r] + (r == q ? 1.0 : 0.0); }
            FN(inv_det)(D, Mx, Am);
            const num det2 = FN(inv_det)(D, tmp, dummy);
            for (int i = 0; i < N; ++i) {
                const num* v = V + (size_t)i * D; num g = 0.0;
                for (int r = 0; r < D; ++r) { num s = 0.0; for (int q = 0; q < D; ++q) s += Am[r * D + q] * v[q]; AV[(size_t)i * D + r] = s; g += v[r] * s; }
                gq[i] = g;
            }
            const double* W = c->Wv + (size_t)a * N * N;
            num T = 0.0;
            /* row sums in parallel, then one fixed-order reduction in extended precision (independent of the thread count) */
#pragma omp parallel for schedule(dynamic, 8)
            for (int i = 0; i < N; ++i) {
                const num* avi = AV + (size_t)i * D; num row = 0.0;
                for (int j = i; j < N; ++j) {
                    const num* vj = V + (size_t)j * D; num G = 0.0;
                    for (int r = 0; r < D; ++r) G += vj[r] * avi[r];
                    row += (i == j ? 1.0 : 2.0) * W[(size_t)i * N + j] * N_EXP(-0.125 * (gq[i] + 2.0 * G + gq[j]));
                }
                rowsum[i] = row;
            }
            { lnum e = 0.0; for (int i = 0; i < N; ++i) e += rowsum[i]; T = (num)e; }
            mt[a] = mu[a];
            ct[a * ds + a] = sf2 - T / N_SQRT(det2) - mu[a] * mu[a];
        }
        /* cross-covariances, consistent form of covariance_prop (:187-237; :402-465 with z1^T A z2) */
        for (int a = 0; a < ds; ++a)
            for (int b = a + 1; b < ds; ++b) {
                const double* la = c->lam + a * D; const double* lb = c->lam + b * D;
                num Rm[MAXD * MAXD], Ri[MAXD * MAXD], Am[MAXD * MAXD];
                for (int r = 0; r < D; ++r) for (int q = 0; q < D; ++q) Rm[r * D + q] = S[r * D + q] * (1.0 / la[q] + 1.0 / lb[q]) + (r == q ? 1.0 : 0.0);
                const num detR = FN(inv_det)(D, Rm, Ri);
                for (int r = 0; r < D; ++r) for (int q = 0; q < D; ++q) { num s = 0.0; for (int l = 0; l < D; ++l) s += Ri[r * D + l] * S[l * D + q]; Am[r * D + q] = s; }
                /* per point: z1_i = La^-1 (x_i - u), w_i = Am^T z1_i, e1_i = -1/2 k1_i + 1/2 z1^T Am z1; likewise z2_j, e2_j */
                for (int i = 0; i < N; ++i) {
                    const num* v = V + (size_t)i * D; num z1[MAXD], z2[MAXD], k1 = 0.0, k2 = 0.0, q1 = 0.0, q2 = 0.0;
                    for (int r = 0; r < D; ++r) { z1[r] = -v[r] / la[r]; z2[r] = -v[r] / lb[r]; k1 += v[r] * v[r] / la[r]; k2 += v[r] * v[r] / lb[r]; }
                    for (int q = 0; q < D; ++q) { num s = 0.0; for (int r = 0; r < D; ++r) s += z1[r] * Am[r * D + q]; AV[(size_t)i * D + q] = s; }
                    for (int r = 0; r < D; ++r) { num s1 = 0.0, s2 = 0.0; for (int q = 0; q < D; ++q) { s1 += Am[r * D + q] * z1[q]; s2 += Am[r * D + q] * z2[q]; } q1 += z1[r] * s1; q2 += z2[r] * s2; Z2[(size_t)i * D + r] = z2[r]; }
                    gq[2 * i] = -0.5 * k1 + 0.5 * q1; gq[2 * i + 1] = -0.5 * k2 + 0.5 * q2;
                }
                const double* ba = c->beta + (size_t)a * N; const double* bb = c->beta + (size_t)b * N;
                num Qs = 0.0;
#pragma omp parallel for schedule(static)
                for (int i = 0; i < N; ++i) {
                    const num* wi = AV + (size_t)i * D; num row = 0.0;
                    for (int j = 0; j < N; ++j) {
                        const num* zj = Z2 + (size_t)j * D; num cr = 0.0;
                        for (int r = 0; r < D; ++r) cr += wi[r] * zj[r];
                        row += bb[j] * N_EXP(gq[2 * i] + gq[2 * j + 1] + cr);
                    }
                    rowsum[i] = ba[i] * row;
                }
                { lnum e = 0.0; for (int i = 0; i < N; ++i) e += rowsum[i]; Qs = (num)e; }
                const double sfab = c->sf[a] * c->sf[a] * c->sf[b] * c->sf[b];
                const num cv = sfab / N_SQRT(detR) * Qs - mu[a] * mu[b];
                ct[a * ds + b] = cv; ct[b * ds + a] = cv;
            }
    }
    free(rowsum);
}

/* one trajectory: means [H+1][ds], covs [H+1][ds][ds]; returns the cost (src/mpc.py:182-189) */
static num FN(rollout_one)(const ctx_t* c, const double* x0, const num* U, num* means, num* covs) {
    const int N = c->N, ds = c->ds, da = c->da, D = c->D, H = c->H;
    for (int k = 0; k < ds; ++k) { means[k] = x0[k]; for (int l = 0; l < ds; ++l) covs[k * ds + l] = (k == l) ? 1e-3 : 0.0; }   /* dynamics.py:145-148 */
    num* V = (num*)malloc(sizeof(num) * (size_t)N * D);       /* v_i = u - x_i */
    num* AV = (num*)malloc(sizeof(num) * (size_t)N * D);
    num* gq = (num*)malloc(sizeof(num) * (size_t)N * 2);
    num* Z2 = (num*)malloc(sizeof(num) * (size_t)N * D);
    for (int t = 1; t <= H; ++t) {
        num u[MAXD], S[MAXD * MAXD];
        for (int k = 0; k < D * D; ++k) S[k] = 0.0;
        for (int k = 0; k < ds; ++k) { u[k] = means[(t - 1) * ds + k]; for (int l = 0; l < ds; ++l) S[k * D + l] = covs[((size_t)(t - 1) * ds + k) * ds + l]; }
        for (int k = 0; k < da; ++k) { u[ds + k] = U[(t - 1) * da + k]; S[(ds + k) * D + ds + k] = (double)1e-3f; }              /* dynamics.py:154-163 */
        FN(step)(c, u, S, means + (size_t)t * ds, covs + (size_t)t * ds * ds, V, AV, gq, Z2);
    }
    free(V); free(AV); free(gq); free(Z2);
    /* cost_torch, src/mpc.py:179-189 */
    num total = 0.0;
    const double g = c->gamma;
    for (int i = 0; i <= H; ++i) {
        const num* m = means + (size_t)i * ds; const num* Sg = covs + (size_t)i * ds * ds;
        num a[MAXD * 2 * MAXD], e[MAXD];
        for (int k = 0; k < ds; ++k) e[k] = m[k] - c->xref[k];
        if (g == 0.0) {
            for (int r = 0; r < ds; ++r) { num qe = 0.0; for (int q = 0; q < ds; ++q) { qe += c->Q[r * ds + q] * e[q]; total += c->Q[r * ds + q] * Sg[q * ds + r]; } total += e[r] * qe; }
            continue;
        }
        for (int r = 0; r < ds; ++r) for (int q = 0; q < ds; ++q) {
            num s = 0.0; for (int l = 0; l < ds; ++l) s += c->Q[r * ds + l] * Sg[l * ds + q];
            a[r * 2 * ds + q] = (r == q ? 1.0 : 0.0) + g * s; a[r * 2 * ds + ds + q] = c->Q[r * ds + q];
        }
        const num det = FN(gauss_jordan)(ds, ds, a, 2 * ds);          /* (Q^-1 + g Sig)^-1 = (I + g Q Sig)^-1 Q */
        num quad = 0.0;
        for (int r = 0; r < ds; ++r) { num s = 0.0; for (int q = 0; q < ds; ++q) s += a[r * 2 * ds + ds + q] * e[q]; quad += e[r] * s; }
        total += N_LOG(det) / g + quad;
    }
    for (int j = 0; j < H; ++j)
        for (int k = 0; k < da; ++k) { num rd = 0.0; for (int l = 0; l < da; ++l) rd += c->R[k * da + l] * (U[j * da + l] - c->uref[l]); total += (U[j * da + k] - c->uref[k]) * rd; }
    return total;
}

#ifndef CPLX
static void rollout_step_real_entry(const ctx_t* c, const double* u, const double* S, double* mean, double* cov, double* scr) {
    const size_t nd = (size_t)c->N * c->D;
    step_real(c, u, S, mean, cov, scr, scr + nd, scr + 2 * nd, scr + 2 * nd + 2 * (size_t)c->N);
}
num rollout_one_cplx_entry(const ctx_t* c, const double* x0, const double* U, const double* dir, double h, double* dcost);

/* per-data constants: beta_a = Kinv_a y_a, Wv_a = sym(Kinv_a - beta beta^T) o exp(-1/4 d^2_Lambda) sf^4 */
static int build_constants(int N, int ds, int D, const double* X, const double* Kinv, const double* Y, const double* lam,
                           const double* sf, double** beta_out, double** Wv_out) {
    double* beta = (double*)malloc(sizeof(double) * (size_t)ds * N);
    double* Wv = (double*)malloc(sizeof(double) * (size_t)ds * N * N);
    if (!beta || !Wv) return -2;
    for (int a = 0; a < ds; ++a) {
        const double* K = Kinv + (size_t)a * N * N;
#pragma omp parallel for
        for (int i = 0; i < N; ++i) { double s = 0.0; for (int j = 0; j < N; ++j) s += K[(size_t)i * N + j] * Y[(size_t)j * ds + a]; beta[(size_t)a * N + i] = s; }
        const double sf4 = sf[a] * sf[a] * sf[a] * sf[a];
#pragma omp parallel for
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double d2 = 0.0;
                for (int k = 0; k < D; ++k) { const double d = X[(size_t)i * D + k] - X[(size_t)j * D + k]; d2 += d * d / lam[a * D + k]; }
                Wv[((size_t)a * N + i) * N + j] = (0.5 * (K[(size_t)i * N + j] + K[(size_t)j * N + i]) - beta[(size_t)a * N + i] * beta[(size_t)a * N + j]) * sf4 * exp(-0.25 * d2);
            }
    }
    *beta_out = beta; *Wv_out = Wv;
    return 0;
}

/* Single step (unit parity with mean_prop_torch / variance_prop_torch / covariance_prop at a given full S):
 * Y [N][ds] column a = targets of GP a; u [D]; S [D][D]; mean [ds]; cov [ds][ds]. */
int gpmpc_cpu_moment_match_fullcov(int N, int ds, int D, const double* X, const double* Kinv, const double* Y,
                                   const double* lam, const double* sf, const double* u, const double* S, double* mean,
                                   double* cov, int nthreads) {
    if (D > MAXD || ds < 1 || ds > D || N < 1) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double *beta, *Wv;
    if (build_constants(N, ds, D, X, Kinv, Y, lam, sf, &beta, &Wv)) return -2;
    ctx_t c = {N, ds, D - ds, D, 1, X, lam, sf, beta, Wv, 0.0, 0, 0, 0, 0};
    double* scr = (double*)malloc(sizeof(double) * (size_t)N * (3 * D + 2));
    rollout_step_real_entry(&c, u, S, mean, cov, scr);
    free(scr); free(beta); free(Wv);
    return 0;
}

/* B trajectories: means [B][H+1][ds], covs [B][H+1][ds][ds], cost [B]; ndir directional derivatives of the cost per
 * trajectory along dirs [B][ndir][H][da] by the complex step -> ddir [B][ndir]. */
int gpmpc_cpu_rollout_fullcov(int N, int ds, int da, int H, int B, const double* X, const double* Kinv, const double* Y,
                              const double* lam, const double* sf, const double* x0, const double* U, double gamma,
                              const double* Q, const double* R, const double* xref, const double* uref, double* means,
                              double* covs, double* cost, int ndir, const double* dirs, double* ddir, int nthreads) {
    const int D = ds + da;
    if (D > MAXD || ds < 1 || N < 1 || H < 1) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    double *beta, *Wv;
    if (build_constants(N, ds, D, X, Kinv, Y, lam, sf, &beta, &Wv)) return -2;
    ctx_t c = {N, ds, da, D, H, X, lam, sf, beta, Wv, gamma, Q, R, xref, uref};
    for (int b = 0; b < B; ++b) {
        cost[b] = rollout_one_real(&c, x0 + (size_t)b * ds, U + (size_t)b * H * da, means + (size_t)b * (H + 1) * ds, covs + (size_t)b * (H + 1) * ds * ds);
        for (int d = 0; d < ndir; ++d)
            rollout_one_cplx_entry(&c, x0 + (size_t)b * ds, U + (size_t)b * H * da, dirs + ((size_t)b * ndir + d) * H * da, 1e-20, ddir + (size_t)b * ndir + d);
    }
    free(beta); free(Wv);
    return 0;
}
#else
double rollout_one_cplx_entry(const ctx_t* c, const double* x0, const double* U, const double* dir, double h, double* dcost) {
    const int n = c->H * c->da;
    num* Uc = (num*)malloc(sizeof(num) * n);
    num* m = (num*)malloc(sizeof(num) * (size_t)(c->H + 1) * c->ds);
    num* s = (num*)malloc(sizeof(num) * (size_t)(c->H + 1) * c->ds * c->ds);
    for (int k = 0; k < n; ++k) Uc[k] = U[k] + I * (h * dir[k]);
    const num v = rollout_one_cplx(c, x0, Uc, m, s);
    *dcost = cimag(v) / h;
    free(Uc); free(m); free(s);
    return creal(v);
}
#endif
