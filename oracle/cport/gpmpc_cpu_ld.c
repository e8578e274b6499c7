/* TEST INFRASTRUCTURE ONLY -- the diagonal-covariance rollout of gpmpc_cpu.c (forward pass: means and variances) with
 * EVERY operation in x87 extended precision (long double, 64-bit significand; expl / sqrtl): beta = Ky_inv y, the folded
 * weights, the O(N) and O(N^2) sums and the moment formulas.  Given the same fp64 inputs (X, Ky_inv, y, hyper-parameters,
 * x0, U) it is ~2000 times more accurate than any fp64 evaluation, which makes it the yardstick at small noise levels, where
 * the variance is a sum that cancels to 1e-10 ... 1e-13 of its terms and two fp64 evaluation orders of the REFERENCE's own
 * formula (src/tools/uncertainty_prop.py:341-399) disagree with each other by per cents (tools/accuracy_stress.py).
 * Restates src/dynamics.py:126-191 with src/tools/uncertainty_prop.py:296-399 like gpmpc_cpu.c.  Small N only (x87 speed). */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 8
typedef long double ld;

int gpmpc_cpu_rollout_ld(int N, int ds, int da, int H, int B, const double* X, const double* Kinv, const double* Y,
                         const double* lam, const double* sf, const double* x0, const double* U, double* means,
                         double* vars, int nthreads) {
    const int D = ds + da;
    if (D > MAXD || ds < 1 || N < 1) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    ld* beta = (ld*)malloc(sizeof(ld) * (size_t)ds * N);
    ld* M = (ld*)malloc(sizeof(ld) * (size_t)ds * N * N);
    ld* rowz = (ld*)malloc(sizeof(ld) * (size_t)N);
    if (!beta || !M || !rowz) { free(beta); free(M); free(rowz); return -2; }
    for (int a = 0; a < ds; ++a) {
        const double* K = Kinv + (size_t)a * N * N;
#pragma omp parallel for
        for (int i = 0; i < N; ++i) { ld s = 0.0L; for (int j = 0; j < N; ++j) s += (ld)K[(size_t)i * N + j] * (ld)Y[(size_t)j * ds + a]; beta[(size_t)a * N + i] = s; }
        const ld sf2 = (ld)sf[a] * (ld)sf[a], sf4 = sf2 * sf2;
#pragma omp parallel for
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                ld d2 = 0.0L;
                for (int k = 0; k < D; ++k) { const ld d = (ld)X[(size_t)i * D + k] - (ld)X[(size_t)j * D + k]; d2 += d * d / (ld)lam[a * D + k]; }
                M[((size_t)a * N + i) * N + j] = (0.5L * ((ld)K[(size_t)i * N + j] + (ld)K[(size_t)j * N + i]) - beta[(size_t)a * N + i] * beta[(size_t)a * N + j]) * sf4 * expl(-0.25L * d2);
            }
    }
    const ld act_var = (ld)(double)1e-3f;                     /* src/dynamics.py:162: float32 1e-3 */
    for (int b = 0; b < B; ++b) {
        ld mu[MAXD], va[MAXD];
        for (int k = 0; k < ds; ++k) { mu[k] = (ld)x0[(size_t)b * ds + k]; va[k] = (ld)1e-3; means[((size_t)b * (H + 1)) * ds + k] = (double)mu[k]; vars[((size_t)b * (H + 1)) * ds + k] = (double)va[k]; }
        for (int t = 1; t <= H; ++t) {
            ld u[MAXD], s[MAXD], nmu[MAXD], nva[MAXD];
            for (int k = 0; k < ds; ++k) { u[k] = mu[k]; s[k] = va[k]; }
            for (int k = 0; k < da; ++k) { u[ds + k] = (ld)U[((size_t)b * H + (t - 1)) * da + k]; s[ds + k] = act_var; }
            for (int a = 0; a < ds; ++a) {
                const double* la = lam + a * D; const ld sf2 = (ld)sf[a] * (ld)sf[a];
                ld Bk[MAXD], sc[MAXD], detm = 1.0L, detv = 1.0L;
                for (int k = 0; k < D; ++k) {
                    Bk[k] = 1.0L / (s[k] + (ld)la[k]);
                    sc[k] = sqrtl(0.125L / (0.5L * (ld)la[k] + s[k]));
                    detm *= s[k] / (ld)la[k] + 1.0L; detv *= 2.0L * s[k] / (ld)la[k] + 1.0L;
                }
                const ld cm = sf2 / sqrtl(detm), c = 1.0L / sqrtl(detv);
                ld S0 = 0.0L;
                for (int i = 0; i < N; ++i) {
                    ld q = 0.0L;
                    for (int k = 0; k < D; ++k) { const ld d = u[k] - (ld)X[(size_t)i * D + k]; q += Bk[k] * d * d; }
                    S0 += beta[(size_t)a * N + i] * expl(-0.5L * q);
                }
                const ld m = cm * S0;
                const ld* Ma = M + (size_t)a * N * N;
#pragma omp parallel for schedule(dynamic, 8)
                for (int i = 0; i < N; ++i) {
                    ld hi[MAXD], z0 = 0.0L;
                    for (int k = 0; k < D; ++k) hi[k] = sc[k] * (u[k] - (ld)X[(size_t)i * D + k]);
                    for (int j = i; j < N; ++j) {
                        ld ss = 0.0L;
                        for (int k = 0; k < D; ++k) { const ld mm = hi[k] + sc[k] * (u[k] - (ld)X[(size_t)j * D + k]); ss += mm * mm; }
                        z0 += (i == j ? 1.0L : 2.0L) * Ma[(size_t)i * N + j] * expl(-ss);
                    }
                    rowz[i] = z0;
                }
                ld Z0 = 0.0L;
                for (int i = 0; i < N; ++i) Z0 += rowz[i];
                nmu[a] = m; nva[a] = sf2 - c * Z0 - m * m;
            }
            for (int k = 0; k < ds; ++k) {
                mu[k] = nmu[k]; va[k] = nva[k];
                means[((size_t)b * (H + 1) + t) * ds + k] = (double)mu[k]; vars[((size_t)b * (H + 1) + t) * ds + k] = (double)va[k];
            }
        }
    }
    free(beta); free(M); free(rowz);
    return 0;
}
