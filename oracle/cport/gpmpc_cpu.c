/* TEST INFRASTRUCTURE ONLY -- plain-C / OpenMP CPU port of the O(N^2) rollout (the algorithm the GPU runs), used as
 * (i) the "algorithmic" CPU baseline of bench.py next to the faithful torch restatement, and (ii) an independent
 * full-size check of the analytic gradient (the torch oracle needs 46 GiB of autograd state at N = 2048, H = 20).
 * It is pinned to oracle/gpmpc_oracle.py (itself pinned to the reference's golden vectors) by
 * tests/test_oracle_golden.py::test_cport_matches_torch_oracle.  Never linked into the product.
 *
 * Restates, per trajectory: Dynamics.forward_propagate_torch (src/dynamics.py:126-191) with mean_prop_torch /
 * variance_prop_torch (src/tools/uncertainty_prop.py:296-399) in their elementwise O(N^2) form, cost_torch
 * (src/mpc.py:156-200, diagonal Sigma, general Q) and the gradient of src/mpc.py:251 as an analytic adjoint.
 * Direct exponent |h_i + h_j|^2 and libm exp: no table, no expansion -- deliberately a different evaluation order
 * from the HIP kernels.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 8

static double action_var(void) { return (double)1e-3f; }

/* Gauss-Jordan: solves Mx Z = Q (n x n), returns det(Mx); a is [n][2n] = [Mx | Q] on entry, [I | Z] on exit */
static double gj(int n, double* a) {
    const int ld = 2 * n; double det = 1.0;
    for (int k = 0; k < n; ++k) {
        int piv = k; double best = fabs(a[k * ld + k]);
        for (int r = k + 1; r < n; ++r) if (fabs(a[r * ld + k]) > best) { best = fabs(a[r * ld + k]); piv = r; }
        if (piv != k) { for (int c = 0; c < ld; ++c) { double t = a[k * ld + c]; a[k * ld + c] = a[piv * ld + c]; a[piv * ld + c] = t; } det = -det; }
        const double pv = a[k * ld + k]; det *= pv;
        for (int c = 0; c < ld; ++c) a[k * ld + c] /= pv;
        for (int r = 0; r < n; ++r) if (r != k) { const double f = a[r * ld + k]; for (int c = 0; c < ld; ++c) a[r * ld + c] -= f * a[k * ld + c]; }
    }
    return det;
}

static double state_cost(int ds, double gamma, const double* Q, const double* xref, const double* mu, const double* var,
                         double* dmu, double* dvar) {
    double e[MAXD], a[MAXD * 2 * MAXD], ze[MAXD], zte[MAXD];
    for (int k = 0; k < ds; ++k) e[k] = mu[k] - xref[k];
    const int ld = 2 * ds;
    double det = 1.0;
    for (int r = 0; r < ds; ++r) for (int c = 0; c < ds; ++c) {
        a[r * ld + c] = (r == c ? 1.0 : 0.0) + gamma * Q[r * ds + c] * var[c];
        a[r * ld + ds + c] = Q[r * ds + c];
    }
    if (gamma != 0.0) det = gj(ds, a);
    double quad = 0.0, trq = 0.0;
    for (int k = 0; k < ds; ++k) {
        double s = 0.0, st = 0.0;
        for (int l = 0; l < ds; ++l) { s += a[k * ld + ds + l] * e[l]; st += a[l * ld + ds + k] * e[l]; }
        ze[k] = s; zte[k] = st; quad += e[k] * s; trq += Q[k * ds + k] * var[k];
    }
    for (int k = 0; k < ds; ++k) { dmu[k] = ze[k] + zte[k]; dvar[k] = a[k * ld + ds + k] - gamma * zte[k] * ze[k]; }
    return (gamma == 0.0 ? trq : log(det) / gamma) + quad;
}

int gpmpc_cpu_rollout(int N, int ds, int da, int H, int B, const double* X, const double* Kinv, const double* Y,
                      const double* lam, const double* sf, const double* x0, const double* U, double gamma,
                      const double* Q, const double* R, const double* xref, const double* uref, double* means,
                      double* vars, double* cost, double* grad, int nthreads) {
    const int D = ds + da, nz = 2 * ds, nc = 2 * ds + da;
    if (D > MAXD || ds < 1 || N < 1) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    /* per-data constants: beta_a = Kinv_a y_a, M_a = sym(Kinv_a - beta beta^T) o exp(-1/4 d2) sf^4 */
    double* beta = (double*)malloc(sizeof(double) * ds * N);
    double* M = (double*)malloc(sizeof(double) * (size_t)ds * N * N);
    if (!beta || !M) return -2;
    for (int a = 0; a < ds; ++a) {
        const double* K = Kinv + (size_t)a * N * N;
#pragma omp parallel for
        for (int i = 0; i < N; ++i) { double s = 0.0; for (int j = 0; j < N; ++j) s += K[(size_t)i * N + j] * Y[(size_t)j * ds + a]; beta[a * N + i] = s; }
        const double sf4 = sf[a] * sf[a] * sf[a] * sf[a];
#pragma omp parallel for
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                double d2 = 0.0;
                for (int k = 0; k < D; ++k) { const double d = X[(size_t)i * D + k] - X[(size_t)j * D + k]; d2 += d * d / lam[a * D + k]; }
                M[((size_t)a * N + i) * N + j] = (0.5 * (K[(size_t)i * N + j] + K[(size_t)j * N + i]) - beta[a * N + i] * beta[a * N + j]) * sf4 * exp(-0.25 * d2);
            }
    }
    double* J = (double*)malloc(sizeof(double) * (size_t)H * nz * nc);
    double* rowz = (double*)malloc(sizeof(double) * (size_t)N * (1 + 2 * MAXD));
    if (!J || !rowz) { free(beta); free(M); free(J); free(rowz); return -2; }
    for (int b = 0; b < B; ++b) {
        double* mu = means + (size_t)b * (H + 1) * ds; double* va = vars + (size_t)b * (H + 1) * ds;
        for (int k = 0; k < ds; ++k) { mu[k] = x0[(size_t)b * ds + k]; va[k] = 1e-3; }
        memset(J, 0, sizeof(double) * (size_t)H * nz * nc);
        for (int t = 1; t <= H; ++t) {
            double u[MAXD], s[MAXD];
            for (int k = 0; k < ds; ++k) { u[k] = mu[(t - 1) * ds + k]; s[k] = va[(t - 1) * ds + k]; }
            for (int k = 0; k < da; ++k) { u[ds + k] = U[((size_t)b * H + (t - 1)) * da + k]; s[ds + k] = action_var(); }
            for (int a = 0; a < ds; ++a) {
                const double* la = lam + a * D; const double sf2 = sf[a] * sf[a];
                double Bk[MAXD], Ak[MAXD], sc[MAXD], detm = 1.0, detv = 1.0;
                for (int k = 0; k < D; ++k) { Bk[k] = 1.0 / (s[k] + la[k]); Ak[k] = 1.0 / (0.5 * la[k] + s[k]); sc[k] = sqrt(0.125 * Ak[k]); detm *= s[k] / la[k] + 1.0; detv *= 2.0 * s[k] / la[k] + 1.0; }
                const double cm = sf2 / sqrt(detm), c = 1.0 / sqrt(detv);
                double S0 = 0.0, S1[MAXD] = {0}, S2[MAXD] = {0};
                for (int i = 0; i < N; ++i) {
                    double q = 0.0, d[MAXD];
                    for (int k = 0; k < D; ++k) { d[k] = u[k] - X[(size_t)i * D + k]; q += Bk[k] * d[k] * d[k]; }
                    const double p = beta[a * N + i] * exp(-0.5 * q);
                    S0 += p; for (int k = 0; k < D; ++k) { S1[k] += p * d[k]; S2[k] += p * d[k] * d[k]; }
                }
                const double m = cm * S0;
                double Z0 = 0.0, Z1[MAXD] = {0}, Z2[MAXD] = {0};
                const double* Ma = M + (size_t)a * N * N;
                /* Row sums in parallel, then ONE fixed-order reduction in extended precision: the result does not depend on the
                 * thread count or schedule, and the checker stays more accurate than what it checks (thread-local running
                 * sums of N^2 / 2 / threads cancelling terms drift by 1e-3 of the variance at N = 8192). */
#pragma omp parallel for schedule(dynamic, 8)
                for (int i = 0; i < N; ++i) {
                    double hi[MAXD], z0 = 0.0, z1[MAXD] = {0}, z2[MAXD] = {0};
                    for (int k = 0; k < D; ++k) hi[k] = sc[k] * (u[k] - X[(size_t)i * D + k]);
                    for (int j = i; j < N; ++j) {
                        double mm[MAXD], ss = 0.0;
                        for (int k = 0; k < D; ++k) { mm[k] = hi[k] + sc[k] * (u[k] - X[(size_t)j * D + k]); ss += mm[k] * mm[k]; }
                        const double P = (i == j ? 1.0 : 2.0) * Ma[(size_t)i * N + j] * exp(-ss);
                        z0 += P; for (int k = 0; k < D; ++k) { z1[k] += P * mm[k]; z2[k] += P * mm[k] * mm[k]; }
                    }
                    double* rz = rowz + (size_t)i * (1 + 2 * MAXD);
                    rz[0] = z0; for (int k = 0; k < D; ++k) { rz[1 + k] = z1[k]; rz[1 + MAXD + k] = z2[k]; }
                }
                {
                    long double e0 = 0.0L, e1[MAXD] = {0}, e2[MAXD] = {0};
                    for (int i = 0; i < N; ++i) {
                        const double* rz = rowz + (size_t)i * (1 + 2 * MAXD);
                        e0 += rz[0]; for (int k = 0; k < D; ++k) { e1[k] += rz[1 + k]; e2[k] += rz[1 + MAXD + k]; }
                    }
                    Z0 = (double)e0; for (int k = 0; k < D; ++k) { Z1[k] = (double)e1[k]; Z2[k] = (double)e2[k]; }
                }
                const double T = c * Z0;
                mu[t * ds + a] = m; va[t * ds + a] = sf2 - T - m * m;
                double* jm = J + ((size_t)(t - 1) * nz + a) * nc; double* jv = J + ((size_t)(t - 1) * nz + ds + a) * nc;
                for (int k = 0; k < D; ++k) {
                    const double dmu_du = -Bk[k] * cm * S1[k], dmu_ds = -0.5 * m * Bk[k] + 0.5 * Bk[k] * Bk[k] * cm * S2[k];
                    const double dT_du = -4.0 * sc[k] * c * Z1[k], dT_ds = Ak[k] * (c * Z2[k] - 0.5 * T);
                    const double dv_du = -dT_du - 2.0 * m * dmu_du, dv_ds = -dT_ds - 2.0 * m * dmu_ds;
                    if (k < ds) { jm[k] = dmu_du; jm[ds + k] = dmu_ds; jv[k] = dv_du; jv[ds + k] = dv_ds; }
                    else { jm[2 * ds + k - ds] = dmu_du; jv[2 * ds + k - ds] = dv_du; }
                }
            }
        }
        /* cost and adjoint */
        double total = 0.0, dl[ (64 + 1) * 2 * MAXD ];
        if (H > 64) { free(beta); free(M); free(J); free(rowz); return -3; }
        for (int i = 0; i <= H; ++i) total += state_cost(ds, gamma, Q, xref, mu + i * ds, va + i * ds, dl + i * nz, dl + i * nz + ds);
        double* g = grad + (size_t)b * H * da;
        for (int j = 0; j < H; ++j) {
            double d[MAXD];
            for (int k = 0; k < da; ++k) d[k] = U[((size_t)b * H + j) * da + k] - uref[k];
            for (int k = 0; k < da; ++k) { double rd = 0.0, rtd = 0.0; for (int l = 0; l < da; ++l) { rd += R[k * da + l] * d[l]; rtd += R[l * da + k] * d[l]; } total += d[k] * rd; g[j * da + k] = rd + rtd; }
        }
        cost[b] = total;
        double adj[2 * MAXD], nx[2 * MAXD];
        for (int r = 0; r < nz; ++r) adj[r] = dl[H * nz + r];
        for (int t = H; t >= 1; --t) {
            const double* Jt = J + (size_t)(t - 1) * nz * nc;
            for (int k = 0; k < da; ++k) { double s2 = 0.0; for (int r = 0; r < nz; ++r) s2 += Jt[r * nc + nz + k] * adj[r]; g[(t - 1) * da + k] += s2; }
            for (int cc = 0; cc < nz; ++cc) { double s2 = dl[(t - 1) * nz + cc]; for (int r = 0; r < nz; ++r) s2 += Jt[r * nc + cc] * adj[r]; nx[cc] = s2; }
            for (int r = 0; r < nz; ++r) adj[r] = nx[r];
        }
    }
    free(beta); free(M); free(J); free(rowz);
    return 0;
}
