// Full-covariance rollout (BASELINE config 5): the state distribution carries the full ds x ds covariance, with
// off-diagonal terms from the exact cross-covariance Cov[f_a, f_b] (covariance_prop_torch,
// src/tools/uncertainty_prop.py:402-465, consistent form).  The reference's rollout propagates variances only
// (src/dynamics.py:184-189: "So far, only paying attention to mean and variance. Implement covariance.") -- this is
// the build's extension of Dynamics.forward_propagate_torch + RiskSensitiveMPC.objective/gradient to full Sigma,
// oracled by composing the reference's own single-step functions (oracle/gpmpc_oracle.py).
//
// Per step: k_fc_assemble builds (u_t, S_t) = ([mu_{t-1}; U_{t-1}], blkdiag(Sigma_{t-1}, float32(1e-3) I)) for every
// trajectory, gpmpc_moment_match evaluates means, full covariance and their Jacobians (one pair-kernel launch over the
// ds variance units and the ds(ds-1)/2 cross units), the Jacobians of all steps are kept, and k_fc_tail evaluates the
// risk-sensitive cost with full Sigma (src/mpc.py:179-198) and runs the reverse sweep.
#include "gpmpc_internal.h"

int gpmpc_moment_match_ex(const gpmpc_pack* p, int nq, const double* u, const double* S, unsigned flags,
                          double* out_mean, double* out_var, double* out_cov, double* out_l, double* dmean_du,
                          double* dmean_dS, double* dvar_du, double* dvar_dS, double* dcov_du, double* dcov_dS,
                          void* workspace, size_t workspace_bytes, void* stream, int ns2);

struct FcArgs {
    int B, H, ds, da, D, grad;
    const double* x0; const double* U;
    double* u; double* S;                 // [B][D], [B][D][D]   inputs of the current step
    double* mean; double* cov;            // [B][ds], [B][ds][ds] outputs of the current step
    double* out_means; double* out_covs;  // [B][H+1][ds], [B][H+1][ds][ds]
    // Jacobians of every step t = 1..H at slice t-1: [H][B][...]
    double* dmean_du; double* dmean_dS; double* dcov_du; double* dcov_dS;
    double* out_cost; double* out_grad;
    gpmpc_cost_params cost;
};

// step t >= 1: record the state of step t-1 and assemble the input distribution of step t
__global__ void k_fc_assemble(FcArgs A, int t) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B) return;
    const int ds = A.ds, da = A.da, D = A.D, H = A.H;
    double* om = A.out_means + ((size_t)b * (H + 1) + (t - 1)) * ds;
    double* oc = A.out_covs + ((size_t)b * (H + 1) + (t - 1)) * ds * ds;
    double* u = A.u + (size_t)b * D;
    double* S = A.S + (size_t)b * D * D;
    for (int k = 0; k < ds; ++k) {
        const double m = (t == 1) ? A.x0[(size_t)b * ds + k] : A.mean[(size_t)b * ds + k];
        om[k] = m; u[k] = m;
        for (int l = 0; l < ds; ++l) {
            const double c = (t == 1) ? (k == l ? GPMPC_INIT_VAR : 0.0) : A.cov[((size_t)b * ds + k) * ds + l];
            oc[k * ds + l] = c;
            S[k * D + l] = c;
        }
        for (int l = ds; l < D; ++l) { S[k * D + l] = 0.0; S[l * D + k] = 0.0; }
    }
    if (t > H) return;                        // final call: only records step H
    for (int k = 0; k < da; ++k) {
        u[ds + k] = A.U[((size_t)b * H + (t - 1)) * da + k];
        for (int l = 0; l < da; ++l) S[(ds + k) * D + ds + l] = (k == l) ? GPMPC_ACTION_VAR : 0.0;
    }
}

// State cost with a full covariance and its derivatives (src/mpc.py:182-185):
//   (1/gamma) log det(I + gamma Q Sig) + e^T Z e,  Z = (I + gamma Q Sig)^-1 Q = (Q^-1 + gamma Sig)^-1
//   d/dmu = (Z + Z^T) e,   d/dSig = sym( Z^T - gamma (Z^T e)(Z e)^T )         (gamma = 0: Q^T, tr(Q Sig) + e^T Q e)
__device__ static double fc_state_cost(int ds, const gpmpc_cost_params& C, const double* mu, const double* Sig, double* w,
                                       double* dmu, double* dSig) {
    const double g = C.gamma;
    double e[GPMPC_MAX_DS];
    for (int k = 0; k < ds; ++k) e[k] = mu[k] - C.x_ref[k];
    const int ld = 2 * ds;
    double det = 1.0;
    if (g == 0.0) {
        for (int r = 0; r < ds; ++r) for (int c = 0; c < ds; ++c) w[r * ld + ds + c] = C.Q[r * ds + c];
    } else {
        for (int r = 0; r < ds; ++r)
            for (int c = 0; c < ds; ++c) {
                double s = 0.0;
                for (int l = 0; l < ds; ++l) s += C.Q[r * ds + l] * Sig[l * ds + c];
                w[r * ld + c] = (r == c ? 1.0 : 0.0) + g * s;
                w[r * ld + ds + c] = C.Q[r * ds + c];
            }
        for (int k = 0; k < ds; ++k) {
            int piv = k; double best = fabs(w[k * ld + k]);
            for (int r = k + 1; r < ds; ++r) { const double v = fabs(w[r * ld + k]); if (v > best) { best = v; piv = r; } }
            if (piv != k) {
                for (int c = 0; c < ld; ++c) { const double tmp = w[k * ld + c]; w[k * ld + c] = w[piv * ld + c]; w[piv * ld + c] = tmp; }
                det = -det;
            }
            const double pv = w[k * ld + k];
            det *= pv;
            const double inv = 1.0 / pv;
            for (int c = 0; c < ld; ++c) w[k * ld + c] *= inv;
            for (int r = 0; r < ds; ++r) {
                if (r == k) continue;
                const double f = w[r * ld + k];
                for (int c = 0; c < ld; ++c) w[r * ld + c] = fma(-f, w[k * ld + c], w[r * ld + c]);
            }
        }
    }
    double ze[GPMPC_MAX_DS], zte[GPMPC_MAX_DS], quad = 0.0, trq = 0.0;
    for (int k = 0; k < ds; ++k) {
        double s = 0.0, st = 0.0;
        for (int l = 0; l < ds; ++l) { s += w[k * ld + ds + l] * e[l]; st += w[l * ld + ds + k] * e[l]; trq += C.Q[k * ds + l] * Sig[l * ds + k]; }
        ze[k] = s; zte[k] = st;
        quad += e[k] * s;
    }
    if (dmu) {
        for (int k = 0; k < ds; ++k) dmu[k] = ze[k] + zte[k];
        for (int k = 0; k < ds; ++k)
            for (int l = 0; l < ds; ++l) {
                const double gkl = w[l * ld + ds + k] - g * zte[k] * ze[l];     // Z^T - gamma (Z^T e)(Z e)^T
                const double glk = w[k * ld + ds + l] - g * zte[l] * ze[k];
                dSig[k * ds + l] = 0.5 * (gkl + glk);
            }
    }
    return (g == 0.0 ? trq : log(det) / g) + quad;
}

__device__ static double fc_input_cost(int H, int da, const gpmpc_cost_params& C, const double* U, double* gU) {
    double c = 0.0;
    for (int j = 0; j < H; ++j) {
        double d[GPMPC_MAX_D];
        for (int k = 0; k < da; ++k) d[k] = U[j * da + k] - C.u_ref[k];
        for (int k = 0; k < da; ++k) {
            double rd = 0.0, rtd = 0.0;
            for (int l = 0; l < da; ++l) { rd += C.R[k * da + l] * d[l]; rtd += C.R[l * da + k] * d[l]; }
            c += d[k] * rd;
            if (gU) gU[j * da + k] += rd + rtd;
        }
        if (C.has_R_delta) {
            for (int k = 0; k < da; ++k) d[k] = U[j * da + k] - (j == 0 ? C.last_u[k] : U[(j - 1) * da + k]);
            for (int k = 0; k < da; ++k) {
                double rd = 0.0, rtd = 0.0;
                for (int l = 0; l < da; ++l) { rd += C.R_delta[k * da + l] * d[l]; rtd += C.R_delta[l * da + k] * d[l]; }
                c += d[k] * rd;
                if (gU) { gU[j * da + k] += rd + rtd; if (j > 0) gU[(j - 1) * da + k] -= rd + rtd; }
            }
        }
    }
    return c;
}

#define GPMPC_FC_WORKERS 32
// One workgroup (64 threads) per trajectory.  dynamic LDS:
//   [WORKERS][ds*2ds] LU scratch | [H+1] cost terms | [H+1][ds + ds*ds] local derivatives | adjoint (2 x (ds + ds*ds)) | g (D + D*D)
__global__ __launch_bounds__(64) void k_fc_tail(FcArgs A) {
    extern __shared__ double s_dyn[];
    const int b = blockIdx.x, ds = A.ds, da = A.da, D = A.D, H = A.H, nz = ds + ds * ds;
    double* s_lu = s_dyn;
    double* s_ct = s_lu + GPMPC_FC_WORKERS * ds * 2 * ds;
    double* s_dl = s_ct + (H + 1);
    double* s_adj = s_dl + (size_t)(H + 1) * nz;
    double* s_g = s_adj + 2 * nz;
    const double* mu = A.out_means + (size_t)b * (H + 1) * ds;
    const double* Sg = A.out_covs + (size_t)b * (H + 1) * ds * ds;
    for (int i = threadIdx.x; i <= H && threadIdx.x < GPMPC_FC_WORKERS; i += GPMPC_FC_WORKERS)
        s_ct[i] = fc_state_cost(ds, A.cost, mu + i * ds, Sg + i * ds * ds, s_lu + threadIdx.x * ds * 2 * ds,
                                A.grad ? s_dl + (size_t)i * nz : nullptr, A.grad ? s_dl + (size_t)i * nz + ds : nullptr);
    __syncthreads();
    const double* U = A.U + (size_t)b * H * da;
    double* gU = A.grad ? A.out_grad + (size_t)b * H * da : nullptr;
    if (threadIdx.x == 0) {
        if (gU) for (int q = 0; q < H * da; ++q) gU[q] = 0.0;
        double total = 0.0;
        for (int i = 0; i <= H; ++i) total += s_ct[i];
        total += fc_input_cost(H, da, A.cost, U, gU);
        A.out_cost[b] = total;
    }
    if (!A.grad) return;
    double* adj = s_adj; double* nxt = s_adj + nz;
    for (int r = threadIdx.x; r < nz; r += blockDim.x) adj[r] = s_dl[(size_t)H * nz + r];
    __syncthreads();
    const size_t B = A.B;
    for (int t = H; t >= 1; --t) {
        const double* dm_du = A.dmean_du + ((size_t)(t - 1) * B + b) * ds * D;
        const double* dm_dS = A.dmean_dS + ((size_t)(t - 1) * B + b) * ds * D * D;
        const double* dc_du = A.dcov_du + ((size_t)(t - 1) * B + b) * ds * ds * D;
        const double* dc_dS = A.dcov_dS + ((size_t)(t - 1) * B + b) * ds * ds * D * D;
        // g[k] (k < D): d/du_k ; g[D + k*D + l]: d/dS_kl
        for (int e = threadIdx.x; e < D + D * D; e += blockDim.x) {
            double s = 0.0;
            if (e < D) {
                for (int a = 0; a < ds; ++a) s = fma(adj[a], dm_du[a * D + e], s);
                for (int ab = 0; ab < ds * ds; ++ab) s = fma(adj[ds + ab], dc_du[ab * D + e], s);
            } else {
                const int kl = e - D;
                for (int a = 0; a < ds; ++a) s = fma(adj[a], dm_dS[a * D * D + kl], s);
                for (int ab = 0; ab < ds * ds; ++ab) s = fma(adj[ds + ab], dc_dS[ab * D * D + kl], s);
            }
            s_g[e] = s;
        }
        __syncthreads();
        for (int r = threadIdx.x; r < nz + da; r += blockDim.x) {
            if (r < ds) nxt[r] = s_dl[(size_t)(t - 1) * nz + r] + s_g[r];
            else if (r < nz) { const int k = (r - ds) / ds, l = (r - ds) - k * ds; nxt[r] = s_dl[(size_t)(t - 1) * nz + r] + s_g[D + k * D + l]; }
            else gU[(t - 1) * da + (r - nz)] += s_g[ds + (r - nz)];
        }
        __syncthreads();
        double* tmp = adj; adj = nxt; nxt = tmp;
    }
}

struct FcPlan { size_t off_u, off_S, off_mean, off_cov, off_var, off_dvu, off_dvS, off_dmu, off_dmS, off_dcu, off_dcS, off_mm, mm_bytes, total; };

static void plan_fc(const gpmpc_pack* p, int B, int H, bool grad, FcPlan* r) {
    const size_t ds = p->ds, D = p->D, HB = (size_t)(grad ? H : 1) * B;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n * sizeof(double) + 255) & ~(size_t)255; return o; };
    r->off_u = take((size_t)B * D); r->off_S = take((size_t)B * D * D);
    r->off_mean = take((size_t)B * ds); r->off_cov = take((size_t)B * ds * ds); r->off_var = take((size_t)B * ds);
    r->off_dvu = take((size_t)B * ds * D); r->off_dvS = take((size_t)B * ds * D * D);
    r->off_dmu = take(HB * ds * D); r->off_dmS = take(HB * ds * D * D);
    r->off_dcu = take(HB * ds * ds * D); r->off_dcS = take(HB * ds * ds * D * D);
    r->mm_bytes = gpmpc_moment_match_workspace_bytes(p, B);
    r->off_mm = off; off += (r->mm_bytes + 255) & ~(size_t)255;
    r->total = off;
}

extern "C" size_t gpmpc_rollout_fullcov_workspace_bytes(const gpmpc_pack* p, int B, int H, unsigned flags) {
    if (!p || B < 1 || H < 1) return 0;
    FcPlan r;
    plan_fc(p, B, H, (flags & GPMPC_WANT_GRAD) != 0, &r);
    return r.total;
}

extern "C" int gpmpc_rollout_fullcov(const gpmpc_pack* p, int B, int H, const double* x0, const double* U,
                                     const gpmpc_cost_params* cost, unsigned flags, double* out_means, double* out_covs,
                                     double* out_cost, double* out_grad, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!p) return GPMPC_E_ARG;
    if (int rc_dev = gpmpc_check_device(p)) return rc_dev;
    if (!p || !x0 || !U || !cost || !out_means || !out_covs || !out_cost || !workspace || B < 1 || H < 1) return GPMPC_E_ARG;
    if (!p->built || (p->npairs > 0 && !p->fullcov)) return GPMPC_E_STATE;
    const bool grad = (flags & GPMPC_WANT_GRAD) != 0;
    if (grad && !out_grad) return GPMPC_E_ARG;
    FcPlan r;
    plan_fc(p, B, H, grad, &r);
    if (workspace_bytes < r.total) return GPMPC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const size_t ds = p->ds, D = p->D;
    FcArgs A;
    memset(&A, 0, sizeof(A));
    A.B = B; A.H = H; A.ds = p->ds; A.da = p->da; A.D = p->D; A.grad = grad ? 1 : 0;
    A.x0 = x0; A.U = U;
    A.u = (double*)(ws + r.off_u); A.S = (double*)(ws + r.off_S);
    A.mean = (double*)(ws + r.off_mean); A.cov = (double*)(ws + r.off_cov);
    A.out_means = out_means; A.out_covs = out_covs;
    A.dmean_du = (double*)(ws + r.off_dmu); A.dmean_dS = (double*)(ws + r.off_dmS);
    A.dcov_du = (double*)(ws + r.off_dcu); A.dcov_dS = (double*)(ws + r.off_dcS);
    A.out_cost = out_cost; A.out_grad = out_grad; A.cost = *cost;
    double* var = (double*)(ws + r.off_var);
    double* dvu = (double*)(ws + r.off_dvu); double* dvS = (double*)(ws + r.off_dvS);
    const dim3 gb((B + 63) / 64), tb(64);
    for (int t = 1; t <= H; ++t) {
        hipLaunchKernelGGL(k_fc_assemble, gb, tb, 0, s, A, t);
        const size_t sl = grad ? (size_t)(t - 1) * B : 0;
        int rc = gpmpc_moment_match_ex(p, B, A.u, A.S, grad ? GPMPC_WANT_GRAD : 0u, A.mean, var, A.cov, nullptr,
                                    grad ? A.dmean_du + sl * ds * D : nullptr, grad ? A.dmean_dS + sl * ds * D * D : nullptr,
                                    grad ? dvu : nullptr, grad ? dvS : nullptr,
                                    grad ? A.dcov_du + sl * ds * ds * D : nullptr, grad ? A.dcov_dS + sl * ds * ds * D * D : nullptr,
                                    ws + r.off_mm, r.mm_bytes, stream, p->ds);
        if (rc != GPMPC_OK) return rc;
    }
    hipLaunchKernelGGL(k_fc_assemble, gb, tb, 0, s, A, H + 1);      // records step H
    const size_t nz = ds + ds * ds;
    const size_t lds = sizeof(double) * ((size_t)GPMPC_FC_WORKERS * ds * 2 * ds + (H + 1) + (size_t)(H + 1) * nz + 2 * nz + D + D * D);
    if (lds > 60 * 1024) return GPMPC_E_ARG;
    hipLaunchKernelGGL(k_fc_tail, dim3(B), dim3(64), lds, s, A);
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}
