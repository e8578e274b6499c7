// Dispatch of the templated pair kernel and the GP prediction entry point
// (GaussianProcessRegression.compute_pred_train_covariance / predict_latent_vars, src/gpr.py:253-332).
#include "gpmpc_internal.h"

int gpmpc_launch_pair(int D, bool diag, bool grad, int tb, int waves, const PairArgs& a, hipStream_t s) {
    if (waves < 1 || waves > 4) return GPMPC_E_ARG;
    switch (D) {
        case 1: return gpmpc_launch_pair_D<1>(diag, grad, tb, waves, a, s);
        case 2: return gpmpc_launch_pair_D<2>(diag, grad, tb, waves, a, s);
        case 3: return gpmpc_launch_pair_D<3>(diag, grad, tb, waves, a, s);
        case 4: return gpmpc_launch_pair_D<4>(diag, grad, tb, waves, a, s);
        case 5: return gpmpc_launch_pair_D<5>(diag, grad, tb, waves, a, s);
        case 6: return gpmpc_launch_pair_D<6>(diag, grad, tb, waves, a, s);
        case 7: return gpmpc_launch_pair_D<7>(diag, grad, tb, waves, a, s);
        case 8: return gpmpc_launch_pair_D<8>(diag, grad, tb, waves, a, s);
    }
    return GPMPC_E_ARG;
}

int gpmpc_launch_pair_sb(int D, bool grad, int tb, int ns2, int waves, const PairSbArgs& a, hipStream_t s) {
    if (waves < 1 || waves > 4) return GPMPC_E_ARG;
    switch (D) {
        case 1: return gpmpc_launch_pair_sb_D<1>(grad, tb, ns2, waves, a, s);
        case 2: return gpmpc_launch_pair_sb_D<2>(grad, tb, ns2, waves, a, s);
        case 3: return gpmpc_launch_pair_sb_D<3>(grad, tb, ns2, waves, a, s);
        case 4: return gpmpc_launch_pair_sb_D<4>(grad, tb, ns2, waves, a, s);
        case 5: return gpmpc_launch_pair_sb_D<5>(grad, tb, ns2, waves, a, s);
        case 6: return gpmpc_launch_pair_sb_D<6>(grad, tb, ns2, waves, a, s);
        case 7: return gpmpc_launch_pair_sb_D<7>(grad, tb, ns2, waves, a, s);
        case 8: return gpmpc_launch_pair_sb_D<8>(grad, tb, ns2, waves, a, s);
    }
    return GPMPC_E_ARG;
}

int gpmpc_launch_pair_sbs(int D, bool grad, int ng, int ns2, const PairSbsArgs& a, hipStream_t s) {
    switch (D) {
        case 2: return gpmpc_launch_pair_sbs_D<2>(grad, ng, ns2, a, s);
        case 3: return gpmpc_launch_pair_sbs_D<3>(grad, ng, ns2, a, s);
        case 4: return gpmpc_launch_pair_sbs_D<4>(grad, ng, ns2, a, s);
        case 5: return gpmpc_launch_pair_sbs_D<5>(grad, ng, ns2, a, s);
        case 6: return gpmpc_launch_pair_sbs_D<6>(grad, ng, ns2, a, s);
        case 7: return gpmpc_launch_pair_sbs_D<7>(grad, ng, ns2, a, s);
        case 8: return gpmpc_launch_pair_sbs_D<8>(grad, ng, ns2, a, s);
    }
    return GPMPC_E_ARG;
}

int gpmpc_launch_pair_sbf(int D, bool grad, int ns2, int waves, const PairSbfArgs& a, hipStream_t s) {
    if (waves < 1 || waves > 4) return GPMPC_E_ARG;
    switch (D) {
        case 1: return gpmpc_launch_pair_sbf_D<1>(grad, ns2, waves, a, s);
        case 2: return gpmpc_launch_pair_sbf_D<2>(grad, ns2, waves, a, s);
        case 3: return gpmpc_launch_pair_sbf_D<3>(grad, ns2, waves, a, s);
        case 4: return gpmpc_launch_pair_sbf_D<4>(grad, ns2, waves, a, s);
        case 5: return gpmpc_launch_pair_sbf_D<5>(grad, ns2, waves, a, s);
        case 6: return gpmpc_launch_pair_sbf_D<6>(grad, ns2, waves, a, s);
        case 7: return gpmpc_launch_pair_sbf_D<7>(grad, ns2, waves, a, s);
        case 8: return gpmpc_launch_pair_sbf_D<8>(grad, ns2, waves, a, s);
    }
    return GPMPC_E_ARG;
}

int gpmpc_launch_pair_sbfx(int D, bool grad, int ns2, const PairSbfxArgs& a, hipStream_t s) {
    switch (D) {
        case 3: return gpmpc_launch_pair_sbfx_D<3>(grad, ns2, a, s);
        case 4: return gpmpc_launch_pair_sbfx_D<4>(grad, ns2, a, s);
        case 5: return gpmpc_launch_pair_sbfx_D<5>(grad, ns2, a, s);
        case 6: return gpmpc_launch_pair_sbfx_D<6>(grad, ns2, a, s);
    }
    return GPMPC_E_ARG;
}

// K*[r][i] = sf^2 exp(-1/2 sum_k (xp_rk - x_ik)^2 / lambda_k)      (src/gpr.py:266-283)
__global__ void k_cross_kernel(const double* __restrict__ Xp, int p, const double* __restrict__ X, int N, int D,
                               const double* __restrict__ lam, double sf2, double* __restrict__ K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (i >= N) return;
    double d2 = 0.0;
    for (int k = 0; k < D; ++k) {
        const double d = Xp[(size_t)r * D + k] - X[(size_t)i * D + k];
        d2 = fma(d * d, 1.0 / lam[k], d2);
    }
    K[(size_t)r * N + i] = sf2 * exp(-0.5 * d2);
}

// out[r] = A[r] . v   (rows of a row-major [rows][N] matrix); one workgroup per row.
// Used for beta = Ky_inv y (src/tools/uncertainty_prop.py:327, src/gpr.py:306) and mean = K* beta.
__global__ __launch_bounds__(256) void k_rows_dot(const double* __restrict__ A, const double* __restrict__ v, int N,
                                                   double* __restrict__ out) {
    __shared__ double s_scr[16], s_out[1];
    const int r = blockIdx.x;
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < N; i += blockDim.x) acc[0] = fma(A[(size_t)r * N + i], v[i], acc[0]);
    block_sum<1>(acc, s_scr, s_out);
    if (threadIdx.x == 0) out[r] = s_out[0];
}

// W = K* Ky_inv   ([p][N]).  One workgroup of 16 waves per (64 columns, 8 test points): the waves split the contraction
// index (each reads 512 contiguous bytes of a Ky_inv row per step, shared by the 8 test points whose K* entries are
// wave-uniform) and combine in LDS in a fixed order.  A single test point at N = 2048 used to walk the 2048 rows with one
// thread per column (0.59 ms -> 0.15 ms for the whole predict call); 512 points re-read Ky_inv 64 instead of 512 times.
#define GPMPC_PRED_RT 8
__global__ __launch_bounds__(1024) void k_pred_w(const double* __restrict__ K, const double* __restrict__ Kinv, int N, int p,
                                                 double* __restrict__ W) {
    __shared__ double s_part[16][GPMPC_PRED_RT][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r0 = blockIdx.y * GPMPC_PRED_RT;
    const int c = blockIdx.x * 64 + lane;
    const int per = (N + 15) / 16, k0 = w * per, k1 = (k0 + per < N) ? k0 + per : N;
    double s[GPMPC_PRED_RT];
#pragma unroll
    for (int q = 0; q < GPMPC_PRED_RT; ++q) s[q] = 0.0;
    if (c < N)
        for (int k = k0; k < k1; ++k) {
            const double a = Kinv[(size_t)k * N + c];
#pragma unroll
            for (int q = 0; q < GPMPC_PRED_RT; ++q) {
                const int r = r0 + q < p ? r0 + q : p - 1;                  // padded rows recompute the last point
                s[q] = fma(K[(size_t)r * N + k], a, s[q]);
            }
        }
#pragma unroll
    for (int q = 0; q < GPMPC_PRED_RT; ++q) s_part[w][q][lane] = s[q];
    __syncthreads();
    for (int q = w; q < GPMPC_PRED_RT; q += 16) {
        if (r0 + q < p && c < N) {
            double t = 0.0;
#pragma unroll
            for (int ww = 0; ww < 16; ++ww) t += s_part[ww][q][lane];
            W[(size_t)(r0 + q) * N + c] = t;
        }
    }
}

// cov[r][s] = K**(r,s) - W[r] . K*[s] + noise_var [r == s]     (src/gpr.py:320-330)
__global__ __launch_bounds__(256) void k_pred_cov(const double* __restrict__ Xp, int p, int D, const double* __restrict__ lam,
                                                   double sf2, const double* __restrict__ W, const double* __restrict__ K,
                                                   int N, double noise_var, double* __restrict__ cov) {
    __shared__ double s_scr[16], s_out[1];
    const int r = blockIdx.x, s = blockIdx.y;
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < N; i += blockDim.x) v[0] = fma(W[(size_t)r * N + i], K[(size_t)s * N + i], v[0]);
    block_sum<1>(v, s_scr, s_out);
    if (threadIdx.x == 0) {
        double d2 = 0.0;
        for (int k = 0; k < D; ++k) {
            const double d = Xp[(size_t)r * D + k] - Xp[(size_t)s * D + k];
            d2 = fma(d * d, 1.0 / lam[k], d2);
        }
        cov[(size_t)r * p + s] = sf2 * exp(-0.5 * d2) - s_out[0] + (r == s ? noise_var : 0.0);
    }
}

extern "C" int gpmpc_matvec(int rows, int cols, const double* A_dev, const double* v_dev, double* out_dev, void* stream) {
    if (rows < 1 || cols < 1 || !A_dev || !v_dev || !out_dev) return GPMPC_E_ARG;
    hipLaunchKernelGGL(k_rows_dot, dim3(rows), dim3(256), 0, (hipStream_t)stream, A_dev, v_dev, cols, out_dev);
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}

extern "C" size_t gpmpc_predict_workspace_bytes(int n, int D, int np) {
    if (n < 1 || np < 1 || D < 1) return 0;
    return 2 * ((sizeof(double) * (size_t)np * n + 255) & ~(size_t)255) + 256;
}

extern "C" int gpmpc_predict(int n, int D, const double* X, const double* lambdas_host, double sigma_f,
                             const double* beta, const double* Kinv, double noise_var, int np, const double* Xp,
                             double* out_K, double* out_mean, double* out_cov, void* workspace, size_t workspace_bytes,
                             void* stream) {
    if (n < 1 || D < 1 || D > GPMPC_MAX_D || np < 1 || !X || !lambdas_host || !Xp || !workspace) return GPMPC_E_ARG;
    if ((out_mean && !beta) || (out_cov && !Kinv)) return GPMPC_E_ARG;
    if (workspace_bytes < gpmpc_predict_workspace_bytes(n, D, np)) return GPMPC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const size_t slab = (sizeof(double) * (size_t)np * n + 255) & ~(size_t)255;
    double* K = out_K ? out_K : (double*)workspace;
    double* W = (double*)((char*)workspace + slab);
    double* lam = (double*)((char*)workspace + 2 * slab);
    if (int rcu = gpmpc_upload_small(lam, lambdas_host, sizeof(double) * D, s)) return rcu;
    const double sf2 = sigma_f * sigma_f;
    hipLaunchKernelGGL(k_cross_kernel, dim3((n + 255) / 256, np), dim3(256), 0, s, Xp, np, X, n, D, lam, sf2, K);
    if (out_mean) hipLaunchKernelGGL(k_rows_dot, dim3(np), dim3(256), 0, s, K, beta, n, out_mean);
    if (out_cov) {
        hipLaunchKernelGGL(k_pred_w, dim3((n + 63) / 64, (np + GPMPC_PRED_RT - 1) / GPMPC_PRED_RT), dim3(1024), 0, s, K, Kinv, n, np, W);
        hipLaunchKernelGGL(k_pred_cov, dim3(np, np), dim3(256), 0, s, Xp, np, D, lam, sf2, W, K, n, noise_var, out_cov);
    }
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}

// ---------------------------------------------------------------------------
// O(N^2) append of ONE observation to an explicit inverse (block-inverse / Schur-complement update; the idea of the
// reference's update_Ky_inv_mat, src/gpr.py:137-157, which it abandons in favour of the O(N^3) rebuild):
//   Ky' = [[Ky, k],[k^T, kappa]],   v = Ky_inv k,  w = Ky_inv^T k,  q = 1 / (kappa - k^T v)
//   Ky'_inv = [[Ky_inv + q v w^T, -q v], [-q w^T, q]]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_append_vw(const double* __restrict__ Kinv, size_t ld, const double* __restrict__ k, int n,
                                                    double* __restrict__ v, double* __restrict__ wv) {
    // row r: v[r] = Kinv[r] . k ; wv[r] = Kinv[:, r] . k  (second read is strided; n^2 doubles once per append)
    __shared__ double s_scr[32], s_out[2];
    const int r = blockIdx.x;
    double acc[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        acc[0] = fma(Kinv[(size_t)r * ld + i], k[i], acc[0]);
        acc[1] = fma(Kinv[(size_t)i * ld + r], k[i], acc[1]);
    }
    block_sum<2>(acc, s_scr, s_out);
    if (threadIdx.x == 0) { v[r] = s_out[0]; wv[r] = s_out[1]; }
}

__global__ __launch_bounds__(256) void k_append_q(const double* __restrict__ k, const double* __restrict__ v, int n, double kappa,
                                                   double* __restrict__ q) {
    __shared__ double s_scr[16], s_out[1];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc[0] = fma(k[i], v[i], acc[0]);
    block_sum<1>(acc, s_scr, s_out);
    if (threadIdx.x == 0) q[0] = 1.0 / (kappa - s_out[0]);
}

// Kf_in / Ky_in / Kf_out / Ky_out (optional, all or none): the kernel matrices follow in the same pass -- old block copied, new row and
// column k, corner kff (+ noise on Ky) -- so that an append is ONE set of launches and no host-side concatenation.
__global__ void k_append_fill(const double* __restrict__ Kinv, size_t ld_in, const double* __restrict__ v, const double* __restrict__ wv,
                              const double* __restrict__ q, int n, double* __restrict__ out, size_t ld_out,
                              const double* __restrict__ k, const double* __restrict__ Kf_in, const double* __restrict__ Ky_in, size_t ld_k,
                              double* __restrict__ Kf_out, double* __restrict__ Ky_out, double kff, double noise_var) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, m = n + 1;
    if (j >= m) return;
    const double qq = q[0];
    double val;
    if (i < n && j < n) val = fma(qq * v[i], wv[j], Kinv[(size_t)i * ld_in + j]);
    else if (i < n) val = -qq * v[i];
    else if (j < n) val = -qq * wv[j];
    else val = qq;
    out[(size_t)i * ld_out + j] = val;
    if (Kf_out) {
        double kf, ky;
        if (i < n && j < n) { kf = Kf_in[(size_t)i * ld_k + j]; ky = Ky_in[(size_t)i * ld_k + j]; }
        else if (i < n) { kf = k[i]; ky = kf; }
        else if (j < n) { kf = k[j]; ky = kf; }
        else { kf = kff; ky = kff + noise_var; }
        Kf_out[(size_t)i * ld_out + j] = kf;
        Ky_out[(size_t)i * ld_out + j] = ky;
    }
}

// k[i] = sigma_f^2 exp(-1/2 sum_d (x_i - x_new)_d^2 / lambda_d)   (src/gpr.py:124-135 against every training input)
__global__ void k_append_kvec(const double* __restrict__ X, int n, int D, const double* __restrict__ xnew, const double* __restrict__ lam,
                              double sf2, double* __restrict__ k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d2 = 0.0;
    for (int d = 0; d < D; ++d) { const double t = X[(size_t)i * D + d] - xnew[d]; d2 = fma(t * t, 1.0 / lam[d], d2); }
    k[i] = sf2 * exp(-0.5 * d2);
}

extern "C" size_t gpmpc_kinv_append_workspace_bytes(int n) { return n < 1 ? 0 : sizeof(double) * (2 * (size_t)n + 8); }

extern "C" int gpmpc_kinv_append(int n, const double* Kinv_dev, const double* k_dev, double kappa, double* out_dev,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 1 || !Kinv_dev || !k_dev || !out_dev || !workspace) return GPMPC_E_ARG;
    if (workspace_bytes < gpmpc_kinv_append_workspace_bytes(n)) return GPMPC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* v = (double*)workspace; double* wv = v + n; double* q = wv + n;
    hipLaunchKernelGGL(k_append_vw, dim3(n), dim3(256), 0, s, Kinv_dev, (size_t)n, k_dev, n, v, wv);
    hipLaunchKernelGGL(k_append_q, dim3(1), dim3(256), 0, s, k_dev, v, n, kappa, q);
    hipLaunchKernelGGL(k_append_fill, dim3((n + 1 + 255) / 256, n + 1), dim3(256), 0, s, Kinv_dev, (size_t)n, v, wv, q, n, out_dev,
                       (size_t)(n + 1), (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, (size_t)0, (double*)nullptr,
                       (double*)nullptr, 0.0, 0.0);
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}

// The whole data-update of ONE appended observation for a GP kept in capacity-padded buffers (closed loop, src/simulator.py:55 ->
// src/gpr.py:90-122, :159-171): k = K_f(X, x_new), then Kf, Ky and Ky_inv of the n + 1 points written into the OUTPUT buffers (leading
// dimension ld_out >= n + 1) from the INPUT buffers (ld_in >= n); input and output must not alias (ping-pong two buffer sets).
// Four launches, no allocation, no host round trip.
struct AppendHyp { double lam[GPMPC_MAX_D]; double sf2; };

// k_append_kvec + k_append_vw in one launch: row workgroup r evaluates k itself (workgroup 0 also stores it) and v[r], wv[r]
__global__ __launch_bounds__(256) void k_append_vw2(const double* __restrict__ X, int n, int D, const double* __restrict__ xnew, AppendHyp H,
                                                     const double* __restrict__ Kinv, size_t ld, double* __restrict__ kout,
                                                     double* __restrict__ v, double* __restrict__ wv) {
    __shared__ double s_scr[32], s_out[2];
    const int r = blockIdx.x;
    double acc[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double d2 = 0.0;
        for (int d = 0; d < D; ++d) { const double t = X[(size_t)i * D + d] - xnew[d]; d2 = fma(t * t, 1.0 / H.lam[d], d2); }
        const double ki = H.sf2 * exp(-0.5 * d2);
        if (r == 0) kout[i] = ki;
        acc[0] = fma(Kinv[(size_t)r * ld + i], ki, acc[0]);
        acc[1] = fma(Kinv[(size_t)i * ld + r], ki, acc[1]);
    }
    block_sum<2>(acc, s_scr, s_out);
    if (threadIdx.x == 0) { v[r] = s_out[0]; wv[r] = s_out[1]; }
}

// k_append_q + k_append_fill in one launch: every workgroup evaluates q = 1 / (kappa - k . v) itself (n products, fixed order)
__global__ __launch_bounds__(256) void k_append_fill2(const double* __restrict__ Kinv, size_t ld_in, const double* __restrict__ v,
                                                       const double* __restrict__ wv, double kappa, int n, double* __restrict__ out, size_t ld_out,
                                                       const double* __restrict__ k, const double* __restrict__ Kf_in, const double* __restrict__ Ky_in,
                                                       size_t ld_k, double* __restrict__ Kf_out, double* __restrict__ Ky_out, double kff, double noise_var) {
    __shared__ double s_scr[16], s_out[1];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc[0] = fma(k[i], v[i], acc[0]);
    block_sum<1>(acc, s_scr, s_out);
    const double qq = 1.0 / (kappa - s_out[0]);
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, m = n + 1;
    if (j >= m) return;
    double val;
    if (i < n && j < n) val = fma(qq * v[i], wv[j], Kinv[(size_t)i * ld_in + j]);
    else if (i < n) val = -qq * v[i];
    else if (j < n) val = -qq * wv[j];
    else val = qq;
    out[(size_t)i * ld_out + j] = val;
    if (Kf_out) {
        double kf, ky;
        if (i < n && j < n) { kf = Kf_in[(size_t)i * ld_k + j]; ky = Ky_in[(size_t)i * ld_k + j]; }
        else if (i < n) { kf = k[i]; ky = kf; }
        else if (j < n) { kf = k[j]; ky = kf; }
        else { kf = kff; ky = kff + noise_var; }
        Kf_out[(size_t)i * ld_out + j] = kf;
        Ky_out[(size_t)i * ld_out + j] = ky;
    }
}

extern "C" size_t gpmpc_gp_append_workspace_bytes(int n, int D) { (void)D; return n < 1 ? 0 : sizeof(double) * (3 * (size_t)n + 8); }   // k, v, w (the two-launch form keeps nothing else)

extern "C" int gpmpc_gp_append(int n, int D, const double* X_dev, const double* xnew_dev, const double* lambdas_host, double sigma_f,
                               double noise_var, const double* Kf_in, const double* Ky_in, size_t ld_k_in, const double* Kinv_in, size_t ld_in,
                               double* Kf_out, double* Ky_out, double* Kinv_out, size_t ld_out, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (n < 1 || D < 1 || D > GPMPC_MAX_D || !X_dev || !xnew_dev || !lambdas_host || !Kf_in || !Ky_in || !Kinv_in || !Kf_out || !Ky_out ||
        !Kinv_out || !workspace || ld_in < (size_t)n || ld_k_in < (size_t)n || ld_out < (size_t)n + 1)
        return GPMPC_E_ARG;
    if (Kf_in == Kf_out || Ky_in == Ky_out || Kinv_in == Kinv_out) return GPMPC_E_ARG;
    if (workspace_bytes < gpmpc_gp_append_workspace_bytes(n, D)) return GPMPC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* v = (double*)workspace; double* wv = v + n; double* k = wv + n;
    // Two launches (they were five: lambda upload, k vector, v / w, q, fill -- each a dependent launch boundary on a path that is
    // launch latency and nothing else at n ~ 300): the hyper-parameters travel as kernel arguments, every row workgroup evaluates
    // the k vector itself, every fill workgroup the scalar q.  Same expressions, same summation orders: bit-identical results.
    AppendHyp hyp;
    memset(&hyp, 0, sizeof(hyp));
    for (int d = 0; d < D; ++d) hyp.lam[d] = lambdas_host[d];
    hyp.sf2 = sigma_f * sigma_f;
    hipLaunchKernelGGL(k_append_vw2, dim3(n), dim3(256), 0, s, X_dev, n, D, xnew_dev, hyp, Kinv_in, ld_in, k, v, wv);
    hipLaunchKernelGGL(k_append_fill2, dim3((n + 1 + 255) / 256, n + 1), dim3(256), 0, s, Kinv_in, ld_in, (const double*)v, (const double*)wv,
                       hyp.sf2 + noise_var, n, Kinv_out, ld_out, (const double*)k, Kf_in, Ky_in, ld_k_in, Kf_out, Ky_out, hyp.sf2, noise_var);
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}
