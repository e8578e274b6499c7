// Gradient of the log marginal likelihood with respect to the log hyper-parameters, in one pass over Ky_inv.
// Reference: the autograd backward through inv / det in update_hyperparams (src/gpr.py:334-338) on the likelihood of
// src/gpr.py:240-251; the same quantity as marginal_likelihood_grad (src/gpr.py:200-238) computes from dense
// N x N x D derivative tensors:
//     d ml / d theta = 1/2 tr( (alpha alpha^T - Ky_inv) dKy/dtheta ),   alpha = Ky_inv r,  r = y - f_nom(X)
//     dKy/dlog lambda_k = Kf o (x_ik - x_jk)^2 / (2 lambda_k),   dKy/dlog sigma_f = 2 Kf,   dKy/dlog sigma_n = 2 sigma_n^2 I
// Kf is recomputed on the fly (one exp per element): the kernel reads Ky_inv once (8 N^2 bytes, HBM-bound) and never
// materialises a derivative matrix.  Per-workgroup partials are written and summed in a fixed order (bit-reproducible).
#include <hip/hip_runtime.h>
#include "gpmpc_internal.h"

#define GPMPC_ML_ROWS 8        // rows of Ky_inv per workgroup

template <int D>
__global__ __launch_bounds__(256) void k_ml_partial(int N, const double* __restrict__ X, const double* __restrict__ Kinv,
                                                    const double* __restrict__ alpha, const double* __restrict__ lam,
                                                    double sf2, double* __restrict__ part) {
    __shared__ double scratch[4 * (D + 1)];
    __shared__ double xi[GPMPC_ML_ROWS][D], ai[GPMPC_ML_ROWS];
    const int i0 = blockIdx.x * GPMPC_ML_ROWS;
    for (int t = threadIdx.x; t < GPMPC_ML_ROWS * D; t += blockDim.x) {
        const int r = t / D, k = t - r * D;
        xi[r][k] = i0 + r < N ? X[(size_t)(i0 + r) * D + k] : 0.0;
    }
    if (threadIdx.x < GPMPC_ML_ROWS) ai[threadIdx.x] = i0 + threadIdx.x < N ? alpha[i0 + threadIdx.x] : 0.0;
    double il[D];
#pragma unroll
    for (int k = 0; k < D; ++k) il[k] = 1.0 / lam[k];
    __syncthreads();
    double acc[D + 1];
#pragma unroll
    for (int m = 0; m <= D; ++m) acc[m] = 0.0;
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        double xj[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xj[k] = X[(size_t)j * D + k];
        const double aj = alpha[j];
        for (int r = 0; r < GPMPC_ML_ROWS && i0 + r < N; ++r) {
            double d2[D], e = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) { const double d = xi[r][k] - xj[k]; d2[k] = d * d * il[k]; e += d2[k]; }
            const double w = (ai[r] * aj - Kinv[(size_t)(i0 + r) * N + j]) * (sf2 * exp(-0.5 * e));
            acc[D] += w;
#pragma unroll
            for (int k = 0; k < D; ++k) acc[k] = fma(w, d2[k], acc[k]);
        }
    }
    block_sum<D + 1>(acc, scratch, part + (size_t)blockIdx.x * (D + 1));
}

// out[0..D-1] = d/dlog lambda_k, out[D] = d/dlog sigma_f, out[D+1] = d/dlog sigma_n, out[D+2] = r^T alpha
__global__ __launch_bounds__(256) void k_ml_finish(int N, int D, int nblocks, const double* __restrict__ part,
                                                   const double* __restrict__ Kinv, const double* __restrict__ alpha,
                                                   const double* __restrict__ resid, double noise_var,
                                                   double* __restrict__ out) {
    __shared__ double scratch[4 * 2];
    __shared__ double red[2];
    double v[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double a = alpha[i];
        v[0] += a * a - Kinv[(size_t)i * N + i];
        v[1] += resid[i] * a;
    }
    block_sum<2>(v, scratch, red);
    if (threadIdx.x <= D) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * (D + 1) + threadIdx.x];
        // dKy/dlog lambda_k carries 1/2 (the d2 above is (dx)^2 / lambda_k), and so does the trace formula
        out[threadIdx.x] = threadIdx.x < D ? 0.25 * s : s;
    }
    if (threadIdx.x == 0) { out[D + 1] = noise_var * red[0]; out[D + 2] = red[1]; }
}

extern "C" size_t gpmpc_ml_grad_workspace_bytes(int n, int D) {
    if (n < 1 || D < 1 || D > GPMPC_MAX_D) return 0;
    const size_t nblocks = (size_t)(n + GPMPC_ML_ROWS - 1) / GPMPC_ML_ROWS;
    return sizeof(double) * (nblocks * (D + 1) + GPMPC_MAX_D);
}

template <int D>
static void launch_ml_partial(int n, int nblocks, const double* X, const double* Kinv, const double* alpha, const double* lam,
                              double sf2, double* part, hipStream_t s) {
    hipLaunchKernelGGL(k_ml_partial<D>, dim3(nblocks), dim3(256), 0, s, n, X, Kinv, alpha, lam, sf2, part);
}

extern "C" int gpmpc_ml_grad(int n, int D, const double* X_dev, const double* Ky_inv_dev, const double* alpha_dev,
                             const double* resid_dev, const double* lambdas_host, double sigma_f, double noise_var,
                             double* out_dev, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 1 || D < 1 || D > GPMPC_MAX_D || !X_dev || !Ky_inv_dev || !alpha_dev || !resid_dev || !lambdas_host || !out_dev ||
        !workspace)
        return GPMPC_E_ARG;
    if (workspace_bytes < gpmpc_ml_grad_workspace_bytes(n, D)) return GPMPC_E_WORKSPACE;
    for (int k = 0; k < D; ++k) if (!(lambdas_host[k] > 0.0)) return GPMPC_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nblocks = (n + GPMPC_ML_ROWS - 1) / GPMPC_ML_ROWS;
    double* part = (double*)workspace;
    double* lam = part + (size_t)nblocks * (D + 1);
    if (int rcu = gpmpc_upload_small(lam, lambdas_host, sizeof(double) * D, s)) return rcu;
    const double sf2 = sigma_f * sigma_f;
    switch (D) {
        case 1: launch_ml_partial<1>(n, nblocks, X_dev, Ky_inv_dev, alpha_dev, lam, sf2, part, s); break;
        case 2: launch_ml_partial<2>(n, nblocks, X_dev, Ky_inv_dev, alpha_dev, lam, sf2, part, s); break;
        case 3: launch_ml_partial<3>(n, nblocks, X_dev, Ky_inv_dev, alpha_dev, lam, sf2, part, s); break;
        case 4: launch_ml_partial<4>(n, nblocks, X_dev, Ky_inv_dev, alpha_dev, lam, sf2, part, s); break;
        case 5: launch_ml_partial<5>(n, nblocks, X_dev, Ky_inv_dev, alpha_dev, lam, sf2, part, s); break;
        case 6: launch_ml_partial<6>(n, nblocks, X_dev, Ky_inv_dev, alpha_dev, lam, sf2, part, s); break;
        case 7: launch_ml_partial<7>(n, nblocks, X_dev, Ky_inv_dev, alpha_dev, lam, sf2, part, s); break;
        case 8: launch_ml_partial<8>(n, nblocks, X_dev, Ky_inv_dev, alpha_dev, lam, sf2, part, s); break;
        default: return GPMPC_E_ARG;
    }
    hipLaunchKernelGGL(k_ml_finish, dim3(1), dim3(256), 0, s, n, D, nblocks, part, Ky_inv_dev, alpha_dev, resid_dev, noise_var, out_dev);
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}
