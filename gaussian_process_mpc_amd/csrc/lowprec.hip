// Reduced-precision variants of the rollout's N^2 sum, for the "fp64 vs fp32 tolerance sweep" of BASELINE config 3
// (SURVEY.md section 0 / build plan item 8).  NOT a fast path and never selected by default: the variance
// var = sf^2 - c Z0 - mu^2 is a cancelling sum (sum|terms| / |var| ~ 1e9 at N = 2048, sigma_n = 1e-2), so single
// precision is expected to FAIL the 1e-4 tolerance; these kernels exist to measure by how much.
//   mode 1 (GPMPC_FP32_ACCUM): exponent and exp in fp64, the products M_ij e^-s and their sum in fp32
//   mode 2 (GPMPC_FP32_ALL)  : transformed points, exponent, exp (v_exp_f32) and the sum in fp32; only M_ij is read as fp64
// Objective only (Z0), diagonal S; same work list, partial-sum layout and reduction as the fp64 kernels, so the head
// and tail kernels are unchanged.  Reference sum: src/tools/uncertainty_prop.py:372-399.
#include <hip/hip_runtime.h>
#include "gpmpc_internal.h"

template <int D, int MODE>
__global__ __launch_bounds__(256) void k_pair_lowprec(PairArgs A) {
    __shared__ double s_red[4];
    const int wi = blockIdx.x % A.nwork, b = blockIdx.x / A.nwork;
    const int unit = A.work[wi * 4 + 0], i0 = A.work[wi * 4 + 1], j0 = A.work[wi * 4 + 2], j1 = A.work[wi * 4 + 3];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, Np = A.Np;
    const int i = i0 + tid;
    const double* __restrict__ prm = A.pp + ((size_t)b * A.nunits + unit) * A.pps;      // cvec[D], scale[D]
    const double* __restrict__ Ma = A.M + (size_t)unit * Np * Np;
    double hi[D];
    float hif[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double x = i < Np ? A.XT[(size_t)k * Np + i] : 0.0;
        hi[k] = fma(-prm[D + k], x, prm[k]);
        hif[k] = (float)hi[k];
    }
    float accf = 0.0f;
    if (i < Np) {
        const int jstart = j0 > (i0 & ~63) ? j0 : (i0 & ~63);
        for (int j = jstart; j < j1; ++j) {
            const double mij = Ma[(size_t)j * Np + i];
            if (MODE == 1) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const double m = hi[k] + fma(-prm[D + k], A.XT[(size_t)k * Np + j], prm[k]);
                    s = fma(m, m, s);
                }
                accf += (float)(mij * exp(-s));
            } else {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const float m = hif[k] + (float)fma(-prm[D + k], A.XT[(size_t)k * Np + j], prm[k]);
                    s = fmaf(m, m, s);
                }
                accf = fmaf((float)mij, __expf(-s), accf);
            }
        }
    }
    const double ws = wave_sum((double)accf);
    if (lane == 0) s_red[w] = ws;
    __syncthreads();
    if (tid == 0) A.part[((size_t)b * A.nwork + wi) * A.nm] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

template <int D>
static void launch_lp(int mode, const PairArgs& a, hipStream_t s) {
    if (mode == 1) hipLaunchKernelGGL((k_pair_lowprec<D, 1>), dim3(a.B * a.nwork), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_pair_lowprec<D, 2>), dim3(a.B * a.nwork), dim3(256), 0, s, a);
}

// a: the 256x256 work list of the variance units, nm == 1 (objective only)
int gpmpc_launch_pair_lowprec(int D, int mode, const PairArgs& a, hipStream_t s) {
    if ((mode != 1 && mode != 2) || a.nm != 1) return GPMPC_E_ARG;
    switch (D) {
        case 1: launch_lp<1>(mode, a, s); break;
        case 2: launch_lp<2>(mode, a, s); break;
        case 3: launch_lp<3>(mode, a, s); break;
        case 4: launch_lp<4>(mode, a, s); break;
        case 5: launch_lp<5>(mode, a, s); break;
        case 6: launch_lp<6>(mode, a, s); break;
        case 7: launch_lp<7>(mode, a, s); break;
        case 8: launch_lp<8>(mode, a, s); break;
        default: return GPMPC_E_ARG;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("reduced-precision pair kernel launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}
