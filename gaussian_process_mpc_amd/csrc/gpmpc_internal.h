// Internal declarations shared by the HIP translation units of libgpmpc_hip.so.
// gfx950 (MI355X) only: 64-wide wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../include/gpmpc.h"

#define GPMPC_WAVE 64

// Variance of the initial state and of the action noise, as the reference has them
// (src/dynamics.py:148 float64 1e-3; src/dynamics.py:162 float32 1e-3 promoted to float64).
#define GPMPC_INIT_VAR   1e-3
#define GPMPC_ACTION_VAR ((double)1e-3f)

struct gpmpc_tiling {
    int it;         // rows per tile
    int waves;      // waves per workgroup of the v1 kernel: it / 64
    int jt;         // j-extent of a tile (multiple of 64)
    int ntiles;
    int* tiles_dev; // [ntiles][3] = {i0, j0, j1}
};

struct gpmpc_pack {
    int N, Np, ds, da, D;
    int built;
    double* X;      // dev [Np][D], rows >= N zero
    double* XT;     // dev [D][Np]
    double* beta;   // dev [ds][Np], zero padded
    double* M;      // dev [ds][Np][Np]: element (i,j), i<=j, lives at [j*Np+i]; weight 2 off the diagonal
    double* lam;    // dev [ds][D]
    double* sf;     // dev [ds]
    double lam_host[GPMPC_MAX_DS][GPMPC_MAX_D];
    double sf_host[GPMPC_MAX_DS];
    gpmpc_tiling tilings[3];   // [0] 256x256 (big batches), [1] 64x64 one-wave tiles (small batches), [2] 128x256 (v2, RI=2)
};

// Number of pair-kernel output moments per (trajectory, GP, tile).
static inline int gpmpc_num_moments(int D, bool diag, bool grad) {
    if (!grad) return 1;
    return diag ? 1 + 2 * D : 1 + D + D * (D + 1) / 2;
}

struct PairArgs {
    const double* M;
    const double* XT;
    const double* pp;     // [B][ds][pps]: cvec[D] then transform (diag: scale[D]; full: Cm[D][D] upper)
    double* part;         // [B][ds][ntiles][nm]
    const int* tiles;
    int Np, ds, B, ntiles, pps, nm;
    int ns2;              // leading dims whose second moments are needed (diag+grad path); D = all
};

// Implemented in pair_d*.hip (one translation unit per D so the build parallelises).
int gpmpc_launch_pair(int D, bool diag, bool grad, int tb, int waves, const PairArgs& a, hipStream_t s);
template <int D> int gpmpc_launch_pair_D(bool diag, bool grad, int tb, int waves, const PairArgs& a, hipStream_t s);
// v2 (MFMA moment accumulation; diagonal S, forward+gradient, D <= 7); variant: 0 = TB4/RI2, 1 = TB2/RI4, 2 = TB1/RI4
int gpmpc_launch_pair_mfma(int D, int variant, const PairArgs& a, hipStream_t s);
template <int D> int gpmpc_launch_pair_mfma_D(int variant, const PairArgs& a, hipStream_t s);

void gpmpc_set_error(const char* what, hipError_t e);
#define GPMPC_HIP(call)                                              \
    do {                                                             \
        hipError_t e_ = (call);                                      \
        if (e_ != hipSuccess) { gpmpc_set_error(#call, e_); return GPMPC_E_LAUNCH; } \
    } while (0)

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum NV per-thread values over the workgroup (<= 16 waves).  Result in out[0..NV) (LDS), visible to all
// threads after return.  scratch: LDS, >= 16*NV doubles.  Deterministic summation order.
template <int NV>
__device__ __forceinline__ void block_sum(const double (&v)[NV], double* scratch, double* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double s = wave_sum(v[k]);
        if (lane == 0) scratch[w * NV + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int ww = 0; ww < nw; ++ww) s += scratch[ww * NV + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}
