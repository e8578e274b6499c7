// Microbenchmark (MI355X): does v_mfma_f64_16x16x4_f64 overlap with fp64 VALU work?
//   mode 0: MFMA only   mode 1: VALU fp64 FMA only   mode 2: both interleaved in one wave
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_overlap.hip -o mfma_f64_overlap ; run: ./mfma_f64_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int MODE, int NM, int NV>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    v4f64 acc[8];
    double f[16];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f64){seed, seed, seed, seed};
    for (int i = 0; i < 16; ++i) f[i] = seed + i + threadIdx.x;
    double a = seed + threadIdx.x, b = 1.0000001;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < NV / 16; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = fma(f[i], b, a);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NM, int NV>
double run(int blocks, int threads, int iters) {
    double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(out);
    return ms;
}

int main() {
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps) {                  // waves per SIMD
        const int threads = 256, blocks = 256 * wps;     // one (or two) 4-wave workgroups per CU
        double t0 = run<0, 8, 32>(blocks, threads, iters), t1 = run<1, 8, 32>(blocks, threads, iters), t2 = run<2, 8, 32>(blocks, threads, iters);
        // per wave per iteration: 8 MFMA (8*1024 FMA) and 32 wave64 v_fma_f64
        const double clk = 2.4e9;
        printf("waves/SIMD=%d  MFMA-only %.3f ms (%.1f cyc/MFMA/SIMD)  VALU-only %.3f ms (%.2f cyc/v_fma/SIMD)  both %.3f ms  (sum %.3f, max %.3f)\n",
               wps, t0, t0 * 1e-3 * clk / (iters * 8.0 * wps), t1, t1 * 1e-3 * clk / (iters * 32.0 * wps), t2, t0 + t1, t0 > t1 ? t0 : t1);
    }
    return 0;
}
