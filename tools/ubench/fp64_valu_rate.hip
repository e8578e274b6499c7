// Microbenchmark (MI355X): sustained issue cost of a wave64 v_fma_f64 / v_add_f64 / v_mul_f64 as a function of
// waves per SIMD, in SHADER cycles (s_memtime), plus the clock the chip holds (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* stamps, int iters, double seed) {
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = seed + i + threadIdx.x;
    const double a = seed + threadIdx.x * 1e-9, b = 1.0000001;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == 0) f[i] = fma(f[i], b, a);
                if (OP == 1) f[i] = f[i] + a;
                if (OP == 2) f[i] = f[i] * b;
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
}

template <int OP>
void run(const char* name, int wps) {
    const int threads = 256, blocks = 256 * wps, iters = 20000, waves = blocks * 4;
    double* out; unsigned long long* st;
    (void)hipMalloc(&out, sizeof(double) * blocks * threads);
    (void)hipMalloc(&st, sizeof(unsigned long long) * 2 * waves);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, st, 100, 1.0);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, st, iters, 1.0);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * waves);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w]; clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 100.0; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    // each wave issued iters*64 instructions while sharing its SIMD with wps-1 others
    printf("%-4s waves/SIMD=%d  median wave cycles/instr %.2f  => SIMD cycles per wave64 instr %.2f   clock %.0f MHz\n", name, wps,
           cyc[waves / 2] / (iters * 64.0), cyc[waves / 2] / (iters * 64.0) / wps, clk[waves / 2]);
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    for (int wps : {1, 2, 3, 4, 8}) {
        if (wps == 1) { run<0>("fma", 1); run<1>("add", 1); run<2>("mul", 1); }
        if (wps == 2) { run<0>("fma", 2); run<1>("add", 2); }
        if (wps == 3) run<0>("fma", 3);
        if (wps == 4) { run<0>("fma", 4); run<2>("mul", 4); }
        if (wps == 8) run<0>("fma", 8);
    }
    return 0;
}
