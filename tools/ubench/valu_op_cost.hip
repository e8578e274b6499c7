// Microbenchmark (MI355X): issue cost of the wave64 VALU instructions the table exp of the pair kernel is made of,
// relative to v_fma_f64, at 2 waves per SIMD (two co-resident 256-thread workgroups per CU).  Cycles from s_memtime.
//   hipcc -O3 --offload-arch=gfx950 valu_op_cost.hip -o valu_op_cost && ./valu_op_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* stamps, int iters, double seed) {
    double f[16];
    int n[16];
    for (int i = 0; i < 16; ++i) { f[i] = seed + 0.37 * i + threadIdx.x * 1e-3; n[i] = threadIdx.x + i; }
    const double a = 0.999999, b = 1e-7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define ONE(i)                                                                                                      \
    if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(a), "v"(b));                            \
    if (OP == 1) asm volatile("v_rndne_f64 %0, %0" : "+v"(f[i]));                                                   \
    if (OP == 2) asm volatile("v_fract_f64 %0, %0" : "+v"(f[i]));                                                   \
    if (OP == 3) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(f[i]));                                     \
    if (OP == 4) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(f[i]) : "v"(n[i] & 1));                               \
    if (OP == 5) asm volatile("v_and_b32 %0, 0x7ff, %0" : "+v"(n[i]));                                              \
    if (OP == 6) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(n[i]));                                              \
    if (OP == 7) asm volatile("v_ashrrev_i32 %0, 11, %0" : "+v"(n[i]));                                             \
    if (OP == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(a));                                        \
    if (OP == 9) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(b));                                        \
    if (OP == 10) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));                 \
    if (OP == 11) asm volatile("v_max_i32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));                         \
    if (OP == 12) asm volatile("v_mov_b64 %0, %1" : "=v"(f[i]) : "v"(a));                                           \
    if (OP == 13) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));                         \
    if (OP == 14) asm volatile("v_add_u32 %0, 0xffd00000, %0" : "+v"(n[i]));                                        \
    if (OP == 15) asm volatile("v_mul_u32_u24 %0, 8, %0" : "+v"(n[i]));                                             \
    if (OP == 16) asm volatile("v_bfe_u32 %0, %0, 3, 11" : "+v"(n[i]));                                             \
    if (OP == 17) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));                         \
    if (OP == 18) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(n[i]) : "v"(n[(i + 1) & 15] & 3));                 \
    if (OP == 19) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(n[i]));                                             \
    if (OP == 20) asm volatile("v_and_or_b32 %0, %0, %2, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]), "v"(0x3ff8));              \
    if (OP == 21) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 15]));
            REP16(ONE)
#undef ONE
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    int m = 0;
    for (int i = 0; i < 16; ++i) { s += f[i]; m += n[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + m;
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
static double run(const char* name, double ref) {
    const int threads = 256, blocks = 512, iters = 4000, waves = blocks * 4;
    double* out; unsigned long long* st;
    (void)hipMalloc(&out, sizeof(double) * blocks * threads);
    (void)hipMalloc(&st, sizeof(unsigned long long) * waves);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, st, 50, 1.0);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, st, iters, 1.0);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double c = (double)h[waves / 2] / (iters * 64.0) / 2.0;     // two waves share a SIMD
    printf("%-16s SIMD cycles per wave64 instruction %.2f%s\n", name, c, ref > 0 ? "" : "  (reference)");
    if (ref > 0) printf("%-16s   = %.2f x v_fma_f64\n", "", c / ref);
    (void)hipFree(out); (void)hipFree(st);
    return c;
}

int main() {
    const double r = run<0>("v_fma_f64", 0);
    run<8>("v_mul_f64", r); run<9>("v_add_f64", r);
    run<1>("v_rndne_f64", r); run<2>("v_fract_f64", r); run<3>("v_cvt_i32_f64", r); run<4>("v_ldexp_f64", r);
    run<5>("v_and_b32", r); run<6>("v_lshlrev_b32", r); run<7>("v_ashrrev_i32", r); run<10>("v_lshl_add_u32", r);
    run<11>("v_max_i32", r); run<12>("v_mov_b64", r);
    run<13>("v_add_u32", r); run<14>("v_add_u32 lit", r); run<15>("v_mul_u32_u24", r); run<16>("v_bfe_u32", r);
    run<17>("v_xor_b32", r); run<18>("v_lshlrev_b32 v", r); run<19>("v_lshrrev_b32", r); run<20>("v_and_or_b32", r);
    run<21>("v_sub_u32", r);
    return 0;
}
