// Microbenchmark (MI355X), round 5: three numbers the whole-horizon kernel's column loop (csrc/traj_persist.h) hinges on.
//   1  does a wave64 fp64 VALU instruction get cheaper when part of EXEC is off?  (the diagonal 64 x 64 blocks of the pair
//      triangle run with half their lanes idle on average: 17 % of all lane slots at N = 300)
//   2  latency of a dependent scalar load (s_load_dwordx2 pointer chase): scalar-cache hit, L2 hit on lines written by vector stores
//      a moment ago (what the G rows are), and a 256 MB footprint
//   3  latency of a dependent LDS read (the exp table look-up)
//   hipcc -O3 --offload-arch=gfx950 latency_probe.hip -o latency_probe && ./latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int MODE>
__global__ __launch_bounds__(256) void k_exec(double* out, unsigned long long* stamps, int iters) {
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = 1.0 + 0.37 * i + threadIdx.x * 1e-3;
    const double a = 0.999999, b = 1e-7;
    const int lane = threadIdx.x & 63;
    bool on = true;
    if (MODE == 1) on = lane < 32;
    if (MODE == 2) on = lane < 16;
    if (MODE == 3) on = (lane & 1) == 0;
    if (MODE == 4) on = lane < 48;
    if (MODE == 5) on = lane >= 32;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(a), "v"(b));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
static void run_exec(const char* name) {
    const int threads = 256, blocks = 512, iters = 2000, waves = blocks * 4;
    double* out; unsigned long long* st;
    (void)hipMalloc(&out, sizeof(double) * blocks * threads);
    (void)hipMalloc(&st, sizeof(unsigned long long) * waves);
    hipLaunchKernelGGL((k_exec<MODE>), dim3(blocks), dim3(threads), 0, 0, out, st, 50);
    hipLaunchKernelGGL((k_exec<MODE>), dim3(blocks), dim3(threads), 0, 0, out, st, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("v_fma_f64, EXEC = %-22s SIMD cycles per wave64 instruction %.2f\n", name, (double)h[waves / 2] / (iters * 64.0) / 2.0);
    (void)hipFree(out); (void)hipFree(st);
}

// pointer chase through scalar loads: buf[i] holds the byte offset of the next element
__global__ __launch_bounds__(64) void k_fill(unsigned long long* buf, size_t n, size_t stride_elems) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i * stride_elems] = ((i + 1) % n) * stride_elems * 8;
}
__global__ __launch_bounds__(64) void k_chase(const unsigned long long* buf, int hops, int inv, unsigned long long* res) {
    typedef const unsigned long long __attribute__((address_space(4))) cu64;
    unsigned long long off = 0;
    if (inv) __builtin_amdgcn_s_dcache_inv();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int h = 0; h < hops; ++h) {
        cu64* p = (cu64*)((const char*)buf + off);
        unsigned long long v;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
        off = v;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { res[0] = t1 - t0; res[1] = off; }
}
static void run_chase(const char* name, size_t n, size_t stride_elems, int hops, int warm, int inv) {
    unsigned long long *buf, *res;
    (void)hipMalloc(&buf, n * stride_elems * 8);
    (void)hipMalloc(&res, 16);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, buf, n, stride_elems);
    (void)hipDeviceSynchronize();
    for (int w = 0; w < warm; ++w) hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, buf, hops, 0, res);
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, buf, hops, inv, res);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
    printf("s_load chase, %-44s %.0f cycles per dependent load\n", name, (double)h[0] / hops);
    (void)hipFree(buf); (void)hipFree(res);
}

__global__ __launch_bounds__(64) void k_lds(int hops, unsigned long long* res) {
    __shared__ int s[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) s[i] = (i * 37 + 11) & 2047;
    __syncthreads();
    int idx = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int h = 0; h < hops; ++h) idx = s[idx];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    res[2 + threadIdx.x] = idx;
    if (threadIdx.x == 0) res[0] = t1 - t0;
}

int main() {
    run_exec<0>("all 64 lanes"); run_exec<1>("lanes 0-31"); run_exec<5>("lanes 32-63"); run_exec<2>("lanes 0-15"); run_exec<4>("lanes 0-47"); run_exec<3>("even lanes");
    run_chase("16 elements, 64 B apart (scalar-cache hits)", 16, 8, 4000, 2, 0);
    run_chase("4096 lines, 64 B apart, fresh (L2 hits)", 4096, 8, 4096, 0, 1);
    run_chase("4096 lines, second pass (256 KB > scalar cache)", 4096, 8, 4096, 2, 0);
    run_chase("64 K lines, 4 KB apart (256 MB: HBM / IC)", 65536, 512, 4096, 0, 1);
    unsigned long long* res; (void)hipMalloc(&res, 8 * 80);
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, 4000, res);
    (void)hipDeviceSynchronize();
    unsigned long long h; (void)hipMemcpy(&h, res, 8, hipMemcpyDeviceToHost);
    printf("dependent ds_read_b32 chain: %.0f cycles per read\n", (double)h / 4000);
    return 0;
}
