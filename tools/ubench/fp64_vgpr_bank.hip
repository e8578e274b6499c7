// Microbenchmark (MI355X): does the VGPR bank placement of a wave64 v_fmac_f64's operands change its issue cost?
// 16 independent accumulators acc_k += s * P with the accumulators' register pairs starting at bank 0 (v[4k]) or at
// bank 2 (v[4k+2]) while P sits at bank 0: same-bank vs different-bank reads of the two 64-bit VGPR operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>


template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* stamps, int iters, double seed) {
    const double sv = seed * 1.0000001;
    unsigned long long t0 = 0, t1 = 0;
    double r = 0;
    if (MODE == 0) {      // accumulators v[64:65], v[68:69], ... (bank 0,1); P = v[60:61] (bank 0,1)
        asm volatile(
            "v_mov_b32 v60, %[pl]\n v_mov_b32 v61, %[ph]\n"
            "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n"
            "v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v88, 0\n v_mov_b32 v89, 0\n v_mov_b32 v92, 0\n v_mov_b32 v93, 0\n"
            "s_memtime %[t0]\n s_waitcnt lgkmcnt(0)\n"
            "1:\n"
            "v_fmac_f64 v[64:65], %[s], v[60:61]\n v_fmac_f64 v[68:69], %[s], v[60:61]\n v_fmac_f64 v[72:73], %[s], v[60:61]\n v_fmac_f64 v[76:77], %[s], v[60:61]\n"
            "v_fmac_f64 v[80:81], %[s], v[60:61]\n v_fmac_f64 v[84:85], %[s], v[60:61]\n v_fmac_f64 v[88:89], %[s], v[60:61]\n v_fmac_f64 v[92:93], %[s], v[60:61]\n"
            "v_fmac_f64 v[64:65], %[s], v[60:61]\n v_fmac_f64 v[68:69], %[s], v[60:61]\n v_fmac_f64 v[72:73], %[s], v[60:61]\n v_fmac_f64 v[76:77], %[s], v[60:61]\n"
            "v_fmac_f64 v[80:81], %[s], v[60:61]\n v_fmac_f64 v[84:85], %[s], v[60:61]\n v_fmac_f64 v[88:89], %[s], v[60:61]\n v_fmac_f64 v[92:93], %[s], v[60:61]\n"
            "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
            "s_memtime %[t1]\n s_waitcnt lgkmcnt(0)\n"
            "v_add_f64 %[r], v[64:65], v[92:93]\n"
            : [t0] "=s"(t0), [t1] "=s"(t1), [r] "=v"(r), [n] "+s"(iters)
            : [s] "s"(sv), [pl] "v"(__double2loint(seed + threadIdx.x)), [ph] "v"(__double2hiint(seed + threadIdx.x))
            : "v60", "v61", "v64", "v65", "v68", "v69", "v72", "v73", "v76", "v77", "v80", "v81", "v84", "v85", "v88", "v89", "v92", "v93", "scc", "memory");
    } else {              // accumulators v[66:67], v[70:71], ... (bank 2,3); P = v[60:61] (bank 0,1)
        asm volatile(
            "v_mov_b32 v60, %[pl]\n v_mov_b32 v61, %[ph]\n"
            "v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n"
            "v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n v_mov_b32 v90, 0\n v_mov_b32 v91, 0\n v_mov_b32 v94, 0\n v_mov_b32 v95, 0\n"
            "s_memtime %[t0]\n s_waitcnt lgkmcnt(0)\n"
            "1:\n"
            "v_fmac_f64 v[66:67], %[s], v[60:61]\n v_fmac_f64 v[70:71], %[s], v[60:61]\n v_fmac_f64 v[74:75], %[s], v[60:61]\n v_fmac_f64 v[78:79], %[s], v[60:61]\n"
            "v_fmac_f64 v[82:83], %[s], v[60:61]\n v_fmac_f64 v[86:87], %[s], v[60:61]\n v_fmac_f64 v[90:91], %[s], v[60:61]\n v_fmac_f64 v[94:95], %[s], v[60:61]\n"
            "v_fmac_f64 v[66:67], %[s], v[60:61]\n v_fmac_f64 v[70:71], %[s], v[60:61]\n v_fmac_f64 v[74:75], %[s], v[60:61]\n v_fmac_f64 v[78:79], %[s], v[60:61]\n"
            "v_fmac_f64 v[82:83], %[s], v[60:61]\n v_fmac_f64 v[86:87], %[s], v[60:61]\n v_fmac_f64 v[90:91], %[s], v[60:61]\n v_fmac_f64 v[94:95], %[s], v[60:61]\n"
            "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
            "s_memtime %[t1]\n s_waitcnt lgkmcnt(0)\n"
            "v_add_f64 %[r], v[66:67], v[94:95]\n"
            : [t0] "=s"(t0), [t1] "=s"(t1), [r] "=v"(r), [n] "+s"(iters)
            : [s] "s"(sv), [pl] "v"(__double2loint(seed + threadIdx.x)), [ph] "v"(__double2hiint(seed + threadIdx.x))
            : "v60", "v61", "v66", "v67", "v70", "v71", "v74", "v75", "v78", "v79", "v82", "v83", "v86", "v87", "v90", "v91", "v94", "v95", "scc", "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, int wps) {
    const int threads = 256, blocks = 256 * wps, iters = 20000, waves = blocks * 4;
    double* out; unsigned long long* st;
    (void)hipMalloc(&out, sizeof(double) * blocks * threads);
    (void)hipMalloc(&st, sizeof(unsigned long long) * waves);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, st, 100, 1.0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, st, iters, 1.0);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-28s waves/SIMD=%d  median cycles per wave64 v_fmac_f64 (per SIMD): %.3f\n", name, wps,
           (double)h[waves / 2] / (iters * 16.0) / wps);
    (void)hipFree(out); (void)hipFree(st);
}

int main() {
    for (int wps : {1, 2}) {
        run<0>("acc bank 0/1, P bank 0/1", wps);
        run<1>("acc bank 2/3, P bank 0/1", wps);
    }
    return 0;
}
