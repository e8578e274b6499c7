// One launch per horizon step for SMALL batches (the B = 1 callbacks of a solver loop, src/mpc.py:202-255, which Ipopt
// issues strictly one after the other): the per-step "head" kernel (finish step t-1, prepare step t) and the staged pair
// kernel (pair_kernel.h) fused into a single grid, so that a rollout is H + 1 launches instead of 2H + 1.
//
// Why fusion by REDUNDANCY and not by a last-arriver hand-off: at these sizes every kernel is a chain of dependent
// latencies (measured on MI355X at N = 512, B = 1: head 7.2 us, pair 6.8 us, boundary 1.9 us, all of it waiting), and an
// in-kernel hand-off (release fence + ticket + acquire fence) costs what a kernel boundary costs (MI355X: ~1.7 us per
// fence).  Instead every workgroup of launch t recomputes the cheap part of the old head itself -- the Z0 sums of step
// t-1, hence the input variances of step t -- from a compact array the previous launch wrote, and the rest of the head
// (mean sums over the N points, Jacobian rows of step t-1) runs in ds extra workgroups of the SAME launch, beside the
// tiles instead of before them.  Nothing inside a launch depends on another workgroup of that launch.
//
//   grid = (nwork + ds, B) x 256 threads
//   blockIdx.x <  nwork : tile (unit a, 64 rows x 64|128 columns) of the N^2 sum of step t      (pair_kernel.h, staged form)
//   blockIdx.x >= nwork : GP a: finish step t-1 (means / vars / Jacobian rows), mean sums of step t (step.hip::prep_step)
//
// Reference restated: Dynamics.forward_propagate_torch (src/dynamics.py:145-189), mean_prop_torch / variance_prop_torch
// (src/tools/uncertainty_prop.py:296-399), exactly as step.hip / pair_kernel.h do; the closed forms are in step.hip.
// State between launches (double-buffered by step parity): sp [2][B][ds][sps], part [2][B][nwork][nm], partz [2][B][nwork].
#pragma once
#include "gpmpc_internal.h"
#include "fast_exp.h"

struct FusedArgs {
    // pack
    const double* XT; const double* beta; const double* lam; const double* sf; const double* M; const int* work;
    int N, Np, nwork;
    int ustart[GPMPC_MAX_DS + 1];          // items of GP a are [ustart[a], ustart[a+1]) (64-row work lists are unit-contiguous)
    // problem
    const double* x0; const double* U; int B, H;
    // outputs / state
    double* means; double* vars; double* jac;
    double* sp; double* part; double* partz;
    int sps, nm;
};

// layout of sp (doubles), as step.hip: 0 c | 1 mu | 2 sf2 | 3 A[D] | 3+D scale[D] | 3+2D dmu_du[D] | 3+3D dmu_ds[D]
template <int D, int NS2, bool GRAD>
__global__ __launch_bounds__(256) void k_step_fused(FusedArgs A, int t) {
    constexpr int DS = NS2, DA = D - NS2, NM = GRAD ? 1 + 2 * D : 1, DP = (D + 1) & ~1;
    constexpr int NV = 1 + 2 * D;
    __shared__ double s_tab[GPMPC_EXP_N];
    __shared__ __attribute__((aligned(16))) double s_hj[64 * DP];
    __shared__ double s_red[16 * NV > 4 * NM ? 16 * NV : 4 * NM];
    __shared__ double s_out[NV];
    __shared__ double s_z4[GPMPC_MAX_DS * 4];
    __shared__ double s_mu[GPMPC_MAX_DS], s_var[GPMPC_MAX_DS], s_z0[GPMPC_MAX_DS], s_c[GPMPC_MAX_DS], s_sf2[GPMPC_MAX_DS];
    __shared__ double s_lam[GPMPC_MAX_DS * D], s_uact[DA > 0 ? DA : 1], s_spp[4 * D];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_mean = (int)blockIdx.x >= A.nwork;
    const int Np = A.Np, pprev = (t - 1) & 1, pcur = t & 1;
    const int am = (int)blockIdx.x - A.nwork;                      // GP of a mean workgroup

    // ---- phase 0: every load that depends on nothing computed in this launch is issued first ----------------------
    int unit = 0, i0 = 0, j0 = 0, j1 = 0;
    double xrow[D], xcol[D], mpre[16];
    if (!is_mean) {
        const int4 wk = reinterpret_cast<const int4*>(A.work)[blockIdx.x];
        unit = wk.x; i0 = wk.y; j0 = wk.z; j1 = wk.w;
        gpmpc_exp_table_to_lds(s_tab);
#pragma unroll
        for (int k = 0; k < D; ++k) xrow[k] = A.XT[(size_t)k * Np + i0 + lane];      // the 4 waves share the tile's 64 rows
        const int jfirst = (j0 + 63 < i0) ? j0 + 64 : j0;                             // first column chunk that carries weight
        if (tid < 64) {
#pragma unroll
            for (int k = 0; k < D; ++k) xcol[k] = A.XT[(size_t)k * Np + jfirst + tid];
        }
        const double* Mc = A.M + (size_t)unit * Np * Np + (size_t)jfirst * Np + i0 + lane;
#pragma unroll
        for (int q = 0; q < 16; ++q) mpre[q] = Mc[(size_t)(w * 16 + q) * Np];
    }
    if (tid < DS * D) s_lam[tid] = A.lam[tid];
    if (DA > 0 && tid < DA) s_uact[tid] = A.U[((size_t)b * A.H + (t - 1)) * DA + tid];
    double sp_c = 0.0, sp_mu = 0.0, sp_sf2 = 0.0;
    if (t > 1 && tid < DS) {
        const double* sp = A.sp + (((size_t)pprev * A.B + b) * DS + tid) * A.sps;
        sp_c = sp[0]; sp_mu = sp[1]; sp_sf2 = sp[2];
    }
    if (is_mean && t > 1 && GRAD && tid < 4 * D)                                     // A, scale, dmu_du, dmu_ds of step t-1
        s_spp[tid] = A.sp[(((size_t)pprev * A.B + b) * DS + am) * A.sps + 3 + tid];

    // ---- phase 1: mean and variance of step t-1 for ALL GPs, identically in every workgroup ------------------------
    if (t == 1) {
        if (tid < DS) { s_mu[tid] = A.x0[(size_t)b * DS + tid]; s_var[tid] = GPMPC_INIT_VAR; }
    } else {
        const double* pz = A.partz + ((size_t)pprev * A.B + b) * A.nwork;
#pragma unroll
        for (int a = 0; a < DS; ++a) {
            double s = 0.0;
            for (int wi = A.ustart[a] + tid; wi < A.ustart[a + 1]; wi += 256) s += pz[wi];
            s = wave_sum(s);
            if (lane == 0) s_z4[a * 4 + w] = s;
        }
        __syncthreads();
        if (tid < DS) {
            const double z0 = (s_z4[tid * 4] + s_z4[tid * 4 + 1]) + (s_z4[tid * 4 + 2] + s_z4[tid * 4 + 3]);
            s_mu[tid] = sp_mu; s_z0[tid] = z0; s_c[tid] = sp_c; s_sf2[tid] = sp_sf2;
            s_var[tid] = sp_sf2 - sp_c * z0 - sp_mu * sp_mu;          // no clamp (src/tools/uncertainty_prop.py:399)
        }
    }
    __syncthreads();

    if (!is_mean) {
        // ---- tile of the N^2 sum of step t (staged form of pair_kernel.h: diagonal S, one trajectory, column split) ----
        const int a = unit;
        double hi[D], sck[D], cv[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const double uk = k < DS ? s_mu[k < DS ? k : 0] : s_uact[k >= DS ? k - DS : 0];
            const double sk = k < DS ? s_var[k < DS ? k : 0] : GPMPC_ACTION_VAR;
            sck[k] = sqrt(0.125 / (0.5 * s_lam[a * D + k] + sk));     // same expression as the mean workgroup stores in sp
            cv[k] = sck[k] * uk;
            hi[k] = fma(-sck[k], xrow[k], cv[k]);
        }
        double acc[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m] = 0.0;
        const double* __restrict__ Ma = A.M + (size_t)unit * Np * Np;
        bool first = true;
        for (int jc = j0; jc < j1; jc += 64) {
            if (jc + 63 < i0) continue;                               // upper-triangular M: nothing left of the diagonal chunk
            __syncthreads();
            if (tid < 64) {
                double x[D];
#pragma unroll
                for (int k = 0; k < D; ++k) x[k] = first ? xcol[k] : A.XT[(size_t)k * Np + jc + tid];
#pragma unroll
                for (int k = 0; k < D; ++k) s_hj[tid * DP + k] = fma(-sck[k], x[k], cv[k]);
            }
            if (!first) {                                             // the first chunk's weights were fetched in phase 0
                const double* __restrict__ Mc = Ma + (size_t)jc * Np + i0 + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) mpre[q] = Mc[(size_t)(w * 16 + q) * Np];
            }
            first = false;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const double* hj = &s_hj[(w * 16 + q) * DP];
                double m[D], sq[D];
#pragma unroll
                for (int k = 0; k < D; ++k) { m[k] = hi[k] + hj[k]; sq[k] = m[k] * m[k]; }
                double s = sq[0];
#pragma unroll
                for (int k = 1; k < D; ++k) s += sq[k];
                const double P = mpre[q] * gpmpc_exp_neg(s, s_tab);
                acc[0] += P;
                if (GRAD) {
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        acc[GRAD ? 1 + k : 0] = fma(P, m[k], acc[GRAD ? 1 + k : 0]);
                        if (k < NS2) acc[GRAD ? 1 + D + k : 0] = fma(P, sq[k], acc[GRAD ? 1 + D + k : 0]);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double s = wave_sum(acc[m]);
            if (lane == 0) s_red[w * NM + m] = s;
        }
        __syncthreads();
        if (tid < NM) {
            const double s = (s_red[tid] + s_red[NM + tid]) + (s_red[2 * NM + tid] + s_red[3 * NM + tid]);
            A.part[(((size_t)pcur * A.B + b) * A.nwork + blockIdx.x) * A.nm + tid] = s;
            if (tid == 0) A.partz[((size_t)pcur * A.B + b) * A.nwork + blockIdx.x] = s;
        }
        return;
    }

    // ---- GP `am`: finish step t-1 (outputs + Jacobian rows), then the mean sums of step t ----------------------------
    const int a = am;
    if (t == 1) {
        if (a == 0 && tid < DS) {
            A.means[((size_t)b * (A.H + 1)) * DS + tid] = s_mu[tid];
            A.vars[((size_t)b * (A.H + 1)) * DS + tid] = GPMPC_INIT_VAR;
        }
    } else {
        if (GRAD) {
            // moments 1..NM-1 of GP a: thread = (moment, channel), 16 channels stride the items; fixed-order combine
            const int ch = tid & 15;
            for (int m = tid >> 4; m < NM; m += 16) {
                const double* p = A.part + (((size_t)pprev * A.B + b) * A.nwork) * A.nm + m;
                double s = 0.0;
                for (int wi = A.ustart[a] + ch; wi < A.ustart[a + 1]; wi += 16) s += p[(size_t)wi * A.nm];
                s_red[m * 16 + ch] = s;
            }
            __syncthreads();
            if (tid < NM) {
                double s = 0.0;
                for (int c = 0; c < 16; ++c) s += s_red[tid * 16 + c];
                s_out[tid] = s;
            }
            __syncthreads();
        }
        if (tid == 0) {
            const double mu = s_mu[a], var = s_var[a];
            A.means[((size_t)b * (A.H + 1) + (t - 1)) * DS + a] = mu;
            A.vars[((size_t)b * (A.H + 1) + (t - 1)) * DS + a] = var;
        }
        if (GRAD && tid < D) {
            const int k = tid, nc = 2 * DS + DA;
            const double c = s_c[a], mu = s_mu[a], T = c * s_z0[a];  // the Z0 sum every workgroup of this launch uses
            const double Ak = s_spp[k], sc = s_spp[D + k], dmu_du = s_spp[2 * D + k], dmu_ds = s_spp[3 * D + k];
            const double dT_du = -4.0 * sc * c * s_out[1 + k];
            const double dv_du = -dT_du - 2.0 * mu * dmu_du;
            double* jm = A.jac + (((size_t)b * A.H + (t - 2)) * 2 * DS + a) * nc;          // row of mu_a (step t-1)
            double* jv = A.jac + (((size_t)b * A.H + (t - 2)) * 2 * DS + DS + a) * nc;     // row of var_a
            if (k < DS) {
                const double dT_ds = Ak * (c * s_out[1 + D + (k < NS2 ? k : 0)] - 0.5 * T);
                const double dv_ds = -dT_ds - 2.0 * mu * dmu_ds;
                jm[k] = dmu_du; jm[DS + k] = dmu_ds;
                jv[k] = dv_du;  jv[DS + k] = dv_ds;
            } else {
                jm[2 * DS + (k - DS)] = dmu_du;
                jv[2 * DS + (k - DS)] = dv_du;
            }
        }
        __syncthreads();
    }
    // mean sums of step t for GP a (step.hip::prep_step without the pair-kernel parameters)
    __shared__ double s_B[D], s_A[D], s_sc[D], s_r1[D], s_r2[D], s_u[D];
    if (tid < D) {
        const int k = tid;
        const double uk = k < DS ? s_mu[k < DS ? k : 0] : s_uact[k >= DS ? k - DS : 0];
        const double sk = k < DS ? s_var[k < DS ? k : 0] : GPMPC_ACTION_VAR;
        const double lam = s_lam[a * D + k];
        s_u[k] = uk;
        s_B[k] = 1.0 / (sk + lam);
        s_A[k] = 1.0 / (0.5 * lam + sk);
        s_sc[k] = sqrt(0.125 / (0.5 * lam + sk));
        s_r1[k] = sk / lam + 1.0;
        s_r2[k] = 2.0 * sk / lam + 1.0;
    }
    __syncthreads();
    double u[D], Bk[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { u[k] = s_u[k]; Bk[k] = s_B[k]; }
    double v[NV];
#pragma unroll
    for (int m = 0; m < NV; ++m) v[m] = 0.0;
    for (int i = tid; i < Np; i += 256) {
        double d[D], q = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { d[k] = u[k] - A.XT[(size_t)k * Np + i]; q = fma(Bk[k] * d[k], d[k], q); }
        const double p = A.beta[(size_t)a * Np + i] * exp(-0.5 * q);
        v[0] += p;
#pragma unroll
        for (int k = 0; k < D; ++k) { v[1 + k] = fma(p, d[k], v[1 + k]); v[1 + D + k] = fma(p * d[k], d[k], v[1 + D + k]); }
    }
    block_sum<NV>(v, s_red, s_out);
    if (tid < D) {
        const int k = tid;
        const double sf = A.sf[a], sf2 = sf * sf;
        double detm = 1.0, detv = 1.0;
        for (int l = 0; l < D; ++l) { detm *= s_r1[l]; detv *= s_r2[l]; }
        const double cm = sf2 / sqrt(detm), c = 1.0 / sqrt(detv);
        const double mu = cm * s_out[0];
        double* sp = A.sp + (((size_t)pcur * A.B + b) * DS + a) * A.sps;
        if (k == 0) { sp[0] = c; sp[1] = mu; sp[2] = sf2; }
        const double Bq = s_B[k];
        sp[3 + k] = s_A[k]; sp[3 + D + k] = s_sc[k];
        sp[3 + 2 * D + k] = -Bq * cm * s_out[1 + k];
        sp[3 + 3 * D + k] = -0.5 * mu * Bq + 0.5 * Bq * Bq * cm * s_out[1 + D + k];
    }
}

template <int D, int NS2, bool GRAD>
static int launch_step_fused_one(const FusedArgs& a, int t, hipStream_t s) {
    hipLaunchKernelGGL((k_step_fused<D, NS2, GRAD>), dim3(a.nwork + NS2, a.B), dim3(256), 0, s, a, t);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { gpmpc_set_error("fused step kernel launch", e); return GPMPC_E_LAUNCH; }
    return GPMPC_OK;
}

// ns2 = state_dim: D - ns2 in {0, 1, 2} action dimensions
template <int D>
int gpmpc_launch_step_fused_D(bool grad, int ns2, const FusedArgs& a, int t, hipStream_t s) {
    if (a.nm != (grad ? 1 + 2 * D : 1)) return GPMPC_E_ARG;
#define GPMPC_FUSED_CASE(GR)                                                                                       \
    if (grad == GR) {                                                                                              \
        if (ns2 == D) return launch_step_fused_one<D, D, GR>(a, t, s);                                             \
        if (D >= 2 && ns2 == D - 1) return launch_step_fused_one<D, (D >= 2 ? D - 1 : D), GR>(a, t, s);            \
        if (D >= 3 && ns2 == D - 2) return launch_step_fused_one<D, (D >= 3 ? D - 2 : D), GR>(a, t, s);            \
        return GPMPC_E_ARG;                                                                                        \
    }
    GPMPC_FUSED_CASE(true)
    GPMPC_FUSED_CASE(false)
#undef GPMPC_FUSED_CASE
    return GPMPC_E_ARG;
}
