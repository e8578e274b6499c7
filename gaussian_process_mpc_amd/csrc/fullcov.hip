// Full-covariance rollout (BASELINE config 5): the state distribution carries the full ds x ds covariance, with
// off-diagonal terms from the exact cross-covariance Cov[f_a, f_b] (covariance_prop_torch,
// src/tools/uncertainty_prop.py:402-465, consistent form).  The reference's rollout propagates variances only
// (src/dynamics.py:184-189: "So far, only paying attention to mean and variance. Implement covariance.") -- this is
// the build's extension of Dynamics.forward_propagate_torch + RiskSensitiveMPC.objective/gradient to full Sigma,
// oracled by composing the reference's own single-step functions (oracle/gpmpc_oracle.py).
//
// Per step: k_fc_assemble builds (u_t, S_t) = ([mu_{t-1}; U_{t-1}], blkdiag(Sigma_{t-1}, float32(1e-3) I)) for every
// trajectory, gpmpc_moment_match evaluates means, full covariance and their Jacobians (one pair-kernel launch over the
// ds variance units and the ds(ds-1)/2 cross units), the Jacobians of all steps are kept, and k_fc_tail evaluates the
// risk-sensitive cost with full Sigma (src/mpc.py:179-198) and runs the reverse sweep.
#include <cstdio>
#include "gpmpc_internal.h"
#include "moment_dev.h"

int gpmpc_timed_pair_sbf(int D, bool grad, int ns2, int waves, const PairSbfArgs& a, hipStream_t s);
int gpmpc_moment_match_ex(const gpmpc_pack* p, int nq, const double* u, const double* S, unsigned flags,
                          double* out_mean, double* out_var, double* out_cov, double* out_l, double* dmean_du,
                          double* dmean_dS, double* dvar_du, double* dvar_dS, double* dcov_du, double* dcov_dS,
                          void* workspace, size_t workspace_bytes, void* stream, int ns2);

struct FcArgs {
    int B, H, ds, da, D, grad;
    const double* x0; const double* U;
    double* u; double* S;                 // [B][D], [B][D][D]   inputs of the current step
    double* mean; double* cov;            // [B][ds], [B][ds][ds] outputs of the current step
    double* out_means; double* out_covs;  // [B][H+1][ds], [B][H+1][ds][ds]
    // Jacobians of every step t = 1..H at slice t-1: [H][B][...]
    double* dmean_du; double* dmean_dS; double* dcov_du; double* dcov_dS;
    double* out_cost; double* out_grad;
    gpmpc_cost_params cost;
};

// step t >= 1: record the state of step t-1 and assemble the input distribution of step t
__global__ void k_fc_assemble(FcArgs A, int t) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B) return;
    const int ds = A.ds, da = A.da, D = A.D, H = A.H;
    double* om = A.out_means + ((size_t)b * (H + 1) + (t - 1)) * ds;
    double* oc = A.out_covs + ((size_t)b * (H + 1) + (t - 1)) * ds * ds;
    double* u = A.u + (size_t)b * D;
    double* S = A.S + (size_t)b * D * D;
    for (int k = 0; k < ds; ++k) {
        const double m = (t == 1) ? A.x0[(size_t)b * ds + k] : A.mean[(size_t)b * ds + k];
        om[k] = m; u[k] = m;
        for (int l = 0; l < ds; ++l) {
            const double c = (t == 1) ? (k == l ? GPMPC_INIT_VAR : 0.0) : A.cov[((size_t)b * ds + k) * ds + l];
            oc[k * ds + l] = c;
            S[k * D + l] = c;
        }
        for (int l = ds; l < D; ++l) { S[k * D + l] = 0.0; S[l * D + k] = 0.0; }
    }
    if (t > H) return;                        // final call: only records step H
    for (int k = 0; k < da; ++k) {
        u[ds + k] = A.U[((size_t)b * H + (t - 1)) * da + k];
        for (int l = 0; l < da; ++l) S[(ds + k) * D + ds + l] = (k == l) ? GPMPC_ACTION_VAR : 0.0;
    }
}

// State cost with a full covariance and its derivatives (src/mpc.py:182-185):
//   (1/gamma) log det(I + gamma Q Sig) + e^T Z e,  Z = (I + gamma Q Sig)^-1 Q = (Q^-1 + gamma Sig)^-1
//   d/dmu = (Z + Z^T) e,   d/dSig = sym( Z^T - gamma (Z^T e)(Z e)^T )         (gamma = 0: Q^T, tr(Q Sig) + e^T Q e)
// One lane per horizon step; the state dimension is a template parameter so that the ds x 2ds elimination lives in registers (as
// runtime-indexed LDS scratch the H + 1 terms were ~70 us of dependent LDS round trips at ds = 4).  Partial pivoting by
// compare-and-swap of whole rows (no runtime-indexed row).
template <int DS>
__device__ static double fc_state_cost(const gpmpc_cost_params& C, const double* __restrict__ mu, const double* __restrict__ Sig,
                                       double* dmu, double* dSig) {
    const double g = C.gamma;
    double sg[DS * DS], e[DS], w[DS][2 * DS];
#pragma unroll
    for (int k = 0; k < DS * DS; ++k) sg[k] = Sig[k];
#pragma unroll
    for (int k = 0; k < DS; ++k) e[k] = mu[k] - C.x_ref[k];
    double det = 1.0;
#pragma unroll
    for (int r = 0; r < DS; ++r)
#pragma unroll
        for (int c = 0; c < DS; ++c) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < DS; ++l) s += C.Q[r * DS + l] * sg[l * DS + c];
            w[r][c] = (r == c ? 1.0 : 0.0) + g * s;
            w[r][DS + c] = C.Q[r * DS + c];
        }
    if (g != 0.0) {
#pragma unroll
        for (int k = 0; k < DS; ++k) {
#pragma unroll
            for (int r = k + 1; r < DS; ++r) {
                if (fabs(w[r][k]) > fabs(w[k][k])) {
#pragma unroll
                    for (int c = 0; c < 2 * DS; ++c) { const double tmp = w[k][c]; w[k][c] = w[r][c]; w[r][c] = tmp; }
                    det = -det;
                }
            }
            const double pv = w[k][k];
            det *= pv;
            const double inv = 1.0 / pv;
#pragma unroll
            for (int c = 0; c < 2 * DS; ++c) w[k][c] *= inv;
#pragma unroll
            for (int r = 0; r < DS; ++r) {
                if (r == k) continue;
                const double f = w[r][k];
#pragma unroll
                for (int c = 0; c < 2 * DS; ++c) w[r][c] = fma(-f, w[k][c], w[r][c]);
            }
        }
    }
    double ze[DS], zte[DS], quad = 0.0, trq = 0.0;
#pragma unroll
    for (int k = 0; k < DS; ++k) {
        double s = 0.0, st = 0.0;
#pragma unroll
        for (int l = 0; l < DS; ++l) { s += w[k][DS + l] * e[l]; st += w[l][DS + k] * e[l]; trq += C.Q[k * DS + l] * sg[l * DS + k]; }
        ze[k] = s; zte[k] = st;
        quad += e[k] * s;
    }
    if (dmu) {
#pragma unroll
        for (int k = 0; k < DS; ++k) dmu[k] = ze[k] + zte[k];
#pragma unroll
        for (int k = 0; k < DS; ++k)
#pragma unroll
            for (int l = 0; l < DS; ++l) {
                const double gkl = w[l][DS + k] - g * zte[k] * ze[l];     // Z^T - gamma (Z^T e)(Z e)^T
                const double glk = w[k][DS + l] - g * zte[l] * ze[k];
                dSig[k * DS + l] = 0.5 * (gkl + glk);
            }
    }
    return (g == 0.0 ? trq : log(det) / g) + quad;
}

__device__ static double fc_input_cost(int H, int da, const gpmpc_cost_params& C, const double* U, double* gU) {
    double c = 0.0;
    for (int j = 0; j < H; ++j) {
        double d[GPMPC_MAX_D];
        for (int k = 0; k < da; ++k) d[k] = U[j * da + k] - C.u_ref[k];
        for (int k = 0; k < da; ++k) {
            double rd = 0.0, rtd = 0.0;
            for (int l = 0; l < da; ++l) { rd += C.R[k * da + l] * d[l]; rtd += C.R[l * da + k] * d[l]; }
            c += d[k] * rd;
            if (gU) gU[j * da + k] += rd + rtd;
        }
        if (C.has_R_delta) {
            for (int k = 0; k < da; ++k) d[k] = U[j * da + k] - (j == 0 ? C.last_u[k] : U[(j - 1) * da + k]);
            for (int k = 0; k < da; ++k) {
                double rd = 0.0, rtd = 0.0;
                for (int l = 0; l < da; ++l) { rd += C.R_delta[k * da + l] * d[l]; rtd += C.R_delta[l * da + k] * d[l]; }
                c += d[k] * rd;
                if (gU) { gU[j * da + k] += rd + rtd; if (j > 0) gU[(j - 1) * da + k] -= rd + rtd; }
            }
        }
    }
    return c;
}

#define GPMPC_FC_WORKERS 32
#define GPMPC_FC_TAIL_TERMS ((GPMPC_MAX_DS + GPMPC_MAX_DS * GPMPC_MAX_DS + 3) / 4)
// One workgroup (256 threads) per trajectory.  dynamic LDS:
//   [WORKERS][ds*2ds] LU scratch | [H+1] cost terms | [H+1][ds + ds*ds] local derivatives | adjoint (2 x (ds + ds*ds)) | g (D + D*D) |
//   U [H*da] | dJ/dU [H*da]
// (the inputs and their gradient live in LDS: as global read-modify-writes of one lane the input cost was 20 dependent round trips.)
// Reverse sweep (round 4): the adjoint of (mu_t, Sigma_t) is pulled through step t's Jacobians by FOUR lanes per needed input entry
// (the D entries of u, the ds x ds state block of S; the action block of S is constant), each lane with its quarter of the
// ds + ds^2 terms, the Jacobian values of step t-1 loaded while step t is summed: the sweep used to be 64 lanes x 30 entries x 20
// dependent global round trips (230 us at H = 20, a tenth of a B = 1 rollout).
template <int DS>
__global__ __launch_bounds__(256) void k_fc_tail(FcArgs A) {
    extern __shared__ double s_dyn[];
    constexpr int ds = DS;
    const int b = blockIdx.x, da = A.da, D = A.D, H = A.H, nz = ds + ds * ds, tid = threadIdx.x;
    double* s_lu = s_dyn;
    double* s_ct = s_lu + GPMPC_FC_WORKERS * ds * 2 * ds;
    double* s_dl = s_ct + (H + 1);
    double* s_adj = s_dl + (size_t)(H + 1) * nz;
    double* s_g = s_adj + 2 * nz;
    double* s_U = s_g + D + D * D;
    double* s_gU = s_U + H * da;
    for (int r = tid; r < H * da; r += blockDim.x) { s_U[r] = A.U[(size_t)b * H * da + r]; s_gU[r] = 0.0; }
    const double* mu = A.out_means + (size_t)b * (H + 1) * ds;
    const double* Sg = A.out_covs + (size_t)b * (H + 1) * ds * ds;
    for (int i = tid; i <= H; i += blockDim.x)
        s_ct[i] = fc_state_cost<DS>(A.cost, mu + i * ds, Sg + i * ds * ds,
                                    A.grad ? s_dl + (size_t)i * nz : nullptr, A.grad ? s_dl + (size_t)i * nz + ds : nullptr);
    // the entry / term assignment of the sweep and the first prefetch (independent of the cost terms)
    const int n = tid >> 2, c4 = tid & 3, ne = D + ds * ds;
    const bool on = A.grad && n < ne;
    const int eu = n, kl = n >= D ? ((n - D) / ds) * D + (n - D) % ds : 0;
    const size_t B = A.B;
    double pf[GPMPC_FC_TAIL_TERMS];
    auto fetch = [&](int t) {
        const double* dm_du = A.dmean_du + ((size_t)(t - 1) * B + b) * ds * D;
        const double* dm_dS = A.dmean_dS + ((size_t)(t - 1) * B + b) * ds * D * D;
        const double* dc_du = A.dcov_du + ((size_t)(t - 1) * B + b) * ds * ds * D;
        const double* dc_dS = A.dcov_dS + ((size_t)(t - 1) * B + b) * ds * ds * D * D;
#pragma unroll
        for (int k = 0; k < GPMPC_FC_TAIL_TERMS; ++k) {
            const int j = c4 + 4 * k;
            double v = 0.0;
            if (on && j < nz) {
                if (n < D) v = j < ds ? dm_du[j * D + eu] : dc_du[(j - ds) * D + eu];
                else v = j < ds ? dm_dS[j * D * D + kl] : dc_dS[(size_t)(j - ds) * D * D + kl];
            }
            pf[k] = v;
        }
    };
    if (A.grad) fetch(H);
    __syncthreads();
    if (tid == 0) {
        double total = 0.0;
        for (int i = 0; i <= H; ++i) total += s_ct[i];
        total += fc_input_cost(H, da, A.cost, s_U, A.grad ? s_gU : nullptr);
        A.out_cost[b] = total;
    }
    if (!A.grad) return;
    double* adj = s_adj; double* nxt = s_adj + nz;
    for (int r = tid; r < nz; r += blockDim.x) adj[r] = s_dl[(size_t)H * nz + r];
    __syncthreads();
    for (int t = H; t >= 1; --t) {
        double cur[GPMPC_FC_TAIL_TERMS];
#pragma unroll
        for (int k = 0; k < GPMPC_FC_TAIL_TERMS; ++k) cur[k] = pf[k];
        if (t > 1) fetch(t - 1);
        // s_g[n]: n < D: d/du_n;  n >= D: d/dS_kl of the state block, (k, l) = ((n - D) / ds, (n - D) % ds)
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < GPMPC_FC_TAIL_TERMS; ++k) {
            const int j = c4 + 4 * k;
            if (j < nz) s = fma(adj[j], cur[k], s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (on && c4 == 0) s_g[n] = s;
        __syncthreads();
        for (int r = tid; r < nz + da; r += blockDim.x) {
            if (r < ds) nxt[r] = s_dl[(size_t)(t - 1) * nz + r] + s_g[r];
            else if (r < nz) nxt[r] = s_dl[(size_t)(t - 1) * nz + r] + s_g[D + (r - ds)];
            else s_gU[(t - 1) * da + (r - nz)] += s_g[ds + (r - nz)];
        }
        __syncthreads();
        double* tmp = adj; adj = nxt; nxt = tmp;
    }
    for (int r = tid; r < H * da; r += blockDim.x) A.out_grad[(size_t)b * H * da + r] = s_gU[r];
}

// ---------------------------------------------------------------------------
// Small batches: TWO launches per horizon step (round 4).  The step-per-four-launches form above (assemble, prep, pair kernel,
// finish) leaves the chip to ~10 workgroups for three of the four launches and runs the pair kernel on 256x256 tiles (N = 2048,
// B = 1: 528 workgroups, two waves per SIMD, each a chain of 256 dependent column round trips: 140 us of a 209 us step).  Here
//   k_fc_head(t)  workgroups per (trajectory, unit), by role: RS "column-row" workgroups (one row chunk of the unit's G rows each; the
//                 first also writes the unit's set-up records and, for unit 0, the trajectory), one "mean" workgroup (variance units:
//                 B = (S + Lambda)^-1, the O(N) mean sums, mean and mean Jacobians) and one "closing" workgroup (the full moment
//                 sums and the Jacobians of ITS unit for step t-1).  The first two kinds sum the Z0 partial sums of EVERY unit (each
//                 needs the whole covariance of step t-1 to assemble (u_t, S_t); bit-identical in all workgroups of a trajectory: same
//                 code, same order) and run their part of the unit's D x D set-up themselves (moment_dev.h);
//   pair kernel   pair_kernel_sbf.h on 256x64 (256x128, 256x256) tiles, two software-pipelined columns per iteration at small launches.
// The set-up records (sp) are double-buffered by step parity: the workgroup of a cross unit reads the records of its two
// variance units of the step being closed while their workgroups write this step's.
// ---------------------------------------------------------------------------
struct FcHeadArgs {
    MomArgs M;                    // M.sp: records of THIS step; Jacobian pointers: slice of the step being closed
    const double* sp_prev;        // records of the step being closed
    const double* x0; const double* U;
    int B, H, da;
    int rsplit, zbase;            // row chunks per unit (RS); role index of blockIdx.z = 0 (see k_fc_head)
    const double* part0;          // [B][nwork] Z0 partial sums of the pair kernel, contiguous
    double* out_means; double* out_covs;
};

template <int D>
__global__ __launch_bounds__(256) void k_fc_head(FcHeadArgs A, int t) {
    constexpr int NMX = 1 + D + D * (D + 1) / 2;
    constexpr bool STAGE = MomPrepLds<D>::STAGE;
    __shared__ MomPrepLds<D> sh;
    __shared__ double s_small[STAGE ? 1 : 256];
    __shared__ double s_z[NMX], s_z0[GPMPC_MAX_DS + GPMPC_MAX_PAIRS], s_mu[GPMPC_MAX_DS], s_cv[GPMPC_MAX_DS * GPMPC_MAX_DS];
    const MomArgs& M = A.M;
    const int q = blockIdx.x, unit = blockIdx.y, tid = threadIdx.x, ds = M.ds, nunits = M.nunits, nm = M.nm, H = A.H;
    // Workgroups of a (trajectory, unit), by what they do after assembling the step's input: [0, RS) the column rows of one row chunk
    // each (the first also the set-up records and the record of the trajectory), RS the mean side of a variance unit, RS + 1 the
    // Jacobians of the step being closed.  The final call (t = H + 1) launches the last kind only.  (The mean side split over row
    // chunks with an arrival counter was built and measured: the device-scope fence it needs writes back and invalidates L2 on this
    // multi-XCD part, 16 k cycles per workgroup; profiles/r04/fc_head_stamps_ticket.txt.)
    const int RS = A.rsplit, zr = blockIdx.z + A.zbase;
    const bool g_role = zr < RS, mean_role = zr == RS, close_role = zr > RS;
    const int rs = g_role ? zr : 0;
    if (mean_role && unit >= ds) return;                                  // cross units have no mean side
    if (close_role && t <= H && !(t >= 2 && M.grad)) return;              // nothing to close
    if (close_role && t > H && !M.grad && unit != 0) return;              // final call without Jacobians: unit 0 records step H
    const bool all_units = !close_role || t > H;                          // needs the whole covariance of step t-1
    // Between two head kernels the pair kernel streams hundreds of MB through the L2: whatever this kernel reads first comes from the
    // Infinity Cache / HBM (~2.5 k cycles per dependent round trip, 5 of them in a row as first written).  Everything small that the
    // later phases read -- hyper-parameters, the set-up records of the step being closed, the work-list index -- is touched NOW, in
    // the shadow of the partial-sum loads, and found in the CU's vector cache afterwards.
    double warm = 0.0;
    {
        const int nlam = ds * D;
        if (tid < nlam) warm += M.lam[tid];
        if (tid < ds) warm += M.sf[tid];
        if (tid <= nunits) warm += (double)M.ustart[tid];
        if (tid < 2 * M.npairs) warm += (double)M.pair_ab[tid];
        if (t >= 2) {
            const double* rec = A.sp_prev + (size_t)q * nunits * M.sps;
            for (int e = tid * 8; e < nunits * M.sps; e += 256 * 8) warm += rec[e];      // one load per 64-byte half line
        }
    }
    // a column-row workgroup's training points: requested now, needed after the sums and the set-up
    double xpre[D + 1];
    {
        const int rchunk = (((M.Np + 255) / 256 + RS - 1) / RS) * 256, i = rs * rchunk + tid;
        const bool in = !close_role && t <= H && i < M.Np && i < (rs + 1) * rchunk;
#pragma unroll
        for (int k = 0; k < D; ++k) xpre[k] = in ? M.XT[(size_t)k * M.Np + i] : 0.0;
        xpre[D] = 0.0;
    }
#if defined(GPMPC_FC_STAMPS)
    if (tid == 0) sh.stamp = (t == 3 && q == 0 && unit == 0 && zr <= RS + 1 && (zr == 0 || zr >= RS)) ? (zr == 0 ? 0 : (zr == RS ? 16 : 32)) : -1;
    __syncthreads();
#endif
    GPMPC_FST(0);
    if (t >= 2) {
        double* red = STAGE ? sh.g : s_small;             // 256 doubles; the staging buffer of the prep phase is free until then
        const double* __restrict__ part = M.part + (size_t)q * M.nwork * nm;
        const double* __restrict__ p0 = A.part0 + (size_t)q * M.nwork;     // (strided by nm doubles in `part`: a cache line per value)
        const int L = 256 / nunits < 64 ? 256 / nunits : 64;      // lanes per unit
        {   // Z0 of every unit (of this unit alone where that is all that is needed; the same lanes, the same order): L lanes per
            // unit, two independent sums per lane, then the L partial sums in order
            const int u = tid / L, l = tid - u * L;
            if (u < nunits && (all_units || u == unit)) {
                const int w0 = M.ustart[u], w1 = M.ustart[u + 1];
                double s0 = 0.0, s1 = 0.0;
                int wi = w0 + l;
#pragma unroll 8
                for (; wi + L < w1; wi += 2 * L) { s0 += p0[wi]; s1 += p0[wi + L]; }
                if (wi < w1) s0 += p0[wi];
                red[tid] = s0 + s1;
            }
        }
        __syncthreads();
        GPMPC_FST(1);
        if (tid < nunits && (all_units || tid == unit)) {
            double s = 0.0;
            for (int l = 0; l < L; ++l) s += red[tid * L + l];
            s_z0[tid] = s;
        }
        __syncthreads();
        GPMPC_FST(2);
        if (nm > 1 && close_role) {   // the other moments of this workgroup's unit: 256 / nm lanes per moment
            const int G = 256 / nm, g = tid / nm, m = tid - g * nm;
            if (g < G) {
                const int w0 = M.ustart[unit], w1 = M.ustart[unit + 1];
                double s0 = 0.0, s1 = 0.0;
                int wi = w0 + g;
#pragma unroll 4
                for (; wi + G < w1; wi += 2 * G) { s0 += part[(size_t)wi * nm + m]; s1 += part[(size_t)(wi + G) * nm + m]; }
                if (wi < w1) s0 += part[(size_t)wi * nm + m];
                red[tid] = s0 + s1;
            }
            __syncthreads();
            if (tid < nm) {
                double s = 0.0;
                for (int gg = 0; gg < G; ++gg) s += red[gg * nm + tid];
                s_z[tid] = tid == 0 ? s_z0[unit] : s;
            }
        } else if (tid == 0) s_z[0] = s_z0[unit];
        GPMPC_FST(3);
        // moments of step t-1 (k_mom_finish's formulas)
        if (all_units && tid < nunits) {
            const double* sp = A.sp_prev + ((size_t)q * nunits + tid) * M.sps;
            const double c = sp[0];
            if (tid < ds) {
                const double mu = sp[1], sf2 = sp[2];
                const double T = c * s_z0[tid];
                s_mu[tid] = mu;
                s_cv[tid * ds + tid] = sf2 - T - mu * mu;
            } else {
                const int pr = tid - ds, a = M.pair_ab[2 * pr], b = M.pair_ab[2 * pr + 1];
                const double mua = A.sp_prev[((size_t)q * nunits + a) * M.sps + 1], mub = A.sp_prev[((size_t)q * nunits + b) * M.sps + 1];
                const double F = c * s_z0[tid];
                const double cov = F - mua * mub;
                s_cv[a * ds + b] = cov; s_cv[b * ds + a] = cov;
            }
        }
        __syncthreads();
    }
    if (all_units) {
        // record the state of step t-1 (one workgroup per trajectory) and assemble the input distribution of step t (k_fc_assemble)
        const bool rec = unit == 0 && ((g_role && rs == 0) || close_role);
        if (tid < ds) {
            const double m = (t == 1) ? A.x0[(size_t)q * ds + tid] : s_mu[tid];
            sh.u[tid] = m;
            if (rec) A.out_means[((size_t)q * (H + 1) + (t - 1)) * ds + tid] = m;
        } else if (tid < D && t <= H) sh.u[tid] = A.U[((size_t)q * H + (t - 1)) * A.da + (tid - ds)];
        if (tid < D * D) {
            const int k = tid / D, l = tid - k * D;
            double c = 0.0;
            if (k < ds && l < ds) {
                c = (t == 1) ? (k == l ? GPMPC_INIT_VAR : 0.0) : s_cv[k * ds + l];
                if (rec) A.out_covs[(((size_t)q * (H + 1) + (t - 1)) * ds + k) * ds + l] = c;
            } else if (k >= ds && l >= ds) c = (k == l) ? GPMPC_ACTION_VAR : 0.0;
            sh.S[tid] = c;
        }
    }
    __syncthreads();
    GPMPC_FST(4);
    if (close_role) {
        if (M.grad) mom_finish_unit_wg<D>(M, A.sp_prev, q, unit, s_z, sh);      // Jacobians of step t-1, this unit
        GPMPC_FST(5);
        if (warm == 1.2345e301) A.out_means[0] = warm;
        return;
    }
    mom_prep_body<D>(M, q, unit, sh, rs, RS, g_role ? 1 : 2, g_role, xpre);
    if (warm == 1.2345e301) A.out_means[0] = warm;           // (never: keeps the warm-up loads)
}

#if defined(GPMPC_FC_STAMPS)
extern "C" int gpmpc_debug_fc_stamps(unsigned long long* host_out) {      // [64]
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fc_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -3;
}
#endif

struct FcPlan2 { int tiling, waves, nwork, nunits, nm, pps, sps, gw, rsplit, cu, fcs /* one lambda: the cross units by pair_kernel_sbfx.h */, fcs_q /* its tile width: 0 64 columns | 1 16 */, ntri /* work items of the variance units */; size_t off_part0, off_pp, off_sp0, off_sp1, off_part, off_G, off_dmu, off_dmS, off_dcu, off_dcS, total; };

// 1: the two-launch form applies and is taken (GPMPC_FC_FORM = 0 / 1 forces)
static int plan_fc2(const gpmpc_pack* p, int B, int H, bool grad, FcPlan2* r) {
    if (p->npairs == 0 || !p->fullcov || p->D - p->ds > 2 || p->tune.pair_sb == 0 || p->tune.fc_form == 0) return 0;
    // Measured over N = 300 ... 2048, B = 1 ... 128 (profiles/r04/fullcov_small_batch_ab.txt): the two-launch form wins while the
    // large-tile launch of the four-launch form is below ~16 k workgroups, and at any batch for N <= 512; its tiles: the narrowest
    // whose launch stays within ~16 k workgroups.
    // One lambda for all GPs, 2 <= ds <= 4 (pair_kernel_sbfx.h: the cross units in ONE pass over the pairs, no weight stream): taken where it
    // is ahead of the per-unit kernels -- tools/fullcov_ab.py --shared-lambda, profiles/r05/fullcov_shared_ab*.txt, ms per rollout call per-unit |
    // shared: N = 2048, ds = 4: B = 1 1.47 | 1.63, 2 2.23 | 1.84, 4 3.99 | 2.81, 16 13.1 | 8.09, 64 47.9 | 28.0, 256 192 | 119 (two launches per step at
    // ANY batch: the four-launch form has no shared variant); N = 1024: B = 4 1.30 | 1.63, 8 2.20 | 1.92, 64 12.8 | 8.87; N = 512, ds = 3: B = 32
    // 1.35 | 1.55, 64 2.37 | 2.16; N = 300, ds = 4: B = 32 0.89 | 0.92, 64 1.37 | 1.11; ds = 2 (ONE cross unit): never.  Below the threshold a launch
    // is a chain of latencies and the cross-unit kernel's wave waits ~1 k cycles per column (a flat ~35 us per horizon step).
    const bool fcs_can = p->shared_lambda && p->fcs_rows && p->tune.shared != 0 && p->tune.fc_shared != 0 &&
                         (p->tune.fc_shared == 1 || (double)B * p->Np * p->Np * p->npairs >= 3.5e7);
    const long w256 = fcs_can ? 0 : (long)B * p->wl[1][0].nwork;
    if (p->tune.fc_form != 1 && w256 >= 16384 && p->Np > 512) return 0;
    r->tiling = (long)B * (fcs_can ? p->wl[1][2].ustart_host[p->ds] : p->wl[1][2].nwork) <= 16384 ? 2 : ((long)B * (fcs_can ? p->wl[1][4].ustart_host[p->ds] : p->wl[1][4].nwork) <= 16384 ? 4 : 0);
    if (p->tune.fc_tiling == 0 || p->tune.fc_tiling == 2 || p->tune.fc_tiling == 4) r->tiling = p->tune.fc_tiling;
    const gpmpc_worklist& w = p->wl[1][r->tiling];
    const size_t ds = p->ds, D = p->D, HB = (size_t)(grad ? H : 0) * B;
    r->waves = w.waves; r->nwork = w.nwork; r->nunits = w.nunits;
    // One lambda for all GPs (2 <= ds <= 4): the cross units share transform, exponent and -- up to beta_a,i beta_b,j -- their weights; one pass
    // of pair_kernel_sbfx.h over the pairs serves all of them without a weight stream, pair_kernel_sbf.h keeps the variance units (their
    // weights hold K_a^-1).  A trajectory's partial-sum slots: the variance units' work items, then pairs x 64x64 tiles.  GPMPC_FC_SHARED=0: off.
    {
        r->fcs = (fcs_can && p->fcs_ustart_dev[r->tiling][0]) ? 1 : 0;
        r->ntri = w.ustart_host[p->ds];
        // 16-column tiles for the smallest launches (a wave's column takes ~1.2 k cycles on its own: scalar loads of two rows, the dependent exponent
        // and exp, ds x 17 accumulations -- four times the waves, each a quarter as long: N = 2048, B = 1 1.63 -> 1.56 ms, N = 1024, B = 1 1.53 -> 1.12,
        // N = 300, ds = 4, B = 1 0.75 -> 0.55; from ~800 tiles of 64 columns the wide ones are ahead: N = 2048, B = 2 1.84 | 2.02 -- fullcov_shared_ab2.txt)
        r->fcs_q = (r->fcs && (long)B * p->fcs_ntile[0] < 800) ? 1 : 0;
        // ... and up to 256 columns per wave once the launch fills the chip many times over (N = 2048, ds = 4: B = 16 8.5 | 9.5 ms, B = 64 29.0 | 28.0,
        // B = 256 116.0 | 109.4; GPMPC_FC_XTILE=0|1|2 forces a width)
        if (r->fcs && (long)B * p->fcs_ntile[0] >= 24000 && p->fcs_tiles256_dev) r->fcs_q = 2;
        {
            static const int xt = getenv("GPMPC_FC_XTILE") ? atoi(getenv("GPMPC_FC_XTILE")) : -1;
            if (r->fcs && xt >= 0 && xt <= 2 && (xt < 2 || p->fcs_tiles256_dev)) r->fcs_q = xt;
        }
        if (r->fcs) r->nwork = p->fcs_total[r->tiling][r->fcs_q];
    }
    r->nm = gpmpc_num_moments(p->D, false, grad);
    // workgroups per (trajectory, unit) of the head kernel: one per 256 rows of column rows while the launch stays within ~4 per CU
    {
        const int blocks = (p->Np + 255) / 256, room = (int)(1024 / ((long)B * r->nunits));
        r->rsplit = blocks < room ? blocks : (room > 1 ? room : 1);
        if (r->rsplit > 16) r->rsplit = 16;
        if (p->tune.fc_rsplit > 0) r->rsplit = p->tune.fc_rsplit > blocks ? blocks : p->tune.fc_rsplit;
    }
    // columns per iteration of the pair kernel (pair_kernel_sbf.h): 2, software-pipelined, while a wave's column chain bounds the launch
    // (N = 2048, B = 1: 68 -> 57 us per launch, 4.7 TB/s of weights; N = 300 ... 1024 at B = 1: -12 %); 1 (8 instead of 5 waves per
    // SIMD) once the chip is full
    {
        const long wgs = (long)B * (r->fcs ? r->ntri : r->nwork);
        r->cu = (wgs < 6000 && !(p->Np < 512 && wgs >= 1024)) ? 2 : 1;
        if (p->tune.fc_cu == 1 || p->tune.fc_cu == 2 || p->tune.fc_cu == 4) r->cu = p->tune.fc_cu;
    }
    r->pps = 2 * (p->D + p->D * p->D); r->sps = msps_of(p->D); r->gw = gpmpc_sbf_gw(p->D, p->ds);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n * sizeof(double) + 255) & ~(size_t)255; return o; };
    r->off_pp = take((size_t)B * r->nunits * r->pps);
    r->off_sp0 = take((size_t)B * r->nunits * r->sps); r->off_sp1 = take((size_t)B * r->nunits * r->sps);
    r->off_part = take((size_t)B * r->nwork * r->nm); r->off_part0 = take((size_t)B * r->nwork);
    r->off_G = take((size_t)B * r->nunits * p->Np * r->gw);
    r->off_dmu = take(HB * ds * D); r->off_dmS = take(HB * ds * D * D);
    r->off_dcu = take(HB * ds * ds * D); r->off_dcS = take(HB * ds * ds * D * D);
    r->total = off;
    return 1;
}

template <int D>
static int run_fc2(const gpmpc_pack* p, const FcPlan2& r, FcArgs& T, bool grad, char* ws, hipStream_t s) {
    const int B = T.B, H = T.H;
    const size_t ds = p->ds, Dz = p->D;
    FcHeadArgs A;
    memset(&A, 0, sizeof(A));
    MomArgs& M = A.M;
    M.XT = p->XT; M.beta = p->beta; M.lam = p->lam; M.sf = p->sf;
    M.N = p->N; M.Np = p->Np; M.ds = p->ds; M.D = p->D; M.nq = B;
    M.pp = (double*)(ws + r.off_pp); M.part = (double*)(ws + r.off_part);
    M.pps = r.pps; M.sps = r.sps; M.nwork = r.nwork; M.nunits = r.nunits; M.nm = r.nm; M.grad = grad ? 1 : 0;
    M.ustart = r.fcs ? p->fcs_ustart_dev[r.tiling][r.fcs_q] : p->wl[1][r.tiling].ustart_dev;
    M.pair_ab = p->pair_ab_dev; M.npairs = p->npairs;
    M.G = (double*)(ws + r.off_G); M.gw = r.gw; M.ns2 = p->ds;
    A.x0 = T.x0; A.U = T.U; A.B = B; A.H = H; A.da = p->da;
    A.out_means = T.out_means; A.out_covs = T.out_covs;
    double* sp[2] = {(double*)(ws + r.off_sp0), (double*)(ws + r.off_sp1)};
    T.dmean_du = (double*)(ws + r.off_dmu); T.dmean_dS = (double*)(ws + r.off_dmS);
    T.dcov_du = (double*)(ws + r.off_dcu); T.dcov_dS = (double*)(ws + r.off_dcS);
    PairSbfArgs Q;
    Q.M = p->M; Q.XT = p->XT; Q.pp = M.pp; Q.G = M.G; Q.part = M.part; Q.work = p->wl[1][r.tiling].work_dev;
    Q.Np = p->Np; Q.B = B; Q.nunits = r.nunits; Q.nwork = r.nwork; Q.pps = r.pps; Q.nm = r.nm; Q.ntri = p->ds;
    Q.cu = r.cu; Q.part0 = (double*)(ws + r.off_part0); A.part0 = Q.part0;
    Q.pstride = 0;
    PairSbfxArgs X;
    memset(&X, 0, sizeof(X));
    if (r.fcs) {
        Q.nwork = r.ntri; Q.pstride = r.nwork;                       // the variance units' items only, within the trajectory's full set of slots
        X.XT = p->XT; X.lam = p->lam; X.beta = p->beta; X.sf = p->sf; X.pp = M.pp; X.G = M.G; X.rows = p->fcs_rows;
        X.part = M.part; X.part0 = Q.part0; X.pair_ab = p->pair_ab_dev;
        X.Np = p->Np; X.N = p->N; X.B = B; X.nunits = r.nunits; X.unit0 = p->ds; X.pps = r.pps; X.nm = r.nm;
        X.tj = p->fcs_tj; X.ntile = p->fcs_ntile[r.fcs_q]; X.jt = r.fcs_q == 1 ? 16 : (r.fcs_q == 2 ? 256 : 64); X.tiles = r.fcs_q == 2 ? p->fcs_tiles256_dev : nullptr;
        X.base = p->fcs_base[r.tiling]; X.pstride = r.nwork;
    }
    for (int t = 1; t <= H + 1; ++t) {
        M.sp = sp[t & 1]; A.sp_prev = sp[(t - 1) & 1];
        if (grad && t >= 2) {
            const size_t sl = (size_t)(t - 2) * B;
            M.dmean_du = T.dmean_du + sl * ds * Dz; M.dmean_dS = T.dmean_dS + sl * ds * Dz * Dz;
            M.dcov_du = T.dcov_du + sl * ds * ds * Dz; M.dcov_dS = T.dcov_dS + sl * ds * ds * Dz * Dz;
        }
        A.rsplit = r.rsplit; A.zbase = t > H ? r.rsplit + 1 : 0;
        hipLaunchKernelGGL(k_fc_head<D>, dim3(B, r.nunits, t > H ? 1 : r.rsplit + 2), dim3(256), 0, s, A, t);
        if (t > H) break;
        if (int rc = gpmpc_timed_pair_sbf(p->D, grad, p->ds, r.waves, Q, s)) return rc;
        // (the cross-unit kernel BESIDE this launch on a side stream was built and measured: N = 2048, B = 1 1.775 -> 1.751 ms, B = 2 2.02 -> 2.17 ms:
        // no gain, one stream)
        if (r.fcs) if (int rc = gpmpc_launch_pair_sbfx(p->D, grad, p->ds, X, s)) return rc;
    }
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}

static int launch_fc_tail(const FcArgs& A, int B, size_t lds, hipStream_t s) {
    switch (A.ds) {
        case 1: hipLaunchKernelGGL(k_fc_tail<1>, dim3(B), dim3(256), lds, s, A); break;
        case 2: hipLaunchKernelGGL(k_fc_tail<2>, dim3(B), dim3(256), lds, s, A); break;
        case 3: hipLaunchKernelGGL(k_fc_tail<3>, dim3(B), dim3(256), lds, s, A); break;
        case 4: hipLaunchKernelGGL(k_fc_tail<4>, dim3(B), dim3(256), lds, s, A); break;
        case 5: hipLaunchKernelGGL(k_fc_tail<5>, dim3(B), dim3(256), lds, s, A); break;
        case 6: hipLaunchKernelGGL(k_fc_tail<6>, dim3(B), dim3(256), lds, s, A); break;
        default: return GPMPC_E_ARG;
    }
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}

struct FcPlan { size_t off_u, off_S, off_mean, off_cov, off_var, off_dvu, off_dvS, off_dmu, off_dmS, off_dcu, off_dcS, off_mm, mm_bytes, total; };

static void plan_fc(const gpmpc_pack* p, int B, int H, bool grad, FcPlan* r) {
    const size_t ds = p->ds, D = p->D, HB = (size_t)(grad ? H : 1) * B;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n * sizeof(double) + 255) & ~(size_t)255; return o; };
    r->off_u = take((size_t)B * D); r->off_S = take((size_t)B * D * D);
    r->off_mean = take((size_t)B * ds); r->off_cov = take((size_t)B * ds * ds); r->off_var = take((size_t)B * ds);
    r->off_dvu = take((size_t)B * ds * D); r->off_dvS = take((size_t)B * ds * D * D);
    r->off_dmu = take(HB * ds * D); r->off_dmS = take(HB * ds * D * D);
    r->off_dcu = take(HB * ds * ds * D); r->off_dcS = take(HB * ds * ds * D * D);
    r->mm_bytes = gpmpc_moment_match_workspace_bytes(p, B);
    r->off_mm = off; off += (r->mm_bytes + 255) & ~(size_t)255;
    r->total = off;
}

extern "C" size_t gpmpc_rollout_fullcov_workspace_bytes(const gpmpc_pack* p, int B, int H, unsigned flags) {
    if (!p || B < 1 || H < 1) return 0;
    FcPlan r;
    plan_fc(p, B, H, (flags & GPMPC_WANT_GRAD) != 0, &r);
    FcPlan2 r2;
    if (plan_fc2(p, B, H, (flags & GPMPC_WANT_GRAD) != 0, &r2) && r2.total > r.total) return r2.total;
    return r.total;
}

extern "C" int gpmpc_rollout_fullcov_describe(const gpmpc_pack* p, int B, int H, unsigned flags, char* out, size_t out_bytes) {
    if (!p || !out || out_bytes < 64 || B < 1 || H < 1) return GPMPC_E_ARG;
    const bool grad = (flags & GPMPC_WANT_GRAD) != 0;
    FcPlan2 r2;
    if (plan_fc2(p, B, H, grad, &r2)) {
        const gpmpc_worklist& w = p->wl[1][r2.tiling];
        snprintf(out, out_bytes, "form=two_launch tiling=%dx%d workgroups=%ld columns_per_iteration=%d head_workgroups_per_unit=%d "
                 "kernel=gpmpc_pair_kernel_sbf<%d,%d,%s,%d>%s shared_cross_units=%d cross_tile_columns=%d", w.it, w.jt, (long)B * (r2.fcs ? r2.ntri : r2.nwork), r2.cu, r2.rsplit, p->D, p->ds,
                 grad ? "true" : "false", r2.cu, r2.fcs ? "+gpmpc_pair_kernel_sbfx" : "", r2.fcs, r2.fcs ? (r2.fcs_q == 1 ? 16 : (r2.fcs_q == 2 ? 256 : 64)) : 0);
    } else {
        snprintf(out, out_bytes, "form=four_launch tiling=by_gpmpc_moment_match workgroups=0 columns_per_iteration=1 "
                 "head_workgroups_per_unit=1 kernel=gpmpc_pair_kernel_sbf<%d,%d,%s,1>|staged", p->D, p->ds, grad ? "true" : "false");
    }
    return GPMPC_OK;
}

extern "C" int gpmpc_rollout_fullcov(const gpmpc_pack* p, int B, int H, const double* x0, const double* U,
                                     const gpmpc_cost_params* cost, unsigned flags, double* out_means, double* out_covs,
                                     double* out_cost, double* out_grad, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!p) return GPMPC_E_ARG;
    if (int rc_dev = gpmpc_check_device(p)) return rc_dev;
    if (!p || !x0 || !U || !cost || !out_means || !out_covs || !out_cost || !workspace || B < 1 || H < 1) return GPMPC_E_ARG;
    if (!p->built || (p->npairs > 0 && !p->fullcov)) return GPMPC_E_STATE;
    const bool grad = (flags & GPMPC_WANT_GRAD) != 0;
    if (grad && !out_grad) return GPMPC_E_ARG;
    FcPlan r;
    plan_fc(p, B, H, grad, &r);
    FcPlan2 r2;
    const bool two = plan_fc2(p, B, H, grad, &r2) != 0;
    if (workspace_bytes < (two ? r2.total : r.total)) return GPMPC_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const size_t ds = p->ds, D = p->D;
    FcArgs A;
    memset(&A, 0, sizeof(A));
    A.B = B; A.H = H; A.ds = p->ds; A.da = p->da; A.D = p->D; A.grad = grad ? 1 : 0;
    A.x0 = x0; A.U = U;
    const size_t nz = ds + ds * ds;
    const size_t lds = sizeof(double) * ((size_t)GPMPC_FC_WORKERS * ds * 2 * ds + (H + 1) + (size_t)(H + 1) * nz + 2 * nz + D + D * D + 2 * (size_t)H * p->da);
    if (lds > 60 * 1024) return GPMPC_E_ARG;
    if (two) {
        A.out_means = out_means; A.out_covs = out_covs; A.out_cost = out_cost; A.out_grad = out_grad; A.cost = *cost;
        int rc = GPMPC_E_ARG;
        switch (p->D) {
            case 2: rc = run_fc2<2>(p, r2, A, grad, ws, s); break;
            case 3: rc = run_fc2<3>(p, r2, A, grad, ws, s); break;
            case 4: rc = run_fc2<4>(p, r2, A, grad, ws, s); break;
            case 5: rc = run_fc2<5>(p, r2, A, grad, ws, s); break;
            case 6: rc = run_fc2<6>(p, r2, A, grad, ws, s); break;
            case 7: rc = run_fc2<7>(p, r2, A, grad, ws, s); break;
            case 8: rc = run_fc2<8>(p, r2, A, grad, ws, s); break;
        }
        if (rc != GPMPC_OK) return rc;
        return launch_fc_tail(A, B, lds, s);
    }
    A.u = (double*)(ws + r.off_u); A.S = (double*)(ws + r.off_S);
    A.mean = (double*)(ws + r.off_mean); A.cov = (double*)(ws + r.off_cov);
    A.out_means = out_means; A.out_covs = out_covs;
    A.dmean_du = (double*)(ws + r.off_dmu); A.dmean_dS = (double*)(ws + r.off_dmS);
    A.dcov_du = (double*)(ws + r.off_dcu); A.dcov_dS = (double*)(ws + r.off_dcS);
    A.out_cost = out_cost; A.out_grad = out_grad; A.cost = *cost;
    double* var = (double*)(ws + r.off_var);
    double* dvu = (double*)(ws + r.off_dvu); double* dvS = (double*)(ws + r.off_dvS);
    const dim3 gb((B + 63) / 64), tb(64);
    for (int t = 1; t <= H; ++t) {
        hipLaunchKernelGGL(k_fc_assemble, gb, tb, 0, s, A, t);
        const size_t sl = grad ? (size_t)(t - 1) * B : 0;
        int rc = gpmpc_moment_match_ex(p, B, A.u, A.S, grad ? GPMPC_WANT_GRAD : 0u, A.mean, var, A.cov, nullptr,
                                    grad ? A.dmean_du + sl * ds * D : nullptr, grad ? A.dmean_dS + sl * ds * D * D : nullptr,
                                    grad ? dvu : nullptr, grad ? dvS : nullptr,
                                    grad ? A.dcov_du + sl * ds * ds * D : nullptr, grad ? A.dcov_dS + sl * ds * ds * D * D : nullptr,
                                    ws + r.off_mm, r.mm_bytes, stream, p->ds);
        if (rc != GPMPC_OK) return rc;
    }
    hipLaunchKernelGGL(k_fc_assemble, gb, tb, 0, s, A, H + 1);      // records step H
    return launch_fc_tail(A, B, lds, s);
}
