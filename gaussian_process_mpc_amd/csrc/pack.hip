// GP state ("pack") construction kernels: everything that depends only on the training data and
// the hyper-parameters, hoisted out of the per-call path of the reference
// (src/tools/uncertainty_prop.py:324-327 beta, :392-394 Lambda_part, :399 Ky_inv - beta beta^T;
// src/gpr.py:163-170 Kf / Ky).
#include "gpmpc_internal.h"
#include <cstdlib>

// X [N][D] -> Xp [Np][D] (zero rows appended) and XT [D][Np]
__global__ void k_pack_points(const double* __restrict__ X, int N, int Np, int D,
                              double* __restrict__ Xp, double* __restrict__ XT) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Np) return;
    for (int k = 0; k < D; ++k) {
        const double v = i < N ? X[(size_t)i * D + k] : 0.0;
        Xp[(size_t)i * D + k] = v;
        XT[(size_t)k * Np + i] = v;
    }
}

// beta_a = Ky_inv_a @ y_a  (src/tools/uncertainty_prop.py:327); one wave per row.
__global__ __launch_bounds__(256) void k_pack_beta(const double* __restrict__ Kinv, size_t ld, size_t gstride, const double* __restrict__ Y,
                                                    int N, int Np, int ds, double* __restrict__ beta) {
    const int a = blockIdx.y;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= Np) return;
    double s = 0.0;
    if (row < N) {
        const double* __restrict__ r = Kinv + (size_t)a * gstride + (size_t)row * ld;
        for (int j = lane; j < N; j += 64) s = fma(r[j], Y[(size_t)j * ds + a], s);
        s = wave_sum(s);
    }
    if (lane == 0) beta[(size_t)a * Np + row] = s;
}

// M_a: element (i,j), i <= j, stored at [j*Np + i]:
//   w_ij * ( (Kinv[i][j] + Kinv[j][i])/2 - beta_i beta_j ) * sigma_f^4 * exp(-1/4 sum_k (x_ik - x_jk)^2 / lambda_k)
// with w = 1 on the diagonal and 2 above it; everything else (lower triangle, padding) is zero.
// 32x32 tiles staged through LDS so that both Kinv[i][j] and Kinv[j][i] are read coalesced.
__global__ __launch_bounds__(256) void k_pack_weights(const double* __restrict__ Kinv, size_t ld, size_t gstride, const double* __restrict__ beta,
                                                       const double* __restrict__ XT, const double* __restrict__ lam,
                                                       const double* __restrict__ sf, int N, int Np, int D,
                                                       double* __restrict__ M) {
    __shared__ double s_t[32][33];
    const int a = blockIdx.z;
    const int ti = blockIdx.x * 32, tj = blockIdx.y * 32;   // tile origin: rows i in [ti,ti+32), cols j in [tj,tj+32)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const double* __restrict__ Ka = Kinv + (size_t)a * gstride;
    double* __restrict__ Ma = M + (size_t)a * Np * Np;
    if (tj + 31 < ti) {   // strictly below the diagonal
        for (int r = ty; r < 32; r += 8) {
            const int j = tj + r, i = ti + tx;
            if (j < Np && i < Np) Ma[(size_t)j * Np + i] = 0.0;
        }
        return;
    }
    // s_t[r][c] = Kinv[ti + r][tj + c]  (row-major read, coalesced over c)
    for (int r = ty; r < 32; r += 8) {
        const int i = ti + r, j = tj + tx;
        s_t[r][tx] = (i < N && j < N) ? Ka[(size_t)i * ld + j] : 0.0;
    }
    __syncthreads();
    const double sf2 = sf[a] * sf[a], sf4 = sf2 * sf2;
    for (int r = ty; r < 32; r += 8) {
        const int j = tj + r, i = ti + tx;                   // write M[j][i], coalesced over i
        if (j >= Np || i >= Np) continue;
        double out = 0.0;
        if (i <= j && j < N) {
            const double kji = Ka[(size_t)j * ld + i];       // coalesced over i
            const double kij = s_t[tx][r];
            double d2 = 0.0;
            for (int k = 0; k < D; ++k) {
                const double d = XT[(size_t)k * Np + i] - XT[(size_t)k * Np + j];
                d2 = fma(d * d, 1.0 / lam[a * D + k], d2);
            }
            const double wsym = 0.5 * (kij + kji) - beta[(size_t)a * Np + i] * beta[(size_t)a * Np + j];
            out = (i == j ? 1.0 : 2.0) * wsym * sf4 * exp(-0.25 * d2);
        }
        Ma[(size_t)j * Np + i] = out;
    }
}

struct gpmpc_small_payload { unsigned w[128]; };
__global__ void k_upload_small(gpmpc_small_payload p, unsigned* __restrict__ dst, int nwords) {
    const int i = threadIdx.x;
    if (i < nwords) dst[i] = p.w[i];
}
int gpmpc_upload_small(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s) {
    if (!dst_dev || !src_host || bytes == 0 || bytes > sizeof(gpmpc_small_payload) || (bytes & 3)) return GPMPC_E_ARG;
    gpmpc_small_payload p;
    memset(&p, 0, sizeof(p));
    memcpy(&p, src_host, bytes);
    hipLaunchKernelGGL(k_upload_small, dim3(1), dim3(128), 0, s, p, (unsigned*)dst_dev, (int)(bytes / 4));
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}

extern "C" int gpmpc_store_host(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
    return gpmpc_upload_small(dst_dev, src_host, bytes, (hipStream_t)stream);
}

// Kf = sigma_f^2 exp(-1/2 d2), Ky = Kf + noise_var I   (src/gpr.py:163-170)
__global__ void k_build_ky(const double* __restrict__ X, int n, int D, const double* __restrict__ lam,
                           double sf2, double noise_var, double* __restrict__ Kf, double* __restrict__ Ky) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j >= n) return;
    double d2 = 0.0;
    for (int k = 0; k < D; ++k) {
        const double d = X[(size_t)i * D + k] - X[(size_t)j * D + k];
        d2 = fma(d * d, 1.0 / lam[k], d2);
    }
    const double kf = sf2 * exp(-0.5 * d2);
    if (Kf) Kf[(size_t)i * n + j] = kf;
    Ky[(size_t)i * n + j] = kf + (i == j ? noise_var : 0.0);
}

// One lambda for all GPs, full-covariance rollout (pair_kernel_sbfx.h): the constant part of a column j of every cross unit,
//   rows[j] = [x_jk (D) | (N/ln2)/4 sum_k x_jk^2 / lambda_k | beta_c,j (c < ds) | pad],   beta = 0 for padded columns
__global__ void k_pack_fcs_rows(const double* __restrict__ beta, const double* __restrict__ XT, const double* __restrict__ lam,
                                int N, int Np, int D, int ds, int rw, double* __restrict__ rows) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Np) return;
    double* r = rows + (size_t)j * rw;
    double e = 0.0;
    for (int k = 0; k < D; ++k) { const double x = XT[(size_t)k * Np + j]; r[k] = x; e = fma(x * x, 1.0 / lam[k], e); }
    r[D] = (0.25 * 0x1.71547652b82fep+11) * e;               // (GPMPC_EXP_NEG_INV_C: the exponent travels scaled, fast_exp.h)
    for (int c = 0; c < ds; ++c) r[D + 1 + c] = j < N ? beta[(size_t)c * Np + j] : 0.0;
    for (int k = D + 1 + ds; k < rw; ++k) r[k] = 0.0;
}

// (allocations on first use; the rows are refreshed by every build while the pack serves the full-covariance rollout with one lambda)
static int fcs_refresh(gpmpc_pack* p, hipStream_t s) {
    if (!p->fullcov || p->npairs == 0 || !p->shared_lambda || p->ds < 2 || p->ds > 4 || p->D - p->ds < 1 || p->D - p->ds > 2) return GPMPC_OK;
    if (!p->fcs_rows) {
        p->fcs_rw = (p->D + 1 + p->ds + 1) & ~1;
        p->fcs_tj = p->Np / 64;
        p->fcs_ntile[0] = p->fcs_tj * (p->fcs_tj + 1) / 2; p->fcs_ntile[1] = 4 * p->fcs_ntile[0];   // 64 x 64 | 64 x 16 tiles of the upper triangle (pair_kernel_sbfx.h)
        {   // ... and row block ti's columns from its diagonal block on in chunks of 256 (launches that fill the chip: a third of a 64-column
            // wave's time is prologue and row sums)
            const int T = p->fcs_tj;
            int n2 = 0;
            for (int ti = 0; ti < T; ++ti) n2 += (p->Np - 64 * ti + 255) / 256;
            int* h = (int*)malloc(sizeof(int) * 3 * (size_t)n2);
            if (!h) return GPMPC_E_ALLOC;
            int k = 0;
            for (int ti = 0; ti < T; ++ti)
                for (int j0 = 64 * ti; j0 < p->Np; j0 += 256) { h[3 * k] = ti; h[3 * k + 1] = j0; h[3 * k + 2] = p->Np - j0 < 256 ? p->Np - j0 : 256; ++k; }
            p->fcs_ntile[2] = n2;
            hipError_t e = hipMalloc(&p->fcs_tiles256_dev, sizeof(int) * 3 * (size_t)n2);
            if (e == hipSuccess) e = hipMemcpy(p->fcs_tiles256_dev, h, sizeof(int) * 3 * (size_t)n2, hipMemcpyHostToDevice);
            free(h);
            if (e != hipSuccess) return GPMPC_E_ALLOC;
        }
        if (hipMalloc(&p->fcs_rows, sizeof(double) * (size_t)p->Np * p->fcs_rw) != hipSuccess) { p->fcs_rows = nullptr; return GPMPC_E_ALLOC; }
        for (int k : {0, 2, 4}) {
            const gpmpc_worklist& w = p->wl[1][k];
            if (!w.work_dev) continue;
            int ust[GPMPC_MAX_DS + GPMPC_MAX_PAIRS + 1];
            const int ntri = w.ustart_host[p->ds];
            for (int u = 0; u <= p->ds; ++u) ust[u] = w.ustart_host[u];
            p->fcs_base[k] = ntri;
            const size_t nb = sizeof(int) * (p->ds + p->npairs + 1);
            for (int q = 0; q < 3; ++q) {
                for (int pr = 1; pr <= p->npairs; ++pr) ust[p->ds + pr] = ntri + pr * p->fcs_ntile[q];
                p->fcs_total[k][q] = ntri + p->npairs * p->fcs_ntile[q];
                if (hipMalloc(&p->fcs_ustart_dev[k][q], nb) != hipSuccess) { p->fcs_ustart_dev[k][q] = nullptr; return GPMPC_E_ALLOC; }
                GPMPC_HIP(hipMemcpy(p->fcs_ustart_dev[k][q], ust, nb, hipMemcpyHostToDevice));
            }
        }
    }
    hipLaunchKernelGGL(k_pack_fcs_rows, dim3((p->Np + 255) / 256), dim3(256), 0, s, p->beta, p->XT, p->lam, p->N, p->Np, p->D, p->ds,
                       p->fcs_rw, p->fcs_rows);
    GPMPC_HIP(hipGetLastError());
    return GPMPC_OK;
}

// Cross-covariance weights of the GP pair (a, b), a < b:  element (i,j) at [j*Np + i]
//   beta_a[i] beta_b[j] sfa^2 sfb^2 exp(-1/2 sum_k (x_ik - x_jk)^2 / (lambda_ak + lambda_bk))
// (the x-independent factor of k_a(x,x_i) k_b(x,x_j), src/tools/uncertainty_prop.py:448-460 as a Gaussian product).
__global__ void k_pack_cross(const double* __restrict__ beta, const double* __restrict__ XT, const double* __restrict__ lam,
                             const double* __restrict__ sf, const int* __restrict__ pair_ab, int N, int Np, int D, int ds,
                             double* __restrict__ Mx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, pr = blockIdx.z;
    if (i >= Np) return;
    const int a = pair_ab[2 * pr], b = pair_ab[2 * pr + 1];
    double out = 0.0;
    if (i < N && j < N) {
        double d2 = 0.0;
        for (int k = 0; k < D; ++k) {
            const double d = XT[(size_t)k * Np + i] - XT[(size_t)k * Np + j];
            d2 = fma(d * d, 1.0 / (lam[a * D + k] + lam[b * D + k]), d2);
        }
        const double s2 = sf[a] * sf[a] * sf[b] * sf[b];
        out = beta[(size_t)a * Np + i] * beta[(size_t)b * Np + j] * s2 * exp(-0.5 * d2);
    }
    Mx[((size_t)pr * Np + j) * Np + i] = out;
}

// tile_slot: the 4th entry of an item is the tile's index within its unit instead of j1 (the shared-lambda kernel computes
// j1 from j0 and writes its partial sums by (GP, tile)); *tiles_out = tiles per unit.
static int build_worklist(int Np, int it, int jt, int ds, int npairs, bool xcd_sort, gpmpc_worklist* w,
                          bool tile_slot = false, int* tiles_out = nullptr) {
    const int ti = (Np + it - 1) / it, tj = (Np + jt - 1) / jt;
    const size_t cap = (size_t)ti * tj * (ds + npairs);
    int* h = (int*)malloc(sizeof(int) * 4 * cap);
    if (!h) return -1;
    int n = 0;
    for (int u = 0; u < ds + npairs; ++u) {
        w->ustart_host[u] = n;
        for (int i0 = 0; i0 < Np; i0 += it)
            for (int j0 = 0; j0 < Np; j0 += jt) {
                const int j1 = j0 + jt < Np ? j0 + jt : Np;
                if (u < ds && j1 <= i0) continue;            // variance unit: tile wholly below the diagonal
                h[4 * n] = u; h[4 * n + 1] = i0; h[4 * n + 2] = j0; h[4 * n + 3] = tile_slot ? n - w->ustart_host[u] : j1; ++n;
            }
        if (tiles_out) *tiles_out = n - w->ustart_host[u];
    }
    w->ustart_host[ds + npairs] = n;
    w->contiguous = 1;
    if (xcd_sort && n >= 16) {
        // Re-order so that the list position p (the pair kernel maps p % 8 to an XCD) groups tiles that share a COLUMN
        // range: they read the same G rows and, per unit, the same column block of M.  Odd units use the mirrored
        // column index so that the triangular tile counts (1..T per column block) balance across the 8 XCDs.
        int* q[8]; int cnt[8] = {0}, pos[8] = {0};
        for (int x = 0; x < 8; ++x) q[x] = (int*)malloc(sizeof(int) * 4 * (size_t)n);
        for (int k = 0; k < n; ++k) {
            const int u = h[4 * k], tjx = h[4 * k + 2] / jt;
            const int x = ((u & 1) ? (tj - 1 - tjx) : tjx) & 7;
            memcpy(q[x] + 4 * cnt[x]++, h + 4 * k, sizeof(int) * 4);
        }
        // within an XCD: column block major, then row tile (consecutive items share their G rows, see pair_kernel_sb.h)
        for (int x = 0; x < 8; ++x)
            qsort(q[x], cnt[x], sizeof(int) * 4, [](const void* a, const void* b) {
                const int* u = (const int*)a; const int* v = (const int*)b;
                if (u[0] != v[0]) return u[0] - v[0];
                if (u[2] != v[2]) return u[2] - v[2];
                return u[1] - v[1];
            });
        for (int p = 0; p < n; ++p) {
            int x = p & 7;
            if (pos[x] >= cnt[x]) {                             // this XCD's queue ran dry: take from the fullest one
                int best = 0;
                for (int y = 1; y < 8; ++y) if (cnt[y] - pos[y] > cnt[best] - pos[best]) best = y;
                x = best;
            }
            memcpy(h + 4 * p, q[x] + 4 * pos[x]++, sizeof(int) * 4);
        }
        for (int x = 0; x < 8; ++x) free(q[x]);
        w->contiguous = 0;
    }
    w->it = it; w->waves = it / 64 > 0 ? it / 64 : 1; w->jt = jt; w->nunits = ds + npairs; w->nwork = n;
    hipError_t e = hipMalloc(&w->work_dev, sizeof(int) * 4 * (size_t)n);
    if (e == hipSuccess) e = hipMemcpy(w->work_dev, h, sizeof(int) * 4 * (size_t)n, hipMemcpyHostToDevice);
    w->perm_dev = nullptr;
    if (e == hipSuccess && !w->contiguous) {                      // per-unit index of the re-ordered list (finish_step, step.hip)
        int* perm = (int*)malloc(sizeof(int) * (size_t)n);
        int fill[GPMPC_MAX_DS + GPMPC_MAX_PAIRS + 1];
        for (int u = 0; u <= ds + npairs; ++u) fill[u] = w->ustart_host[u];
        for (int k = 0; k < n; ++k) perm[fill[h[4 * k]]++] = k;
        e = hipMalloc(&w->perm_dev, sizeof(int) * (size_t)n);
        if (e == hipSuccess) e = hipMemcpy(w->perm_dev, perm, sizeof(int) * (size_t)n, hipMemcpyHostToDevice);
        free(perm);
    }
    if (e == hipSuccess) e = hipMalloc(&w->ustart_dev, sizeof(int) * (ds + npairs + 1));
    if (e == hipSuccess) e = hipMemcpy(w->ustart_dev, w->ustart_host, sizeof(int) * (ds + npairs + 1), hipMemcpyHostToDevice);
    free(h);
    return e == hipSuccess ? 0 : -1;
}

// Work list 7, "balanced runs" (round 5): one trajectory of a large training set (the B = 1 solver callbacks) on 256x64 tiles launches
// n equal workgroups on `slots` workgroup slots -- N = 4096, ds = 6: 3264 on 1024, i.e. 3.19 generations of 30 us workgroups, the last one
// a fifth full (30 of 120 us, profiles/r04/fused_stamps_c4_b1.txt), and every workgroup pays a 7 us prologue for 23 us of columns.
// Here every 256-row tile row of every GP is cut into RUNS of up to 256 columns -- the longest run length L for which the whole
// list still fits ONE generation: sum over (GP, tile row) of ceil(span / L) <= slots -- each run its own work item {unit, i0, j0, j1}
// (multiples of 8 columns; step_fused.h, Q = 256, reads a run's length from its item): one generation per trajectory, a third of the
// prologues.  (A first attempt that only split the LAST partial generation into 256x16 tiles was slower: 3.65 -> 4.23 ms,
// profiles/r05/ab15_split_tail.txt.)  Returns 1 when the list is not worth building (the plain list is about one generation already).
// (host part: the items {unit, i0, j0, j1} as a malloc'd array, ustart[ds + 1]; returns the item count, 0 when the list is not worth building,
// -1 on allocation failure)
static int runs_host(int Np, int ds, int slots, int** items_out, int* ustart) {
    const int it = 256;
    int n64 = 0;
    for (int i0 = 0; i0 < Np; i0 += it) n64 += (Np - (i0 / 64) * 64 + 63) / 64;
    n64 *= ds;
    if (slots < 64 || n64 <= slots + slots / 10) return 0;
    auto count = [&](int L) { int c = 0; for (int i0 = 0; i0 < Np; i0 += it) { const int span = Np - i0; c += (span + L - 1) / L; } return c * ds; };
    int L = 8;
    while (L < 256 && count(L) > slots) L += 8;
    if (count(L) > slots) return 0;                                      // more than one generation even with the longest runs
    const int n = count(L);
    int* h = (int*)malloc(sizeof(int) * 4 * (size_t)n);
    if (!h) return -1;
    int k = 0;
    for (int u = 0; u < ds; ++u) {
        ustart[u] = k;
        for (int i0 = 0; i0 < Np; i0 += it) {
            const int span = Np - i0, runs = (span + L - 1) / L;
            int j = i0;
            for (int q = 0; q < runs; ++q) {                            // equal runs, multiples of 8 columns, the remainder spread over the first ones
                int len = ((span / runs) / 8) * 8;
                const int extra = (span - len * runs) / 8;              // runs that take 8 more columns
                if (q < extra) len += 8;
                if (q == runs - 1) len = Np - j;
                h[4 * k] = u; h[4 * k + 1] = i0; h[4 * k + 2] = j; h[4 * k + 3] = j + len; ++k;
                j += len;
            }
        }
    }
    ustart[ds] = k;
    *items_out = h;
    return k;
}

static int build_worklist_runs(int Np, int ds, int slots, gpmpc_worklist* w) {
    int* h = nullptr;
    const int k = runs_host(Np, ds, slots, &h, w->ustart_host);
    if (k == 0) return 1;
    if (k < 0) return -1;
    w->contiguous = 1;
    w->it = 256; w->waves = 4; w->jt = 256; w->nunits = ds; w->nwork = k;
    w->perm_dev = nullptr;
    hipError_t e = hipMalloc(&w->work_dev, sizeof(int) * 4 * (size_t)k);
    if (e == hipSuccess) e = hipMemcpy(w->work_dev, h, sizeof(int) * 4 * (size_t)k, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&w->ustart_dev, sizeof(int) * (ds + 1));
    if (e == hipSuccess) e = hipMemcpy(w->ustart_dev, w->ustart_host, sizeof(int) * (ds + 1), hipMemcpyHostToDevice);
    free(h);
    return e == hipSuccess ? 0 : -1;
}

// Host-side views of two pieces of launch geometry, for tests that run without a GPU (include/gpmpc.h, "launch geometry")
extern "C" int gpmpc_debug_run_list(int n_padded, int state_dim, int slots, int* items_out, int capacity, int* n_items) {
    if (n_padded < 64 || n_padded % 64 || state_dim < 1 || state_dim > GPMPC_MAX_DS || !n_items || capacity < 0 || (capacity > 0 && !items_out)) return GPMPC_E_ARG;
    int* h = nullptr; int ust[GPMPC_MAX_DS + 1];
    const int k = runs_host(n_padded, state_dim, slots, &h, ust);
    if (k < 0) return GPMPC_E_ALLOC;
    *n_items = k;
    if (k > 0) { if (k <= capacity) memcpy(items_out, h, sizeof(int) * 4 * (size_t)k); free(h); }
    return (k > capacity && capacity > 0) ? GPMPC_E_ARG : GPMPC_OK;
}
extern "C" int gpmpc_debug_xcd_order(int linear_id, int grid_x, int n_tile, int n_traj, int* traj, int* column) {
    if (!traj || !column || grid_x < 1 || n_tile < 1 || n_tile > grid_x || n_traj < 1 || linear_id < 0 || (long)linear_id >= (long)grid_x * n_traj) return GPMPC_E_ARG;
    unsigned bx = 0;
    gpmpc_xcd_remap(linear_id, grid_x, n_tile, n_traj, traj, &bx);
    *column = (int)bx;
    return GPMPC_OK;
}

void gpmpc_read_tuning(gpmpc_tuning* t) {
    auto geti = [](const char* name, int unset) { const char* ev = getenv(name); return ev ? atoi(ev) : unset; };
    t->pair_sb = geti("GPMPC_PAIR_SB", -1);
    t->tiling = geti("GPMPC_TILING", -1);
    t->tb = geti("GPMPC_PAIR_TB", 0);
    t->rgroup = geti("GPMPC_RGROUP", 0);
    t->no_first = getenv("GPMPC_NO_FIRST") ? 1 : 0;
    t->fused = geti("GPMPC_FUSED", -1);
    t->no_xcd_sort = getenv("GPMPC_NO_XCD_SORT") ? 1 : 0;
    t->hchunks = geti("GPMPC_HEAD_CHUNKS", -1);
    t->sbf_min = geti("GPMPC_SBF_MIN", 0);
    t->shared = geti("GPMPC_SHARED", -1);
    t->colunroll = geti("GPMPC_SB_UNROLL", -1);
    t->split = geti("GPMPC_SPLIT", -1);
    t->fused_sb = geti("GPMPC_FUSED_SB", -1);
    t->persist = geti("GPMPC_PERSIST", -1);
    t->xcdmap = geti("GPMPC_XCDMAP", -1);
    t->fc_shared = geti("GPMPC_FC_SHARED", -1);
    t->fc_form = geti("GPMPC_FC_FORM", -1);
    t->fc_tiling = geti("GPMPC_FC_TILING", -1);
    t->fc_rsplit = geti("GPMPC_FC_RSPLIT", 0);
    t->fc_cu = geti("GPMPC_FC_CU", 0);
}

extern "C" int gpmpc_pack_reload_tuning(gpmpc_pack* p) {
    if (!p) return GPMPC_E_ARG;
    if (int rc_dev = gpmpc_check_device(p)) return rc_dev;   // the caches below hold streams / buffers of the pack's device
    const int keep = p->tune.no_xcd_sort;             // baked into the work list at creation
    gpmpc_read_tuning(&p->tune);
    p->tune.no_xcd_sort = keep;
    if (p->graph_cache) { gpmpc_graph_cache_free(p->graph_cache); p->graph_cache = nullptr; }   // captured under the old plan
    if (p->cb_cache) { gpmpc_cb_cache_free(p->cb_cache); p->cb_cache = nullptr; }
    gpmpc_tuned_clear(p->tuned);
    return GPMPC_OK;
}

extern "C" int gpmpc_pack_create(gpmpc_pack** out, int n_train, int state_dim, int action_dim) {
    if (!out || n_train < 1 || state_dim < 1 || action_dim < 0) return GPMPC_E_ARG;
    const int D = state_dim + action_dim;
    if (D > GPMPC_MAX_D || state_dim > GPMPC_MAX_DS || D < 1) return GPMPC_E_ARG;
    gpmpc_pack* p = (gpmpc_pack*)calloc(1, sizeof(gpmpc_pack));
    if (!p) return GPMPC_E_ALLOC;
    p->N = n_train; p->Np = ((n_train + 63) / 64) * 64; p->ds = state_dim; p->da = action_dim; p->D = D;
    gpmpc_read_tuning(&p->tune);
    p->lock = gpmpc_lock_create();
    if (!p->lock) { free(p); return GPMPC_E_ALLOC; }
    if (hipError_t ed = hipGetDevice(&p->device); ed != hipSuccess) { gpmpc_set_error("gpmpc_pack_create: hipGetDevice", ed); gpmpc_lock_destroy(p->lock); free(p); return GPMPC_E_LAUNCH; }
    if (hipDeviceGetAttribute(&p->num_cu, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || p->num_cu < 1) p->num_cu = 256;
    for (int a = 0; a < state_dim; ++a)
        for (int b = a + 1; b < state_dim; ++b) { p->pair_a[p->npairs] = a; p->pair_b[p->npairs] = b; ++p->npairs; }
    const size_t Np = p->Np;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipMalloc(&p->X, sizeof(double) * Np * D);
    if (e == hipSuccess) e = hipMalloc(&p->XT, sizeof(double) * Np * D);
    if (e == hipSuccess) e = hipMalloc(&p->beta, sizeof(double) * Np * state_dim);
    if (e == hipSuccess) e = hipMalloc(&p->M, sizeof(double) * Np * Np * state_dim);
    if (e == hipSuccess) e = hipMalloc(&p->lam, sizeof(double) * state_dim * D);
    if (e == hipSuccess) e = hipMalloc(&p->sf, sizeof(double) * state_dim);
    if (e == hipSuccess) e = hipMalloc(&p->ncol_dev, 64);
    if (e == hipSuccess) { const int nc0 = p->Np; e = hipMemcpy(p->ncol_dev, &nc0, sizeof(int), hipMemcpyHostToDevice); p->ncol_host = nc0; }
    if (e == hipSuccess && p->npairs > 0) {
        int hab[2 * GPMPC_MAX_PAIRS];
        for (int k = 0; k < p->npairs; ++k) { hab[2 * k] = p->pair_a[k]; hab[2 * k + 1] = p->pair_b[k]; }
        e = hipMalloc(&p->pair_ab_dev, sizeof(int) * 2 * p->npairs);
        if (e == hipSuccess) e = hipMemcpy(p->pair_ab_dev, hab, sizeof(int) * 2 * p->npairs, hipMemcpyHostToDevice);
    }
    bool ok = e == hipSuccess;
    // 256-row workgroups: 512 / 1024 rows would cut the G-row re-reads 2x / 4x but ran 8 % / 46 % slower on C3 (C4: -1 %).
    // 256x64: scalar-broadcast kernel for small batches of a large N (enough workgroups to fill the chip at B = 1).
    // 64x128: staged kernel for B = 1 of a large N (fewer partial sums to reduce per step: N = 2048 1.30 -> 1.18 ms per
    // rollout; N <= 512 is 10 % slower on it and stays on 64x64).
    // 256x128: between the 256x64 and the 256x256 shapes, two trajectories per wave (round 3: N = 2048, B = 24 / 32 +19 / +21 %
    // against 256x64, N = 1024, B = 96 / 128 +10 / +14 %; -3...-4 % against 256x256 from ~5 generations of workgroups on).
    // 256x32 / 256x16: the one-launch-per-step form (step_fused.h) on launches that would leave most of the chip without a workgroup
    // (B = 1...4 of a large N): narrower tiles, more and shorter-lived tile workgroups.
    int cfg[7][2] = {{256, 256}, {64, 64}, {256, 64}, {64, 128}, {256, 128}, {256, 32}, {256, 16}};
    if (const char* ev = getenv("GPMPC_JT0")) { const int v = atoi(ev); if (v >= 64 && v % 64 == 0) cfg[0][1] = v; }   // A/B: column extent of the large tiles
    for (int mode = 0; mode < 2 && ok; ++mode)
        for (int k = 0; k < 7 && ok; ++k) {
            if (mode == 1 && k >= 2 && k != 2 && k != 4) continue;      // full-covariance units: 256x256, 64x64, and 256x64 / 256x128 for small batches (fullcov.hip)
            ok = build_worklist(p->Np, cfg[k][0], cfg[k][1], state_dim, mode ? p->npairs : 0,
                                mode == 0 && (k == 0 || k == 4) && !p->tune.no_xcd_sort, &p->wl[mode][k]) == 0;
        }
    // Measured (tools/lib_ab.py, profiles/r05/ab17_runs.txt; ms per rollout of ONE trajectory, 256x64 tiles | runs): N = 4096, ds = 6, da = 1, H = 30
    // 3.63 | 3.34; N = 3584, ds = 5, da = 1, H = 20 1.48 | 1.35; N = 4096, ds = 4, da = 2 1.44 | 1.37; but ds = 4, da = 1 (D = 5, five waves per
    // SIMD) 1.25 | 1.35: there the plain list already streams the weights at 4.3 TB/s, the rate of the full-covariance pair kernel (the
    // practical ceiling of this access pattern), and longer workgroups only lose its overlap of prologues with loops.  From D = 6.
    if (ok && D >= 6 && !getenv("GPMPC_NO_RUNS")) {
        // (workgroup slots of the one-launch form on 256-row tiles: 4 waves per SIMD from D = 6, 5 below -- step_fused.h's launch bounds)
        const int occ = 4;
        // (the launch carries 2 ds more workgroups behind the tiles -- mean sums and finish, step_fused.h --: they need slots of the same generation)
        if (build_worklist_runs(p->Np, state_dim, p->num_cu * occ - 2 * state_dim, &p->wl[0][7]) < 0) ok = false;
    }
    // shared-lambda work lists (pair_kernel_sbs.h): "units" are groups of sh_ng GPs
    if (ok && state_dim >= 2 && action_dim >= 1 && action_dim <= 2) {
        p->sh_ng = gpmpc_sbs_group(state_dim, D);
        if (const char* ev = getenv("GPMPC_SHARED_NG")) { const int v = atoi(ev); if (v >= 2 && v <= 4 && v <= state_dim) p->sh_ng = v; }   // A/B (at pack creation)
        const int groups = (state_dim + p->sh_ng - 1) / p->sh_ng;
        ok = build_worklist(p->Np, 256, cfg[0][1], groups, 0, !p->tune.no_xcd_sort, &p->wl_sh[0], true, &p->sh_tiles[0]) == 0 &&
             build_worklist(p->Np, 256, 64, groups, 0, false, &p->wl_sh[1], true, &p->sh_tiles[1]) == 0 &&
             build_worklist(p->Np, 256, 128, groups, 0, !p->tune.no_xcd_sort, &p->wl_sh[2], true, &p->sh_tiles[2]) == 0;
        if (ok && p->sh_ng > 2 && state_dim % 2 == 0) {
            int tiles2 = 0;
            ok = build_worklist(p->Np, 256, 64, state_dim / 2, 0, false, &p->wl_sh[3], true, &tiles2) == 0;
        }
    }
    if (!ok) { gpmpc_set_error("gpmpc_pack_create", e); gpmpc_pack_destroy(p); return GPMPC_E_ALLOC; }
    *out = p;
    return GPMPC_OK;
}

// Re-use a pack for a training set of a different size with the SAME padded size (the closed loop grows the set by one
// observation per step, src/simulator.py:55: the padded size changes every 64 steps only).  Everything allocated at creation
// depends on the padded size alone, and so do the rollout's launches (rows >= N carry zero weights; no rollout kernel reads
// N): the captured launch sequences of the pack stay valid and are replayed on the refilled buffers.
extern "C" int gpmpc_pack_resize(gpmpc_pack* p, int n_train) {
    if (!p || n_train < 1) return GPMPC_E_ARG;
    if (int rc_dev = gpmpc_check_device(p)) return rc_dev;
    if (((n_train + 63) / 64) * 64 != p->Np) return GPMPC_E_ARG;
    p->N = n_train;
    p->built = 0;
    return GPMPC_OK;
}

extern "C" int gpmpc_pack_destroy(gpmpc_pack* p) {
    if (!p) return GPMPC_OK;
    if (p->X) (void)hipFree(p->X);
    if (p->XT) (void)hipFree(p->XT);
    if (p->beta) (void)hipFree(p->beta);
    if (p->M) (void)hipFree(p->M);
    if (p->lam) (void)hipFree(p->lam);
    if (p->sf) (void)hipFree(p->sf);
    if (p->ncol_dev) (void)hipFree(p->ncol_dev);
    gpmpc_graph_cache_free(p->graph_cache);
    gpmpc_cb_cache_free(p->cb_cache);
    gpmpc_tuned_free(p->tuned);
    gpmpc_lock_destroy(p->lock);
    if (p->pair_ab_dev) (void)hipFree(p->pair_ab_dev);
    if (p->fcs_rows) (void)hipFree(p->fcs_rows);
    for (int k = 0; k < 8; ++k) for (int q = 0; q < 3; ++q) if (p->fcs_ustart_dev[k][q]) (void)hipFree(p->fcs_ustart_dev[k][q]);
    if (p->fcs_tiles256_dev) (void)hipFree(p->fcs_tiles256_dev);
    for (int mode = 0; mode < 2; ++mode)
        for (int k = 0; k < 8; ++k) {
            if (p->wl[mode][k].work_dev) (void)hipFree(p->wl[mode][k].work_dev);
            if (p->wl[mode][k].perm_dev) (void)hipFree(p->wl[mode][k].perm_dev);
            if (p->wl[mode][k].ustart_dev) (void)hipFree(p->wl[mode][k].ustart_dev);
        }
    for (int k = 0; k < 4; ++k) {
        if (p->wl_sh[k].work_dev) (void)hipFree(p->wl_sh[k].work_dev);
        if (p->wl_sh[k].perm_dev) (void)hipFree(p->wl_sh[k].perm_dev);
        if (p->wl_sh[k].ustart_dev) (void)hipFree(p->wl_sh[k].ustart_dev);
    }
    free(p);
    return GPMPC_OK;
}

// beta given directly: [N][ds] -> [ds][Np] zero padded
__global__ void k_pack_copy_beta(const double* __restrict__ Bsrc, int N, int Np, int ds, double* __restrict__ beta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, a = blockIdx.y;
    if (i < Np) beta[(size_t)a * Np + i] = i < N ? Bsrc[(size_t)i * ds + a] : 0.0;
}

// ld / gstride: row stride of a Ky_inv matrix and the stride from one GP's matrix to the next, in doubles (0, 0 = packed [ds][N][N];
// gstride 0 with ld > 0: ONE matrix for every GP -- GPs with identical hyper-parameters and inputs share Ky_inv, src/gpr.py:159-171)
static int pack_build_impl(gpmpc_pack* p, const double* X_dev, const double* Y_dev, bool y_is_beta,
                           const double* Ky_inv_dev, const double* lambdas_host, const double* sigma_f_host,
                           void* stream, size_t ld = 0, size_t gstride = 0) {
    if (!p || !X_dev || !Y_dev || !lambdas_host || !sigma_f_host) return GPMPC_E_ARG;
    if (ld == 0) { ld = (size_t)p->N; gstride = (size_t)p->N * p->N; }
    if (ld < (size_t)p->N) return GPMPC_E_ARG;
    if (int rc_dev = gpmpc_check_device(p)) return rc_dev;
    if (!Ky_inv_dev && !y_is_beta) return GPMPC_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    for (int a = 0; a < p->ds; ++a) {
        p->sf_host[a] = sigma_f_host[a];
        for (int k = 0; k < p->D; ++k) {
            p->lam_host[a][k] = lambdas_host[a * p->D + k];
            if (!(p->lam_host[a][k] > 0.0)) return GPMPC_E_ARG;
        }
    }
    // every GP with bit-identical length-scales (the reference's experiments): the rollout may share exponent and exp
    // across the GPs of a pair (pair_kernel_sbs.h)
    const int shared_before = p->shared_lambda;
    p->shared_lambda = p->sh_ng > 0 ? 1 : 0;
    for (int a = 1; a < p->ds && p->shared_lambda; ++a)
        if (memcmp(p->lam_host[a], p->lam_host[0], sizeof(double) * p->D) != 0) p->shared_lambda = 0;
    // a captured launch sequence carries the kernel choice of the build it was captured under: the shared-lambda kernel
    // replayed on a pack whose lambdas have since become distinct would be WRONG (not just slow)
    if (p->shared_lambda != shared_before) {
        gpmpc_graph_cache_invalidate(p->graph_cache);
        gpmpc_cb_cache_invalidate(p->cb_cache);
        gpmpc_tuned_clear(p->tuned);                         // measured plans name the kernel family too
    }
    // hyper-parameters are tiny: as kernel arguments (consumed before this call returns, gpmpc_upload_small)
    if (int rcu = gpmpc_upload_small(p->lam, lambdas_host, sizeof(double) * p->ds * p->D, s)) return rcu;
    if (int rcu = gpmpc_upload_small(p->sf, sigma_f_host, sizeof(double) * p->ds, s)) return rcu;
    hipLaunchKernelGGL(k_pack_points, dim3((p->Np + 255) / 256), dim3(256), 0, s, X_dev, p->N, p->Np, p->D, p->X, p->XT);
    if (y_is_beta)
        hipLaunchKernelGGL(k_pack_copy_beta, dim3((p->Np + 255) / 256, p->ds), dim3(256), 0, s, Y_dev, p->N, p->Np, p->ds, p->beta);
    else
        hipLaunchKernelGGL(k_pack_beta, dim3((p->Np + 3) / 4, p->ds), dim3(256), 0, s, Ky_inv_dev, ld, gstride, Y_dev, p->N, p->Np, p->ds, p->beta);
    if (Ky_inv_dev)
        hipLaunchKernelGGL(k_pack_weights, dim3(p->Np / 32, p->Np / 32, p->ds), dim3(256), 0, s,
                           Ky_inv_dev, ld, gstride, p->beta, p->XT, p->lam, p->sf, p->N, p->Np, p->D, p->M);
    else
        GPMPC_HIP(hipMemsetAsync(p->M, 0, sizeof(double) * (size_t)p->Np * p->Np * p->ds, s));
    if (p->fullcov && p->npairs > 0)
        hipLaunchKernelGGL(k_pack_cross, dim3((p->Np + 255) / 256, p->Np, p->npairs), dim3(256), 0, s, p->beta, p->XT, p->lam,
                           p->sf, p->pair_ab_dev, p->N, p->Np, p->D, p->ds, p->M + (size_t)p->ds * p->Np * p->Np);
    if (int rcf = fcs_refresh(p, s)) return rcf;
    {   // the column count that carries weight (traj_persist.h): re-sent when it changes (every 8th observation of a growing set)
        int nc = ((p->N + 7) / 8) * 8;
        if (nc > p->Np) nc = p->Np;
        if (nc != p->ncol_host) {
            if (int rcu = gpmpc_upload_small(p->ncol_dev, &nc, sizeof(int), s)) return rcu;
            p->ncol_host = nc;
        }
    }
    GPMPC_HIP(hipGetLastError());
    p->built = 1;
    return GPMPC_OK;
}

// Allocate and fill the cross-covariance weight matrices (needed by the full-covariance rollout and by the
// analytic cross-covariance Jacobians of gpmpc_moment_match).  Later gpmpc_pack_build* calls keep them current.
extern "C" int gpmpc_pack_enable_fullcov(gpmpc_pack* p, void* stream) {
    if (!p) return GPMPC_E_ARG;
    if (int rc_dev = gpmpc_check_device(p)) return rc_dev;
    if (!p) return GPMPC_E_ARG;
    if (p->fullcov || p->npairs == 0) { p->fullcov = 1; return GPMPC_OK; }
    hipStream_t s = (hipStream_t)stream;
    const size_t mat = sizeof(double) * (size_t)p->Np * p->Np;
    double* Mnew = nullptr;
    if (hipMalloc(&Mnew, mat * (p->ds + p->npairs)) != hipSuccess) return GPMPC_E_ALLOC;
    GPMPC_HIP(hipMemcpyAsync(Mnew, p->M, mat * p->ds, hipMemcpyDeviceToDevice, s));
    GPMPC_HIP(hipStreamSynchronize(s));
    (void)hipFree(p->M);
    p->M = Mnew;
    p->fullcov = 1;
    gpmpc_graph_cache_free(p->graph_cache); p->graph_cache = nullptr;    // p->M moved
    gpmpc_cb_cache_free(p->cb_cache); p->cb_cache = nullptr;
    if (p->built) {
        hipLaunchKernelGGL(k_pack_cross, dim3((p->Np + 255) / 256, p->Np, p->npairs), dim3(256), 0, s, p->beta, p->XT, p->lam,
                           p->sf, p->pair_ab_dev, p->N, p->Np, p->D, p->ds, p->M + (size_t)p->ds * p->Np * p->Np);
        GPMPC_HIP(hipGetLastError());
        if (int rcf = fcs_refresh(p, s)) return rcf;
    }
    return GPMPC_OK;
}

extern "C" int gpmpc_pack_build(gpmpc_pack* p, const double* X_dev, const double* Y_dev, const double* Ky_inv_dev,
                                const double* lambdas_host, const double* sigma_f_host, void* stream) {
    return pack_build_impl(p, X_dev, Y_dev, false, Ky_inv_dev, lambdas_host, sigma_f_host, stream);
}

extern "C" int gpmpc_pack_build_strided(gpmpc_pack* p, const double* X_dev, const double* Y_dev, const double* Ky_inv_dev,
                                        size_t ld, size_t gp_stride, const double* lambdas_host, const double* sigma_f_host, void* stream) {
    if (!Ky_inv_dev || ld == 0) return GPMPC_E_ARG;
    return pack_build_impl(p, X_dev, Y_dev, false, Ky_inv_dev, lambdas_host, sigma_f_host, stream, ld, gp_stride);
}

extern "C" int gpmpc_pack_build_beta(gpmpc_pack* p, const double* X_dev, const double* beta_dev, const double* Ky_inv_dev,
                                     const double* lambdas_host, const double* sigma_f_host, void* stream) {
    return pack_build_impl(p, X_dev, beta_dev, true, Ky_inv_dev, lambdas_host, sigma_f_host, stream);
}

extern "C" int gpmpc_pack_dims(const gpmpc_pack* p, int* n, int* np, int* ds, int* da) {
    if (!p) return GPMPC_E_ARG;
    if (n) *n = p->N;
    if (np) *np = p->Np;
    if (ds) *ds = p->ds;
    if (da) *da = p->da;
    return GPMPC_OK;
}
extern "C" int gpmpc_pack_shared_lambda(const gpmpc_pack* p) {
    if (!p) return GPMPC_E_ARG;
    if (!p->built) return GPMPC_E_STATE;
    return p->shared_lambda;
}
extern "C" int gpmpc_pack_export(const gpmpc_pack* p, double* beta_out, double* weights_out, void* stream) {
    if (!p) return GPMPC_E_ARG;
    if (!p->built) return GPMPC_E_STATE;
    hipStream_t s = (hipStream_t)stream;
    const size_t Np = p->Np;
    if (beta_out) GPMPC_HIP(hipMemcpyAsync(beta_out, p->beta, sizeof(double) * Np * p->ds, hipMemcpyDeviceToDevice, s));
    if (weights_out) GPMPC_HIP(hipMemcpyAsync(weights_out, p->M, sizeof(double) * Np * Np * p->ds, hipMemcpyDeviceToDevice, s));   // variance units only
    return GPMPC_OK;
}

extern "C" int gpmpc_build_ky(int n, int D, const double* X_dev, const double* lambdas_host, double sigma_f,
                              double noise_var, double* Kf_dev, double* Ky_dev, void* stream) {
    if (n < 1 || D < 1 || D > GPMPC_MAX_D || !X_dev || !lambdas_host || !Ky_dev) return GPMPC_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    double* lam = nullptr;
    GPMPC_HIP(hipMallocAsync((void**)&lam, sizeof(double) * D, s));
    if (int rcu = gpmpc_upload_small(lam, lambdas_host, sizeof(double) * D, s)) { (void)hipFreeAsync(lam, s); return rcu; }
    hipLaunchKernelGGL(k_build_ky, dim3((n + 255) / 256, n), dim3(256), 0, s, X_dev, n, D, lam, sigma_f * sigma_f,
                       noise_var, Kf_dev, Ky_dev);
    GPMPC_HIP(hipGetLastError());
    GPMPC_HIP(hipFreeAsync(lam, s));
    return GPMPC_OK;
}
