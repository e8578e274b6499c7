// CPU sanitizer driver for the HOST side of libgpmpc_hip.so (tools/run_sanitizers.sh): every entry point of include/gpmpc.h
// is called with invalid arguments -- NULL pointers, dimensions out of range, a pack that was never built -- and must come
// back with a negative GPMPC_E_* code without touching memory it does not own.  Built against the host-only
// AddressSanitizer + UBSan build of the library (make -C gaussian_process_mpc_amd/csrc asan-host).  Runs without a GPU:
// anything that would need one returns GPMPC_E_LAUNCH / GPMPC_E_ALLOC from the failing HIP call instead.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/gpmpc.h"

static int failures = 0;
#define EXPECT_NEG(call) do { long long rc_ = (long long)(call); if (rc_ >= 0) { printf("FAIL %s -> %lld (expected a negative code)\n", #call, rc_); ++failures; } } while (0)
#define EXPECT_ZERO(call) do { long long rc_ = (long long)(call); if (rc_ != 0) { printf("FAIL %s -> %lld (expected 0)\n", #call, rc_); ++failures; } } while (0)

int main() {
    printf("%s; devices visible: %d\n", gpmpc_version(), gpmpc_device_count());
    gpmpc_pack* pk = nullptr;
    double dummy[64] = {0};
    gpmpc_cost_params cp;
    memset(&cp, 0, sizeof(cp));
    EXPECT_NEG(gpmpc_pack_create(nullptr, 10, 2, 1));
    EXPECT_NEG(gpmpc_pack_create(&pk, 0, 2, 1));
    EXPECT_NEG(gpmpc_pack_create(&pk, 10, 0, 1));
    EXPECT_NEG(gpmpc_pack_create(&pk, 10, GPMPC_MAX_DS + 1, 1));
    EXPECT_NEG(gpmpc_pack_create(&pk, 10, 4, GPMPC_MAX_D));
    EXPECT_NEG(gpmpc_pack_create(&pk, 10, 2, -1));
    EXPECT_ZERO(gpmpc_pack_destroy(nullptr));
    EXPECT_NEG(gpmpc_pack_reload_tuning(nullptr));
    EXPECT_NEG(gpmpc_pack_resize(nullptr, 10));
    EXPECT_ZERO(gpmpc_pack_graph_captures(nullptr));
    EXPECT_NEG(gpmpc_pack_build(nullptr, dummy, dummy, dummy, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_pack_build_beta(nullptr, dummy, dummy, nullptr, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_pack_enable_fullcov(nullptr, nullptr));
    EXPECT_NEG(gpmpc_pack_dims(nullptr, nullptr, nullptr, nullptr, nullptr));
    EXPECT_NEG(gpmpc_pack_shared_lambda(nullptr));
    EXPECT_NEG(gpmpc_pack_export(nullptr, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_build_ky(0, 3, dummy, dummy, 1.0, 0.0, nullptr, dummy, nullptr));
    EXPECT_NEG(gpmpc_build_ky(8, GPMPC_MAX_D + 1, dummy, dummy, 1.0, 0.0, nullptr, dummy, nullptr));
    EXPECT_NEG(gpmpc_build_ky(8, 3, nullptr, dummy, 1.0, 0.0, nullptr, dummy, nullptr));
    EXPECT_NEG(gpmpc_build_ky(8, 3, dummy, dummy, 1.0, 0.0, nullptr, nullptr, nullptr));
    EXPECT_ZERO(gpmpc_moment_match_workspace_bytes(nullptr, 4));
    EXPECT_NEG(gpmpc_moment_match(nullptr, 1, dummy, dummy, 0, dummy, dummy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_cost(0, 3, 2, 1, &cp, dummy, dummy, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_cost(1, 0, 2, 1, &cp, dummy, dummy, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_cost(1, 3, GPMPC_MAX_DS + 1, 1, &cp, dummy, dummy, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_cost(1, 3, 2, 1, nullptr, dummy, dummy, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_cost(1, 3, 2, 1, &cp, nullptr, dummy, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_cost_grad(1, 3, 2, 1, &cp, dummy, dummy, dummy, dummy, dummy, nullptr, nullptr, nullptr));      // 1 of 3 derivative outputs
    EXPECT_ZERO(gpmpc_rollout_workspace_bytes(nullptr, 1, 3, 0));
    EXPECT_NEG(gpmpc_rollout(nullptr, 1, 3, dummy, dummy, &cp, 0, nullptr, nullptr, dummy, nullptr, dummy, 64, nullptr));
    EXPECT_ZERO(gpmpc_rollout_jac_workspace_bytes(nullptr, 1, 3));
    EXPECT_NEG(gpmpc_rollout_jac(nullptr, 1, 3, dummy, dummy, dummy, dummy, dummy, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_rollout_vjp(1, 3, 2, 1, nullptr, dummy, dummy, dummy, nullptr, nullptr));
    EXPECT_NEG(gpmpc_rollout_vjp(1, 3, 2, 1, dummy, dummy, dummy, nullptr, nullptr, nullptr));
    EXPECT_NEG(gpmpc_rollout_vjp(0, 3, 2, 1, dummy, dummy, dummy, dummy, nullptr, nullptr));
    EXPECT_NEG(gpmpc_rollout_vjp(1, 3, GPMPC_MAX_DS + 1, 1, dummy, dummy, dummy, dummy, nullptr, nullptr));
    EXPECT_NEG(gpmpc_rollout_vjp(1, 3, 2, 0, dummy, dummy, dummy, dummy, nullptr, nullptr));
    EXPECT_ZERO(gpmpc_rollout_fullcov_workspace_bytes(nullptr, 1, 3, 0));
    EXPECT_NEG(gpmpc_rollout_fullcov(nullptr, 1, 3, dummy, dummy, &cp, 0, dummy, dummy, dummy, nullptr, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_objective_gradient(nullptr, 3, dummy, dummy, &cp, 1, dummy, nullptr));
    // round-4 entry points
    char text[256];
    EXPECT_NEG(gpmpc_plan_describe(nullptr, 1, 3, 0, text, sizeof(text)));
    EXPECT_NEG(gpmpc_rollout_fullcov_describe(nullptr, 1, 3, 0, text, sizeof(text)));
    EXPECT_NEG(gpmpc_pack_autotune(nullptr, 1, 3, 0, text, sizeof(text)));
    EXPECT_NEG(gpmpc_pack_autotune_clear(nullptr));
    EXPECT_NEG(gpmpc_pack_build_strided(nullptr, dummy, dummy, dummy, 8, 64, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_store_host(nullptr, dummy, 8, nullptr));
    EXPECT_NEG(gpmpc_store_host(dummy, nullptr, 8, nullptr));
    EXPECT_ZERO(gpmpc_gp_append_workspace_bytes(0, 3));
    EXPECT_NEG(gpmpc_gp_append(0, 3, dummy, dummy, dummy, 1.0, 0.0, dummy, dummy, 8, dummy, 8, dummy, dummy, dummy, 8, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_gp_append(4, GPMPC_MAX_D + 1, dummy, dummy, dummy, 1.0, 0.0, dummy, dummy, 8, dummy, 8, dummy, dummy, dummy, 8, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_gp_append(4, 3, nullptr, dummy, dummy, 1.0, 0.0, dummy, dummy, 8, dummy, 8, dummy, dummy, dummy, 8, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_gp_append(4, 3, dummy, dummy, dummy, 1.0, 0.0, dummy, dummy, 2, dummy, 8, dummy, dummy, dummy, 8, dummy, 64, nullptr));     // leading dimension < n
    EXPECT_ZERO(gpmpc_timing_enable(0));
    {   // launch geometry, host-side views (round 5): valid calls exercise the host code under the sanitizers, invalid ones return error codes
        int n_items = -1, traj = -1, col = -1;
        EXPECT_ZERO(gpmpc_debug_run_list(4096, 6, 1012, nullptr, 0, &n_items));
        if (n_items > 0) {
            int* items = (int*)malloc(sizeof(int) * 4 * (size_t)n_items);
            EXPECT_ZERO(gpmpc_debug_run_list(4096, 6, 1012, items, n_items, &n_items));
            free(items);
        }
        EXPECT_ZERO(gpmpc_debug_run_list(320, 4, 1016, nullptr, 0, &n_items));
        EXPECT_NEG(gpmpc_debug_run_list(100, 4, 1016, nullptr, 0, &n_items));
        EXPECT_NEG(gpmpc_debug_run_list(4096, 0, 1012, nullptr, 0, &n_items));
        EXPECT_NEG(gpmpc_debug_run_list(4096, 6, 1012, nullptr, 0, nullptr));
        for (int L = 0; L < (24 + 8) * 5; ++L) EXPECT_ZERO(gpmpc_debug_xcd_order(L, 24 + 8, 24, 5, &traj, &col));
        EXPECT_NEG(gpmpc_debug_xcd_order(-1, 32, 24, 5, &traj, &col));
        EXPECT_NEG(gpmpc_debug_xcd_order(0, 32, 33, 5, &traj, &col));
        EXPECT_NEG(gpmpc_debug_xcd_order(0, 32, 24, 5, nullptr, &col));
    }
    double ms = 0; long long nl = 0;
    EXPECT_ZERO(gpmpc_pair_kernel_time(&ms, &nl, 1));
    EXPECT_NEG(gpmpc_pair_kernel_time_class(-1, &ms, &nl));
    EXPECT_NEG(gpmpc_pair_kernel_time_class(99, &ms, &nl));
    EXPECT_ZERO(gpmpc_pair_kernel_time_class(2, &ms, &nl));
    EXPECT_NEG(gpmpc_matvec(0, 4, dummy, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_matvec(4, 4, nullptr, dummy, dummy, nullptr));
    EXPECT_NEG(gpmpc_predict(0, 3, dummy, dummy, 1.0, dummy, dummy, 0.0, 1, dummy, dummy, dummy, dummy, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_predict(4, GPMPC_MAX_D + 1, dummy, dummy, 1.0, dummy, dummy, 0.0, 1, dummy, dummy, dummy, dummy, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_predict(4, 3, nullptr, dummy, 1.0, dummy, dummy, 0.0, 1, dummy, dummy, dummy, dummy, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_kinv_append(0, dummy, dummy, 1.0, dummy, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_kinv_append(4, nullptr, dummy, 1.0, dummy, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_ml_grad(0, 3, dummy, dummy, dummy, dummy, dummy, 1.0, 0.0, dummy, dummy, 64, nullptr));
    EXPECT_NEG(gpmpc_ml_grad(4, 3, nullptr, dummy, dummy, dummy, dummy, 1.0, 0.0, dummy, dummy, 64, nullptr));
    // a pack cannot be created without a device: the failing HIP call must surface as a code, with its text recorded
    int rc = gpmpc_pack_create(&pk, 32, 2, 1);
    if (rc == GPMPC_OK) {                       // (a GPU is present after all: exercise the "not built" state and clean up)
        std::vector<double> ws(1 << 16);
        EXPECT_NEG(gpmpc_rollout(pk, 1, 3, dummy, dummy, &cp, 0, nullptr, nullptr, dummy, nullptr, ws.data(), ws.size() * 8, nullptr));
        EXPECT_NEG(gpmpc_pack_shared_lambda(pk));
        EXPECT_ZERO(gpmpc_pack_destroy(pk));
    } else if (rc >= 0) { printf("FAIL gpmpc_pack_create -> %d\n", rc); ++failures; }
    else printf("gpmpc_pack_create without a device: %d (%s)\n", rc, gpmpc_last_error());
    printf(failures ? "abi_argcheck: %d FAILURES\n" : "abi_argcheck: all argument checks returned error codes (%d failures)\n", failures);
    return failures ? 1 : 0;
}
