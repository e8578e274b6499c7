// One translation unit per input dimension D (compiled with -DGPMPC_PAIR_D=<D>): instantiations of the fused
// small-batch step kernel (step_fused.h).
#include "step_fused.h"
#ifndef GPMPC_PAIR_D
#error "compile with -DGPMPC_PAIR_D=<D>"
#endif
template int gpmpc_launch_step_fused_D<GPMPC_PAIR_D>(bool, int, int, int, const FusedArgs&, int, hipStream_t);

#if defined(GPMPC_FUSED_STAMPS) && GPMPC_PAIR_D == GPMPC_STAMP_D
// diagnostic build: the stamps of the D = GPMPC_STAMP_D (default 4) instances of this translation unit
extern "C" int gpmpc_debug_wg_times(unsigned long long* host_out) {      // [2][8192]
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fused_wg), sizeof(unsigned long long) * 2 * 8192) == hipSuccess ? 0 : -3;
}
extern "C" int gpmpc_debug_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_fused_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -3;
}
#endif
