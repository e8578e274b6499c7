// One translation unit per input dimension D (compiled with -DGPMPC_PAIR_D=<D>): instantiations of the trajectory-persistent
// whole-horizon rollout kernel (traj_persist.h).
#include "traj_persist.h"
#ifndef GPMPC_PAIR_D
#error "compile with -DGPMPC_PAIR_D=<D>"
#endif
template int gpmpc_launch_persist_D<GPMPC_PAIR_D>(bool, int, int, int, const PersistArgs&, hipStream_t);

#if defined(GPMPC_PERSIST_STAMPS) && GPMPC_PAIR_D == GPMPC_STAMP_D
#ifndef GPMPC_STAMP_D
#error "-DGPMPC_STAMP_D=<D>"
#endif
extern "C" int gpmpc_debug_persist_stamps(unsigned long long* host_out) {      // [64]
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_persist_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -3;
}
#endif
