// One translation unit per input dimension D (compiled with -DGPMPC_PAIR_D=<D>) so that the
// template instantiations of the pair kernel build in parallel.
#include "pair_kernel.h"
#include "pair_kernel_sb.h"
#include "pair_kernel_sbf.h"
#include "pair_kernel_sbfx.h"
#include "pair_kernel_sbs.h"
#ifndef GPMPC_PAIR_D
#error "compile with -DGPMPC_PAIR_D=<D>"
#endif
template int gpmpc_launch_pair_D<GPMPC_PAIR_D>(bool, bool, int, int, const PairArgs&, hipStream_t);
template int gpmpc_launch_pair_sb_D<GPMPC_PAIR_D>(bool, int, int, int, const PairSbArgs&, hipStream_t);
template int gpmpc_launch_pair_sbf_D<GPMPC_PAIR_D>(bool, int, int, const PairSbfArgs&, hipStream_t);
template int gpmpc_launch_pair_sbfx_D<GPMPC_PAIR_D>(bool, int, const PairSbfxArgs&, hipStream_t);
template int gpmpc_launch_pair_sbs_D<GPMPC_PAIR_D>(bool, int, int, const PairSbsArgs&, hipStream_t);

#if defined(GPMPC_SB_STAMPS) && GPMPC_PAIR_D == 5
// diagnostic build: per-workgroup stamps of the last D = 5 scalar-broadcast pair launch (tools/sb_stamps.py)
extern "C" int gpmpc_debug_sb_stamps(unsigned long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sb_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -3;
}
#endif
