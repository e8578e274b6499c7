// One translation unit per input dimension D (compiled with -DGPMPC_PAIR_D=<D>) so that the
// template instantiations of the pair kernel build in parallel.
#include "pair_kernel.h"
#include "pair_kernel_sb.h"
#include "pair_kernel_sbf.h"
#include "pair_kernel_sbs.h"
#ifndef GPMPC_PAIR_D
#error "compile with -DGPMPC_PAIR_D=<D>"
#endif
template int gpmpc_launch_pair_D<GPMPC_PAIR_D>(bool, bool, int, int, const PairArgs&, hipStream_t);
template int gpmpc_launch_pair_sb_D<GPMPC_PAIR_D>(bool, int, int, int, const PairSbArgs&, hipStream_t);
template int gpmpc_launch_pair_sbf_D<GPMPC_PAIR_D>(bool, int, int, const PairSbfArgs&, hipStream_t);
template int gpmpc_launch_pair_sbs_D<GPMPC_PAIR_D>(bool, int, int, const PairSbsArgs&, hipStream_t);
