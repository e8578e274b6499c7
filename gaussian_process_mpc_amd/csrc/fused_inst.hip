// One translation unit per input dimension D (compiled with -DGPMPC_PAIR_D=<D>): instantiations of the fused
// small-batch step kernel (step_fused.h).
#include "step_fused.h"
#ifndef GPMPC_PAIR_D
#error "compile with -DGPMPC_PAIR_D=<D>"
#endif
template int gpmpc_launch_step_fused_D<GPMPC_PAIR_D>(bool, int, const FusedArgs&, int, hipStream_t);
