// fh_chain_part.hip -- explicit instantiations of k_fused_chain (csrc/fh_fused.h: the one-pass launch whose step size and buffer roles come from a
// device state block and whose finaliser runs the loop's controller), one per row of group 0 of fh_fused_instances.inc: float64, teams of
// 1 / 2 / 4 members, i.e. every width up to n = 16384.  Its own object so that it compiles next to the four groups of k_fused_dense.
#include <hip/hip_runtime.h>
#include "fh_fused.h"
#define FUSED_INST_0(P, PI, T, X, NB, F) template __global__ void k_fused_chain<P, 1, PI, T, X, NB, F>(const FusedP, const ChainP);
#define FUSED_INST_1(...)
#define FUSED_INST_2(...)
#define FUSED_INST_3(...)
#include "fh_fused_instances.inc"
