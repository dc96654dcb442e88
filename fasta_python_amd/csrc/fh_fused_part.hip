// fh_fused_part.hip -- one group of explicit instantiations of the one-pass kernel (see fh_fused_instances.inc).
// Compiled once per group with -DFH_PART=0..3; fasta_hip.hip only declares these variants `extern template`.
#include "fh_fused.h"

#ifndef FH_PART
#define FH_PART 0          // a bare `hipcc -c fh_fused_part.hip` builds group 0
#endif
#define FH_FUSED_DEFINE(P, PI, T, X, NB, F) template __global__ void k_fused_dense<P, 1, PI, T, X, NB, F>(const FusedP);
#if FH_PART == 0
#define FUSED_INST_0 FH_FUSED_DEFINE
#else
#define FUSED_INST_0(...)
#endif
#if FH_PART == 1
#define FUSED_INST_1 FH_FUSED_DEFINE
#else
#define FUSED_INST_1(...)
#endif
#if FH_PART == 2
#define FUSED_INST_2 FH_FUSED_DEFINE
#else
#define FUSED_INST_2(...)
#endif
#if FH_PART == 3
#define FUSED_INST_3 FH_FUSED_DEFINE
#else
#define FUSED_INST_3(...)
#endif
#include "fh_fused_instances.inc"
