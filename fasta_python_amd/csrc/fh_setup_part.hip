// fh_setup_part.hip -- one group of explicit instantiations of the multi-column set-up kernel (see fh_setup_instances.inc).
// Compiled once per group with -DFH_PART=0..2 (group 2: float32 storage); fasta_hip.hip only declares these variants `extern template`.
#include "fh_setup.h"

#ifndef FH_PART
#define FH_PART 0
#endif
#define FH_SETUP_DEFINE(P, PI, T, NT, NR) template __global__ void k_setup_dense<P, PI, T, NT, NR>(const SetupP);
#if FH_PART == 0
#define SETUP_INST_0 FH_SETUP_DEFINE
#else
#define SETUP_INST_0(...)
#endif
#if FH_PART == 1
#define SETUP_INST_1 FH_SETUP_DEFINE
#else
#define SETUP_INST_1(...)
#endif
#if FH_PART == 2
#define SETUP_INST_2(P, PI, T, NT, NR) template __global__ void k_setup_dense<P, PI, T, NT, NR, 1>(const SetupP);
#else
#define SETUP_INST_2(...)
#endif
#include "fh_setup_instances.inc"
