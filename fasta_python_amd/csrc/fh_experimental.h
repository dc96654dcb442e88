// fh_experimental.h -- tuning keys and test hooks that are NOT part of the public C ABI (include/fasta_hip.h).
//
// (1) Experimental forms of the stencil sweep.  Built only with -DFH_EXPERIMENTAL (`make experimental` ->
//     libfasta_hip_experimental.so; tests/test_gpu_experimental.py runs against that library in its own pytest job).  Each of them
//     is bit-identical to the shipped sweep and measured flat or slower on MI355X twice (profiles/r03_tune_tv.txt,
//     profiles/r04_tune_tv.txt), so the shipped library neither compiles nor accepts them:
//       FH_TUNE_TV_ZFREE   = 10   0 = the round-1 one-pass stencil kernels that stream z (k_fused_tv_step / k_fused_tv_accel)
//       FH_TUNE_TV_LDS_PAD = 13   bytes of unused dynamic LDS per workgroup (an occupancy limiter)
//       FH_TUNE_TV_RING    = 14   trips prefetched by LDS-DMA into a per-wave ring of 2 or 3 two-row slots
//       FH_TUNE_TV_SLOTS   = 15   persistent form: at most this many workgroups per CU walk the chunk ids band-major
// (2) Test hooks (always compiled, never documented in the public header; tests/ sets them through fh_set_tuning):
//       FH_TUNE_TEST_HOOKS = 0x7E57   bit 1: member TEAM-1 of team 0 withholds its first partial (the bounded spins must end the
//                                     launch with the timeout word set); bit 2: this context's co-residency probe answers "no";
//                                     bit 4: the multi-workgroup level search's last workgroup withholds its first record (the others
//                                     time out and search alone: same level); bit 8: ... and that fall-back is off (NaN level ->
//                                     FH_E_TIMEOUT); bits 8..15: (value >> 8) & 0xFF = the 1-based attempt of fh_run's persistent
//                                     launch at whose first grid barrier the last workgroup stays away (the launch ends with
//                                     stopped = 3 after the bounded spins; 0 = off).
#pragma once
enum fh_experimental_key {
  FH_TUNE_TV_ZFREE = 10,
  FH_TUNE_TV_LDS_PAD = 13,
  FH_TUNE_TV_RING = 14,
  FH_TUNE_TV_SLOTS = 15,
  FH_TUNE_TEST_HOOKS = 0x7E57
};
#define FH_HOOK_WITHHOLD_PARTIAL 1
#define FH_HOOK_PROBE_SAYS_NO 2
#define FH_HOOK_LEVEL_WITHHOLD 4
#define FH_HOOK_LEVEL_NO_FALLBACK 8
#define FH_HOOK_RUN_ATTEMPT(h) (((h) >> 8) & 0xFF)
