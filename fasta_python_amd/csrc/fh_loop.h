// fh_loop.h -- options and state of the FBS loop when it runs on the device: shared by the persistent launch (csrc/fh_run.h) and the chain of
// one-pass launches (csrc/fh_fused.h: k_fused_chain).  Host mirrors: fh_run_opts / fh_run_state of include/fasta_hip.h.
#pragma once
#define FR_HIST 8            // doubles per history record: residual, norm_residual, stepsize, f, objective, backtracks, alpha0, better (+ 2: restarted)
#define FR_WINDOW_MAX 64

struct RunOpts {
  int adaptive, accelerate, backtrack, restart, evaluate_objective, stop_rule, window, max_backtracks;
  double stepsize_shrink, tolerance;
};
struct RunState {            // survives between launches: passed in by value, written back to host-mapped memory at the end
  double tau_next, alpha1, max_residual, best_quality;
  unsigned long long iteration, backtracks;
  int stopped;               // 0 = ran out of steps, 1 = the stop rule fired, 3 = a grid barrier / hand-off timed out: the state is that of the last COMPLETED
                             //     iteration (an attempt writes only the buffers that are NOT x0 / g0 / x_accel0 / z_accel0), tau_next the step the
                             //     interrupted iteration started with
  int xi, ti, bi, pc, gc, zc, last_accel;
  int perm[5];               // which of the five physical n-side buffers sits in X[0], X[1], X[2], P[0], P[1]
  double f_window[FR_WINDOW_MAX];   // f_hist[j] at j % FR_WINDOW_MAX for the last `window` iterations
};
