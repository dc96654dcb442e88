"""Same-box A/B of two builds of libfasta_hip.so on the one-pass TV sweep (8192^2 by default): HIP-event time of fh_step and of
fh_step_accel + fh_commit, each library in its own process, interleaved; xprox of the two builds must be bit-identical.
    python scripts/probes/ab_tv.py <old.so> <new.so> [side]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import os, sys, ctypes, hashlib
sys.path.insert(0, %r)
import numpy as np
from fasta_python_amd import hip
lib, side = sys.argv[1], int(sys.argv[2])
probe = ctypes.CDLL(lib)
for name in list(hip.SIGNATURES):
    if not hasattr(probe, name):
        del hip.SIGNATURES[name]
hip.load_library(lib)
import fasta_python_amd as fa
P = side * side
rng = np.random.RandomState(0)
A = fa.GradDivMap((side, side))
ctx = A.ctx
ctx.set_loss_lsq(rng.standard_normal(P))
ctx.set_prox(hip.PROX_TVBALL)
x0 = rng.standard_normal(2 * P) * 0.7
def timed(fn, reps=20):
    for _ in range(3): fn()
    ctx.timing_reset(); ctx.timing_enable(True)
    for _ in range(reps): fn()
    ctx.timing_enable(False)
    ms, cnt = ctx.timing_get(hip.K_FUSED)
    return ms / cnt
ctx.set_vector(hip.VEC_X0, x0); ctx.init()
s = ctx.step(0.1)
h = hashlib.sha1(ctx.get_vector(hip.VEC_XPROX, 2 * P).tobytes()).hexdigest()[:12]
t = min(timed(lambda: ctx.step(0.1)) for _ in range(3))
ctx.set_vector(hip.VEC_X0, x0); ctx.init()
ctx.step_accel(0.1, 0.0, True); ctx.commit(False)
a = ctx.step_accel(0.1, 0.3, False)
ha = hashlib.sha1(ctx.get_vector(hip.VEC_XPROX, 2 * P).tobytes()).hexdigest()[:12]
def acc():
    ctx.step_accel(0.1, 0.3, True); ctx.commit(False)
ta = min(timed(acc) for _ in range(3))
print("%%.5f %%.5f %%s %%s %%.17g %%.17g" %% (t, ta, h, ha, s[hip.S_DXDG], a[hip.S_DXDG]))
''' % ROOT
old, new = sys.argv[1], sys.argv[2]
side = sys.argv[3] if len(sys.argv) > 3 else "8192"
rows = {}
for tag, lib in (("old", old), ("new", new), ("old", old), ("new", new), ("old", old), ("new", new)):
    out = subprocess.run([sys.executable, "-c", WORKER, lib, side], capture_output=True, text=True)
    if out.returncode:
        print(out.stderr[-2000:]); sys.exit(1)
    t, ta, h, ha, d, da = out.stdout.split()[-6:]
    rows.setdefault(tag, []).append((float(t), float(ta), h, ha, float(d), float(da)))
    print(f"{tag}: plain {float(t):.4f} ms  FISTA {float(ta):.4f} ms  xprox sha1 {h} / {ha}", flush=True)
o, n = rows["old"], rows["new"]
P = int(side) ** 2
bo, bn = min(r[0] for r in o), min(r[0] for r in n)
ao, an = min(r[1] for r in o), min(r[1] for r in n)
print(f"# {side}^2: plain {bo:.4f} -> {bn:.4f} ms ({(bn / bo - 1) * 100:+.1f} %; {40 * P / bn / 1e6:.0f} GB/s, frac {40 * P / bn / 1e6 / 8000:.3f}), "
      f"FISTA {ao:.4f} -> {an:.4f} ms ({(an / ao - 1) * 100:+.1f} %; frac {56 * P / an / 1e6 / 8000:.3f}); xprox bit-identical: "
      f"{o[0][2] == n[0][2] and o[0][3] == n[0][3]}; S_DXDG rel diff {abs(o[0][4] - n[0][4]) / abs(o[0][4]):.1e} / {abs(o[0][5] - n[0][5]) / abs(o[0][5]):.1e}")
