"""The stencil at the reference example's own size (tv_denoising.py:113-125: 512 x 512) and a few others: Python driver against the library's
host-side loop, the sweep's own HIP-event time next to both (bench.py:tv_small_runs).  Usage: python scripts/probes/tv_small.py [side...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench


class G:
    local_rank = 0


for side in [int(a) for a in sys.argv[1:]] or [512]:
    r = bench.tv_small_runs(G(), side=side)
    for mode in ("adaptive", "accelerated"):
        for d in ("python_driver", "library_loop"):
            x = r[mode][d]
            print(f"{side:5d}^2 {mode:11s} {d:14s}: {x['iterations/s']:8.0f} it/s  {x['us_per_iteration']:7.2f} us/it  launches {x['launches']:4d}  "
                  f"{x['us_per_launch_wallclock']:.2f} us/launch wall, sweep {x['sweep_us_hip_events']:.2f} us (x{x['launch_to_sweep_ratio']:.1f})  backtracks {x['backtracks']}", flush=True)
    print(f"{side:5d}^2 natural run: {json.dumps(r['natural_run'])}", flush=True)
