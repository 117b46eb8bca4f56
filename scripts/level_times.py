"""Diagnostic (GPU): per-recursion-level ray counts and kernel times of a bench workload, by differencing the rt_stats of
renders at depth 0..D (level-by-level walk, one chain, HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd"))
import numpy as np
import rt_amd as rt
import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "teapot"
W, H, D = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 5)
builders = {"teapot": scenes.teapot_scene, "teapot_low": scenes.teapot_low, "mixed": scenes.mixed_scene, "grid": scenes.instanced_grid}
w, cam = builders[name](W, H)
cam.render(w, D)
prev = None
R = 5
for depth in range(D + 1):
    acc = None
    for _ in range(R):
        s = cam.render(w, depth).stats
        row = np.array([s.rays_primary + s.rays_reflect + s.rays_refract, s.shaded_nodes, s.rays_container, s.ms_trace, s.ms_walk,
                        s.ms_queue, s.ms_kshade, s.ms_total], dtype=float)
        acc = row if acc is None else acc + row
    cur = acc / R
    row = cur if prev is None else cur - prev
    print(f"level {depth}: rays {int(row[0]):8d} hits {int(row[1]):8d} walks {int(row[2]):7d} | trace+shadow {row[3]*1e3:7.1f} us  walk {row[4]*1e3:7.1f} us  "
          f"queue+spawn {row[5]*1e3:6.1f} us  shade {row[6]*1e3:5.1f} us | frame so far {cur[7]:.3f} ms")
    prev = cur
