"""Where is the floor of a traversal launch?  rt_color_at at remaining = 0 (one closest-hit launch + one shadow launch)
for n incoherent rays, n from 64 to 750k, and the same with rays that all miss the mesh's world box."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rt_amd as rt, scenes
w, cam = scenes.teapot_scene(64, 36)
cam.render(w)
rng = np.random.default_rng(3)
N = 750000
o = rng.uniform([-8, 0.01, -8], [8, 4, 8], size=(N, 3))
d = rng.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
o2 = o.copy(); o2[:, 1] += 6.; d2 = d.copy(); d2[:, 1] = np.abs(d2[:, 1])   # above the teapot, pointing up: sky only
for name, oo, dd in (("mixed", o, d), ("sky", o2, d2)):
    for n in (64, 1024, 16384, 65536, 262144, 750000):
        best = [1e9, 1e9]
        for _ in range(6):
            rgb, ids, fl, st = w.color_at_batch(oo[:n], dd[:n], remaining=0, want_stats=True)
            best = [min(best[0], st.ms_trace), min(best[1], st.ms_total)]
        print(f"{name:6s} n={n:7d}: k_trace x{st.n_trace_launches} {best[0]*1e3:7.1f} us   whole call on device {best[1]*1e3:7.1f} us   shadow rays {st.rays_shadow}", flush=True)
