"""Diagnostic (GPU): C5 at a given grid, 3840x2160: per-ray work counters and per-level kernel times.
usage: python scripts/c5_diag.py <grid>    (RT_AMD_TLAS=0/1 in the environment selects ranges / top-level tree)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rt_amd as rt, scenes
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 13
W, H, D = 3840, 2160, 8
w, cam = scenes.instanced_grid(W, H, grid=grid)
cam.render(w, D)
img = cam.render(w, D, want_counters=True)
st = img.stats
names = ["xforms", "box tests", "gates", "tri tests", "tri det exits", "tri u exits", "tri v exits", "spheres", "planes", "cubes", "cylinders", "cones", "bvh nodes"]
traced = st.rays_primary + st.rays_reflect + st.rays_refract
print(f"grid {grid} TLAS={os.environ.get('RT_AMD_TLAS', 'default')}: traced {traced}, shadow {st.rays_shadow}, ms_trace {st.ms_trace:.2f} ms_total {st.ms_total:.2f}")
for label, work, n in (("closest-hit", st.work_trace, traced), ("shadow", st.work_shadow, st.rays_shadow)):
    if n:
        print(f"  {label:12s} per ray: " + ", ".join(f"{names[k]} {work[k] / n:.2f}" for k in range(13) if work[k]))
prev = None
for depth in range(D + 1):
    s = cam.render(w, depth).stats
    cur = np.array([s.rays_primary + s.rays_reflect + s.rays_refract, s.shaded_nodes, s.ms_trace, s.ms_queue, s.ms_kshade, s.ms_total], dtype=float)
    row = cur if prev is None else cur - prev
    print(f"  level {depth}: rays {int(row[0]):9d} hits {int(row[1]):9d} | trace+shadow {row[2]:7.3f} ms queue {row[3]:6.3f} shade {row[4]:6.3f} | frame so far {cur[5]:.3f} ms")
    prev = cur
