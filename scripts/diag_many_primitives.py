import ctypes as C, os, sys, time, math
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import rt_amd as rt
from rt_amd import *
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.RandomState(5)
w = World(); w.lights.append(Light.point_light((-6., 12., -10.), rt.WHITE)); w.objects.append(Shape.plane().set_material(Material(reflective=0.2)))
for _ in range(n):
    r = rng.uniform(0.15, 0.4)
    s = Shape.sphere().set_transform(translate(rng.uniform(-9, 9), rng.uniform(0.3, 5), rng.uniform(-3, 12)) * scale(r, r, r))
    s.set_material(Material(color=tuple(rng.uniform(0.2, 1., 3)), reflective=float(rng.choice([0., 0., 0.3]))))
    w.objects.append(s)
cam = Camera(960, 540, math.pi / 2.5); cam.transform = view_transform((0., 4., -12.), (0., 1.5, 2.), (0., 1., 0.))
cam.render(w, 5)
img = cam.render(w, 5, want_counters=True)
st = img.stats
names = ["xforms", "box tests", "gates", "tri tests", "tri det exits", "tri u exits", "tri v exits", "spheres", "planes", "cubes", "cylinders", "cones", "bvh nodes"]
traced = st.rays_primary + st.rays_reflect + st.rays_refract
print(f"n={n} TLAS={os.environ.get('RT_AMD_TLAS','default')} traced {traced} shadow {st.rays_shadow} ms_trace {st.ms_trace:.2f} walk {st.ms_walk:.2f} queue {st.ms_queue:.2f} shade {st.ms_kshade:.2f} total {st.ms_total:.2f}")
for label, work, k in (("closest", st.work_trace, traced), ("shadow", st.work_shadow, st.rays_shadow)):
    if k: print("  ", label, ", ".join(f"{names[j]} {work[j]/k:.2f}" for j in range(13) if work[j]))
for depth in range(0, 3):
    s = cam.render(w, depth).stats
    print(f"  depth {depth}: rays {s.rays_primary + s.rays_reflect + s.rays_refract} trace {s.ms_trace:.3f} total {s.ms_total:.3f}")
