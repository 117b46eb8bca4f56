"""Phase times of rt_scene_create (RT_AMD_VERBOSE=1) for a 1 M-triangle soup and for C5 at grid 13, host SAH build and
GPU build.  usage: python scripts/create_phases.py [n_soup]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rt_amd as rt, scenes
from rt_amd import Camera, Light, Material, World, view_transform
os.environ["RT_AMD_VERBOSE"] = "1"
def soup(n, seed=1):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-4, 4, size=(n, 3)); p[:, 1] = np.abs(p[:, 1]) * 0.5
    e = rng.uniform(-0.05, 0.05, size=(n, 2, 3))
    v = np.stack([p, p + e[:, 0], p + e[:, 1]], axis=1).reshape(-1, 3)
    text = "\n".join([f"v {a:.6f} {b:.6f} {c:.6f}" for a, b, c in v] + [f"f {3*i+1} {3*i+2} {3*i+3}" for i in range(n)]) + "\n"
    g, _ = rt.obj_to_group(text)
    g.set_material(Material(color=(0.8, 0.5, 0.3), reflective=0.2))
    w = World(); w.lights.append(Light.point_light((-6., 9., -8.), rt.WHITE)); w.objects.append(g)
    cam = Camera(960, 540, np.pi / 3); cam.transform = view_transform((0., 3., -9.), (0., 1., 0.), (0., 1., 0.))
    return w, cam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
_w, _c = scenes.book_world(32, 32); _c.render(_w)
for name, make in ((f"soup {n}", lambda: soup(n)), ("C5 grid 13", lambda: scenes.instanced_grid(1920, 1080, grid=13))):
    for flag in ("0", "1"):
        os.environ["RT_AMD_DEVICE_BVH"] = flag
        t0 = time.perf_counter(); w, cam = make(); t1 = time.perf_counter()
        w.flatten(); t2 = time.perf_counter()
        print(f"== {name}, RT_AMD_DEVICE_BVH={flag}: parse + groups {t1-t0:.2f} s, flatten {t2-t1:.2f} s", flush=True)
        sc = w.scene(); t3 = time.perf_counter()
        print(f"   rt_scene_create {t3-t2:.3f} s", flush=True)
