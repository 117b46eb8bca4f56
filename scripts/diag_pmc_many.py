"""rocprofv3 target: three depth-0 frames of the world of n top-level spheres (scripts/diag_many_primitives.py)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
for _ in range(3):
    cam.render(w, 0)
