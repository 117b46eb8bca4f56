"""Looks for performance cliffs: ms per 960x540 frame (depth 5) of scene shapes a user of the reference plausibly builds."""
import ctypes as C, os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import rt_amd as rt
from rt_amd import *

def base():
    w = World()
    w.lights.append(Light.point_light((-6., 12., -10.), rt.WHITE))
    w.objects.append(Shape.plane().set_material(Material(reflective=0.2)))
    return w

def spheres(n, rng):
    out = []
    for _ in range(n):
        r = rng.uniform(0.15, 0.4)
        s = Shape.sphere().set_transform(translate(rng.uniform(-9, 9), rng.uniform(0.3, 5), rng.uniform(-3, 12)) * scale(r, r, r))
        s.set_material(Material(color=tuple(rng.uniform(0.2, 1., 3)), reflective=float(rng.choice([0., 0., 0.3]))))
        out.append(s)
    return out

def cases():
    rng = np.random.RandomState(5)
    w = base(); w.objects += spheres(2000, rng); yield "2000 top-level spheres", w
    w = base(); g = Shape.group(); [g.add_child(s) for s in spheres(2000, rng)]; w.objects.append(g); yield "one group of 2000 spheres, not divided", w
    w = base(); g = Shape.group(); [g.add_child(s) for s in spheres(2000, rng)]; g.divide(4); w.objects.append(g); yield "the same, divide(4)", w
    w = base(); g = Shape.group(); [g.add_child(s) for s in spheres(40, rng)]
    w.objects.append(Shape.csg("difference", g, Shape.cube().set_transform(translate(0., 2., 4.) * scale(3., 1., 3.)))); yield "difference(group of 40 spheres, cube)", w
    w = base()
    for k in range(12):
        w.objects.append(Shape.glass_sphere().set_transform(translate(0., 1.2, 1.5 * k - 2.) * scale(1., 1., 0.6)))
    yield "12 glass spheres in a row (long n1/n2 lists)", w
    w = base()
    for k in range(6): w.lights.append(Light.point_light((3. * k - 8., 9., -6.), (0.2, 0.2, 0.2)))
    w.objects += spheres(200, rng); yield "200 spheres, 7 lights", w

def more_cases():
    rng = np.random.RandomState(9)
    w = base()
    for k in range(500):
        h = Shape.hexagon().set_transform(translate(rng.uniform(-9, 9), rng.uniform(0.3, 5), rng.uniform(-3, 12)) * rotate(rng.uniform(0, 3), Axis.X) * scale(0.4, 0.4, 0.4))
        w.objects.append(h)
    yield "500 hexagon groups (6 x (sphere + cylinder) each)", w
    w = base()
    for k in range(300):
        c = (rng.uniform(-9, 9), rng.uniform(0.3, 5), rng.uniform(-3, 12))
        a = Shape.cube().set_transform(translate(*c) * scale(0.4, 0.4, 0.4))
        b = Shape.sphere().set_transform(translate(*c) * scale(0.5, 0.5, 0.5))
        w.objects.append(Shape.csg("difference", a, b))
    yield "300 CSG objects (cube - sphere)", w
    import scenes
    w = base()
    t, _ = rt.obj_to_group(scenes.read_asset("teapot.obj"))
    t.set_transform(translate(0., 0., 2.) * scale(1.2, 1.2, 1.2))
    w.objects.append(Shape.csg("difference", t, Shape.sphere().set_transform(translate(0., 1.5, 0.5) * scale(1.6, 1.6, 1.6))))
    yield "difference(teapot mesh, sphere)", w
    w = base()
    for k in range(1000):
        g = Shape.group().set_transform(translate(rng.uniform(-9, 9), rng.uniform(0.3, 5), rng.uniform(-3, 12)))
        g.add_child(Shape.sphere().set_transform(scale(0.25, 0.25, 0.25)))
        g.add_child(Shape.cube().set_transform(translate(0.3, 0., 0.) * scale(0.15, 0.15, 0.15)))
        w.objects.append(g)
    yield "1000 groups of two primitives", w


import itertools
for name, w in itertools.chain(cases(), more_cases()):
    cam = Camera(960, 540, math.pi / 2.5)
    cam.transform = view_transform((0., 4., -12.), (0., 1.5, 2.), (0., 1., 0.))
    t0 = time.perf_counter()
    scene = w.scene(0); cd = cam.desc()
    t_create = time.perf_counter() - t0
    frame = torch.zeros((540, 960, 3), dtype=torch.float64, device="cuda:0")
    opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=1, want_counters=0, stream=None)
    for _ in range(2): assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None) == 0, rt.last_error()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None)
    torch.cuda.synchronize()
    st = rt.Stats(); rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, C.byref(st))
    print(f"{name:48s} {(time.perf_counter() - t0) / 3 * 1e3:9.3f} ms/frame  ({st.total_rays / 1e6:.1f} M rays; scene creation {t_create:.2f} s)", flush=True)
