"""ms per 960x540 frame of n top-level spheres (or small-mesh instances) on a mirror floor, for choosing where the top-level
tree starts to beat nested cull ranges: run with RT_AMD_TLAS_MIN=1 (tree always) and RT_AMD_TLAS=0 (ranges)."""
import ctypes as C, os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rt_amd as rt, scenes
from rt_amd import *
kind = sys.argv[1] if len(sys.argv) > 1 else "spheres"
for n in (6, 8, 12, 16, 24, 32, 48, 64, 96):
    rng = np.random.RandomState(n)
    w = World(); w.lights.append(Light.point_light((-6., 12., -10.), rt.WHITE)); w.objects.append(Shape.plane().set_material(Material(reflective=0.4)))
    for k in range(n):
        pos = (rng.uniform(-6, 6), rng.uniform(0.5, 4), rng.uniform(-2, 8))
        if kind == "spheres":
            r = rng.uniform(0.3, 0.7)
            s = Shape.sphere().set_transform(translate(*pos) * scale(r, r, r))
            s.set_material(Material(color=tuple(rng.uniform(0.2, 1., 3)), reflective=float(rng.choice([0., 0.3]))))
        else:
            s, _ = rt.obj_to_group(scenes.read_asset("teapot-low.obj") if kind == "low" else scenes.read_asset("teapot.obj"))
            s.set_transform(translate(*pos) * rotate(rng.uniform(0, 3), Axis.Y) * scale(0.25, 0.25, 0.25))
            s.set_material(Material(color=tuple(rng.uniform(0.2, 1., 3)), reflective=0.2))
        w.objects.append(s)
    cam = Camera(960, 540, math.pi / 3); cam.transform = view_transform((0., 4., -12.), (0., 1.5, 2.), (0., 1., 0.))
    scene = w.scene(0); cd = cam.desc()
    frame = torch.zeros((540, 960, 3), dtype=torch.float64, device="cuda:0")
    opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=1, want_counters=0, stream=None)
    for _ in range(3): assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None)
    torch.cuda.synchronize()
    print(f"{kind} n={n:3d} TLAS={os.environ.get('RT_AMD_TLAS', '-')} MIN={os.environ.get('RT_AMD_TLAS_MIN', '-')}: {(time.perf_counter() - t0) / 8 * 1e3:7.3f} ms/frame", flush=True)
    w.invalidate()
