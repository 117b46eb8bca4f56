import ctypes as C, os, sys, time, math
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rt_amd as rt, scenes
from rt_amd import *
for grid in (4, 6):
    w = World(); w.lights.append(Light.point_light((-10., 20., -15.), rt.WHITE))
    w.objects.append(Shape.plane().set_material(Material(color=(0.3, 0.3, 0.35), specular=0.2, reflective=0.6)))
    tea = scenes.read_asset("teapot.obj"); span = 8.0
    for gz in range(grid):
        for gx in range(grid):
            t, _ = rt.obj_to_group(tea)
            t.set_transform(translate((gx - (grid - 1) / 2.) * span, 0., gz * span) * rotate(0.4 * (gx + 2 * gz), Axis.Y) * scale(0.9, 0.9, 0.9))
            t.set_material(Material(color=(0.5, 0.5, 0.9), diffuse=0.7, specular=0.5, shininess=60., reflective=0.15))
            w.objects.append(t)
    cam = Camera(7680, 4320, math.pi / 3.); cam.transform = view_transform((0., 9., -16.), (0., 1.5, span * (grid - 1) / 2.), (0., 1., 0.))
    scene = w.scene(0); cd = cam.desc()
    frame = torch.zeros((4320, 7680, 3), dtype=torch.float64, device="cuda:0")
    opts = rt.RenderOpts(max_depth=8, rank=0, n_ranks=1, want_counters=0, stream=None)
    for _ in range(3): assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None)
    torch.cuda.synchronize()
    print(f"teapots only, grid {grid} at 8K, MIN={os.environ.get('RT_AMD_TLAS_MIN', '-')}: {(time.perf_counter() - t0) / 4 * 1e3:7.3f} ms/frame", flush=True)
    del frame; w.invalidate()
