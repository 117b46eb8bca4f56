"""ms per frame of a union chain of `depth` spheres at 960x540 (csg built by a loop), and C4."""
import ctypes as C, os, sys, time, math
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rt_amd as rt, scenes
from rt_amd import *
def chain_world(depth):
    w = World()
    w.lights.append(Light.point_light((-6., 9., -10.), rt.WHITE))
    w.objects.append(Shape.plane().set_material(Material(reflective=0.2)))
    chain = Shape.sphere().set_transform(translate(-3., 1., 0.) * scale(0.5, 0.5, 0.5))
    for k in range(1, depth):
        nxt = Shape.sphere().set_transform(translate(-3. + 0.3 * k, 1. + 0.3 * math.sin(0.4 * k), 0.2 * math.cos(0.3 * k)) * scale(0.5, 0.5, 0.5))
        chain = Shape.csg("union", chain, nxt)
    chain.set_material(Material(color=(0.8, 0.4, 0.2), reflective=0.1))
    w.objects.append(chain)
    cam = Camera(960, 540, math.pi / 2.5)
    cam.transform = view_transform((0., 3., -8.), (0., 1., 0.), (0., 1., 0.))
    return w, cam
for depth in (6, 10, 20, 40):
    w, cam = chain_world(depth)
    scene = w.scene(0); cd = cam.desc()
    frame = torch.zeros((540, 960, 3), dtype=torch.float64, device="cuda:0")
    opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=1, want_counters=0, stream=None)
    for _ in range(2): assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None)
    torch.cuda.synchronize()
    print(f"union chain of {depth:3d}: {(time.perf_counter() - t0) / 3 * 1e3:9.3f} ms/frame", flush=True)
