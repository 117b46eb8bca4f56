"""Scene creation time (program + BVH build + upload) with the host SAH build and with the GPU radix-tree topology
(rt_scene_opts.device_bvh / RT_AMD_DEVICE_BVH=1: the whole tree on the device, rt_lbvh.hip), and the render time each tree gives, on a synthetic mesh of N small triangles plus the headline
teapot scene.  usage: python scripts/bvh_build_time.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rt_amd as rt, scenes
from rt_amd import Camera, Light, Material, World, view_transform

def soup(n, seed=1):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-4, 4, size=(n, 3)); p[:, 1] = np.abs(p[:, 1]) * 0.5
    e = rng.uniform(-0.05, 0.05, size=(n, 2, 3))
    lines = []
    for i in range(n):
        a, b, c = p[i], p[i] + e[i, 0], p[i] + e[i, 1]
        lines += [f"v {a[0]:.6f} {a[1]:.6f} {a[2]:.6f}", f"v {b[0]:.6f} {b[1]:.6f} {b[2]:.6f}", f"v {c[0]:.6f} {c[1]:.6f} {c[2]:.6f}"]
    lines += [f"f {3*i+1} {3*i+2} {3*i+3}" for i in range(n)]
    g, _ = rt.obj_to_group("\n".join(lines) + "\n")
    g.set_material(Material(color=(0.8, 0.5, 0.3), reflective=0.2))
    w = World(); w.lights.append(Light.point_light((-6., 9., -8.), rt.WHITE)); w.objects.append(g)
    cam = Camera(960, 540, np.pi / 3); cam.transform = view_transform((0., 3., -9.), (0., 1., 0.), (0., 1., 0.))
    return w, cam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
_w, _c = scenes.book_world(32, 32); _c.render(_w)  # HIP start-up and kernel loading happen here, not in the first timed create
for name, make in (("teapot scene", lambda: scenes.teapot_scene(1920, 1080)), (f"{n} random triangles", lambda: soup(n))):
    for flag in ("0", "1"):
        os.environ["RT_AMD_DEVICE_BVH"] = flag
        w, cam = make()
        w.flatten()
        t0 = time.perf_counter(); sc = w.scene(); t1 = time.perf_counter()
        cam.render(w, want_stats=False); cam.render(w, want_stats=False)
        t2 = time.perf_counter()
        for _ in range(5): cam.render(w, want_stats=False)
        t3 = time.perf_counter()
        print(f"{name}: RT_AMD_DEVICE_BVH={flag}  rt_scene_create {1e3*(t1-t0):8.1f} ms  meshes built on GPU {rt.lib.rt_scene_meshes_built_on_gpu(sc)}  render (host buffers) {(t3-t2)/5*1e3:7.2f} ms", flush=True)
