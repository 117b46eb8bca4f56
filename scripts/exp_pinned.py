import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo/ray-tracer_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import rt_amd as rt, scenes
w, cam = scenes.teapot_scene(1920, 1080)
scene = w.scene(0); cd = cam.desc()
opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=1, want_counters=0, stream=None)
for name, buf in (("pageable", np.zeros((1080, 1920, 3))), ("pinned", torch.zeros((1080, 1920, 3), dtype=torch.float64).pin_memory().numpy())):
    p = buf.ctypes.data_as(rt.c_dp)
    for _ in range(3): rt.lib.rt_render(scene, C.byref(cd), C.byref(opts), p, None, None, None)
    t0 = time.perf_counter()
    for _ in range(10): rt.lib.rt_render(scene, C.byref(cd), C.byref(opts), p, None, None, None)
    print(name, (time.perf_counter() - t0) / 10 * 1e3, "ms per rt_render", float(buf.sum()))
