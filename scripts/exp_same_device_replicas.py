"""Experiment (GPU): one frame cut into tile shards rendered by R replicas of the scene on the SAME device (rt_render_multi with
devices = [0] * R: a stream, level queues and a host thread per replica), against the plain single-stream frame."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rt_amd as rt, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "teapot"
if which == "teapot": (W, H), depth, (w, cam) = (1920, 1080), 5, scenes.teapot_scene(1920, 1080)
elif which == "mixed": (W, H), depth, (w, cam) = (2560, 1440), 5, scenes.mixed_scene(2560, 1440)
else: (W, H), depth, (w, cam) = (7680, 4320), 8, scenes.instanced_grid(7680, 4320, grid=int(which[4:]))
cd = cam.desc()
opts = rt.RenderOpts(max_depth=depth, rank=0, n_ranks=1, want_counters=0, stream=None)
K = 20 if which in ("teapot", "mixed") else 5
scene = w.scene(0)
frame = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda:0")
for _ in range(4): rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None)
torch.cuda.synchronize(); print(f"{which}: rt_render_frame, frame left in HBM: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/frame")
rgb = rt.pinned_array((H, W, 3), np.float64)
for _ in range(3): rt.lib.rt_render(scene, C.byref(cd), C.byref(opts), rgb.ctypes.data_as(rt.c_dp), None, None, None)
t0 = time.perf_counter()
for _ in range(K): rt.lib.rt_render(scene, C.byref(cd), C.byref(opts), rgb.ctypes.data_as(rt.c_dp), None, None, None)
print(f"{which}: rt_render into a host buffer: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/frame")
for R in (2, 3, 4):
    ms = rt.MultiScene(w, [0] * R)
    for _ in range(4): rt.lib.rt_render_multi(ms.h, C.byref(cd), C.byref(opts), rgb.ctypes.data_as(rt.c_dp), None, None, None, None)
    t0 = time.perf_counter()
    for _ in range(K): rt.lib.rt_render_multi(ms.h, C.byref(cd), C.byref(opts), rgb.ctypes.data_as(rt.c_dp), None, None, None, None)
    print(f"{which}: rt_render_multi, {R} replicas on device 0, host buffer: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/frame (gather kind {ms.gather_kind})")
    ms.close()
