"""Measurement (GPU): a sequence of frames, one at a time (rt_render_frame) against two in flight (rt_render_frame_begin /
rt_render_end: rendered side by side on two streams), device frames and page-locked host frames.
usage: frames_in_flight.py [teapot|mixed|gridN] [frames]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rt_amd as rt, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "teapot"
if which == "teapot": (W, H), depth, (w, cam) = (1920, 1080), 5, scenes.teapot_scene(1920, 1080)
elif which == "mixed": (W, H), depth, (w, cam) = (2560, 1440), 5, scenes.mixed_scene(2560, 1440)
else: (W, H), depth, (w, cam) = (7680, 4320), 8, scenes.instanced_grid(7680, 4320, grid=int(which[4:]))
K = int(sys.argv[2]) if len(sys.argv) > 2 else (40 if which in ("teapot", "mixed") else 8)
cd = cam.desc()
opts = rt.RenderOpts(max_depth=depth, rank=0, n_ranks=1, want_counters=0, stream=None)
scene = w.scene(0)
frames = [torch.zeros((H, W, 3), dtype=torch.float64, device="cuda:0") for _ in range(2)]
def serial(k):
    for j in range(k): assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frames[j & 1].data_ptr(), None, None, None) == 0
def piped(k, begin, bufs, depth=2):
    for j in range(k):
        if j >= depth: assert rt.lib.rt_render_end(scene) == 0, rt.last_error()
        assert begin(scene, C.byref(cd), C.byref(opts), bufs[j % len(bufs)], None, None) == 0, rt.last_error()
    for _ in range(depth): assert rt.lib.rt_render_end(scene) == 0
dev = [f.data_ptr() for f in frames]
serial(4); piped(6, rt.lib.rt_render_frame_begin, dev)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); serial(K); torch.cuda.synchronize(); a = (time.perf_counter() - t0) / K * 1e3
    t0 = time.perf_counter(); piped(K, rt.lib.rt_render_frame_begin, dev); torch.cuda.synchronize(); b = (time.perf_counter() - t0) / K * 1e3
    print(f"{which}: device frames: one at a time {a:.3f} ms/frame, two in flight {b:.3f} ms/frame")
ref = frames[0].cpu().numpy().copy()
serial(2); torch.cuda.synchronize()
assert np.array_equal(frames[0].cpu().numpy(), ref) and np.array_equal(frames[1].cpu().numpy(), ref)
host = [rt.pinned_array((H, W, 3)) for _ in range(4)]
hp = [h.ctypes.data_as(rt.c_dp) for h in host]
for depth in (2, 3, 4):
    piped(8, rt.lib.rt_render_begin, hp, depth)
    t0 = time.perf_counter(); piped(K, rt.lib.rt_render_begin, hp, depth); c = (time.perf_counter() - t0) / K * 1e3
    assert all(np.array_equal(h, ref) for h in host)
    print(f"{which}: page-locked host frames, {depth} in flight: {c:.3f} ms/frame")
