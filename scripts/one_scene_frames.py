"""Renders a few frames of one bench scene without rt_stats (for kernel traces).  usage: one_scene_frames.py grid13|teapot|mixed [frames]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rt_amd as rt, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "grid13"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if which.startswith("grid"):
    (W, H), depth = (7680, 4320), 8
    w, cam = scenes.instanced_grid(W, H, grid=int(which[4:]))
elif which == "mixed":
    (W, H), depth = (2560, 1440), 5
    w, cam = scenes.mixed_scene(W, H)
else:
    (W, H), depth = (1920, 1080), 5
    w, cam = scenes.teapot_scene(W, H)
scene = w.scene(0)
cd = cam.desc()
frame = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda:0")
opts = rt.RenderOpts(max_depth=depth, rank=0, n_ranks=1, want_counters=0, stream=None)
for _ in range(frames):
    assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None) == 0
torch.cuda.synchronize()
