import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo/ray-tracer_amd"); sys.path.insert(0, "/root/repo/tests")
import torch
import rt_amd as rt, scenes
w, cam = scenes.teapot_scene(1920, 1080)
scene = w.scene(0); cd = cam.desc()
frame = torch.zeros((1080, 1920, 3), dtype=torch.float64, device="cuda:0")
opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=1, want_counters=0, stream=None)
for _ in range(8):
    assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None) == 0
torch.cuda.synchronize()
