"""ms per frame of the headline scene when the camera changes every frame (an animation), against a static camera:
does the frame graph (captured per camera) get in the way?   RT_AMD_GRAPH=0 for plain enqueue."""
import ctypes as C, os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rt_amd as rt, scenes
from rt_amd import view_transform
w, cam = scenes.teapot_scene(1920, 1080)
scene = w.scene(0)
frame = torch.zeros((1080, 1920, 3), dtype=torch.float64, device="cuda:0")
opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=1, want_counters=0, stream=None)
base = cam.transform
def run(n, moving):
    t0 = time.perf_counter()
    for k in range(n):
        if moving:
            a = 0.002 * k
            cam.transform = view_transform((8. * math.sin(a), 3.5, -9. * math.cos(a)), (0., 1.2, 0.), (0., 1., 0.))
        cd = cam.desc()
        assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(5, False); run(5, True)
print(f"graph={os.environ.get('RT_AMD_GRAPH', 'default')}: static camera {run(50, False):.3f} ms/frame, moving camera {run(50, True):.3f} ms/frame", flush=True)
