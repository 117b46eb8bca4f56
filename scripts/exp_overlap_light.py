"""Can light streaming kernels (no LDS, few registers) run in the tails of the traversal launches?  A frame rendered on
the library's stream while torch runs f64 pow kernels over 2 M elements on its own stream: both together vs each alone."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rt_amd as rt, scenes
w, cam = scenes.teapot_scene(1920, 1080)
scene = w.scene(0); cd = cam.desc()
frame = torch.zeros((1080, 1920, 3), dtype=torch.float64, device="cuda:0")
opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=1, want_counters=0, stream=None)
x = torch.rand(2_000_000, dtype=torch.float64, device="cuda:0") + 0.5
def frames(k):
    for _ in range(k):
        rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None)
def light(k):
    y = x
    for _ in range(k):
        y = torch.pow(x, 1.7) * 0.5 + y * 0.5
    torch.cuda.synchronize()
    return y
frames(5); light(50)
K, M = 20, 20 * 12     # 12 light kernels-pairs per frame
t0 = time.perf_counter(); frames(K); t_f = time.perf_counter() - t0
t0 = time.perf_counter(); light(M); t_l = time.perf_counter() - t0
th = threading.Thread(target=frames, args=(K,))
t0 = time.perf_counter(); th.start(); light(M); th.join(); t_b = time.perf_counter() - t0
print(f"frames alone {t_f*1e3:.2f} ms, light kernels alone {t_l*1e3:.2f} ms, both together {t_b*1e3:.2f} ms (sum {(t_f+t_l)*1e3:.2f}, max {max(t_f,t_l)*1e3:.2f})")
