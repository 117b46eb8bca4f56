"""A/B timing of the BASELINE configs for a given package directory (default: ray-tracer_amd): ms per rt_render_frame."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = sys.argv[1] if len(sys.argv) > 1 else "ray-tracer_amd"
sys.path.insert(0, os.path.join(ROOT, pkg)); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rt_amd as rt, scenes
cases = [("C2 teapot_low", scenes.teapot_low, (1920, 1080), 5), ("C3 teapot", scenes.teapot_scene, (1920, 1080), 5),
         ("C3s smooth", lambda w, h: scenes.teapot_scene(w, h, obj_text=scenes.smooth_teapot_obj()), (1920, 1080), 5),
         ("C4 mixed", scenes.mixed_scene, (2560, 1440), 5), ("C5 grid4", lambda w, h: scenes.instanced_grid(w, h, grid=4), (7680, 4320), 8)]
if os.environ.get("AB_GRIDS"):   # only C5 at the listed grid sizes
    cases = [("C5 grid%d" % g, (lambda g: lambda w, h: scenes.instanced_grid(w, h, grid=g))(g), (7680, 4320), 8) for g in map(int, os.environ["AB_GRIDS"].split(","))]
elif os.environ.get("AB_GRID13", "1") != "0":
    cases.append(("C5 grid13", lambda w, h: scenes.instanced_grid(w, h, grid=13), (7680, 4320), 8))
for name, b, (W, H), depth in cases:
    w, cam = b(W, H)
    scene = w.scene(0)
    cd = cam.desc()
    frame = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda:0")
    opts = rt.RenderOpts(max_depth=depth, rank=0, n_ranks=1, want_counters=0, stream=None)
    for _ in range(4):
        assert rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize()
    K = 10 if not name.startswith("C5") else 5
    t0 = time.perf_counter()
    for _ in range(K):
        rt.lib.rt_render_frame(scene, C.byref(cd), C.byref(opts), frame.data_ptr(), None, None, None)
    torch.cuda.synchronize()
    print(f"{pkg:16s} {name:14s} {(time.perf_counter() - t0) / K * 1e3:8.3f} ms/frame", flush=True)
    del frame
    w.invalidate()
