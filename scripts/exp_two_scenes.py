"""Experiment (GPU): does the chip have spare capacity while one frame renders?  Renders K frames on one scene, then K
frames on each of T scenes from T host threads at once (separate rt_scene handles: own streams, queues, graphs)."""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rt_amd as rt, scenes

T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
WHAT = sys.argv[2] if len(sys.argv) > 2 else "teapot"   # teapot | mixed | grid4 | grid13
NR = int(sys.argv[3]) if len(sys.argv) > 3 else 1         # render rank 0's shard of an NR-way split instead of the whole frame
K = 30 if WHAT in ("teapot", "mixed") else 8
make = {"teapot": lambda: scenes.teapot_scene(1920, 1080), "mixed": lambda: scenes.mixed_scene(2560, 1440),
        "grid4": lambda: scenes.instanced_grid(7680, 4320, grid=4), "grid13": lambda: scenes.instanced_grid(7680, 4320, grid=13)}[WHAT]
worlds = [make() for _ in range(T)]
frames = [torch.zeros((worlds[0][1].vsize, worlds[0][1].hsize, 3), dtype=torch.float64, device="cuda:0") for _ in range(T)]
hs = [w.scene(0) for w, _ in worlds]
cds = [cam.desc() for _, cam in worlds]
opts = rt.RenderOpts(max_depth=8 if WHAT.startswith("grid") else 5, rank=0, n_ranks=NR, want_counters=0, stream=None)

def run(i, k):
    for _ in range(k):
        assert rt.lib.rt_render_frame(hs[i], C.byref(cds[i]), C.byref(opts), frames[i].data_ptr(), None, None, None) == 0

for i in range(T):
    run(i, 4)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(0, K); torch.cuda.synchronize(); one = (time.perf_counter() - t0) / K * 1e3
th = [threading.Thread(target=run, args=(i, K)) for i in range(T)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
many = (time.perf_counter() - t0) / K * 1e3
print(f"{WHAT} (rank 0 of {NR}): 1 scene: {one:.3f} ms/frame; {T} scenes at once: {many:.3f} ms per round of {T} frames = {many / T:.3f} ms/frame")
