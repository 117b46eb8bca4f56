"""Times rt_render_shard for rank 0 of N = 1, 2, 4, 8 on ONE GPU (what each rank of an N-GPU run would spend rendering,
gather excluded): the strong-scaling limit of the render itself.  Then the same shards two in flight
(rt_render_shard_begin / rt_render_end), which is how bench.py's N > 1 path renders them."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import rt_amd as rt
import scenes

w, cam = scenes.teapot_scene(1920, 1080)
scene = w.scene(0)
cd = cam.desc()
for n in (1, 2, 4, 8):
    cnt = int(rt.lib.rt_shard_pixel_count(C.byref(cd), 0, n))
    buf = torch.zeros((cnt, 3), dtype=torch.float64, device="cuda:0")
    opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=n, want_counters=0, stream=None)
    for _ in range(3):
        assert rt.lib.rt_render_shard(scene, C.byref(cd), C.byref(opts), buf.data_ptr(), None, None, None) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        rt.lib.rt_render_shard(scene, C.byref(cd), C.byref(opts), buf.data_ptr(), None, None, None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    bufs = [torch.zeros((cnt, 3), dtype=torch.float64, device="cuda:0") for _ in range(3)]

    def in_flight(k_frames, depth):
        for k in range(k_frames):
            if k >= depth:
                assert rt.lib.rt_render_end(scene) == 0
            assert rt.lib.rt_render_shard_begin(scene, C.byref(cd), C.byref(opts), bufs[k % depth].data_ptr(), None, None) == 0, rt.last_error()
        for _ in range(min(k_frames, depth)):
            assert rt.lib.rt_render_end(scene) == 0
    res = []
    for depth in (2, 3):
        in_flight(6, depth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K2 = 60
        in_flight(K2, depth)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / K2 * 1e3)
    print(f"rank 0 of {n}: {cnt} slots, {ms:.3f} ms per shard render, {res[0]:.3f} ms with two shards in flight, {res[1]:.3f} with three", flush=True)
