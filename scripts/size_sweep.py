"""Diagnostic (GPU): level-0 kernel times of the teapot frame at several resolutions (rt_stats, depth 0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd"))
import numpy as np
import rt_amd as rt, scenes
for W, H in ((64, 36), (240, 135), (480, 270), (960, 540), (1920, 1080), (2880, 1620), (3840, 2160)):
    w, cam = scenes.teapot_scene(W, H)
    cam.render(w, 0)
    acc = np.zeros(4)
    for _ in range(5):
        s = cam.render(w, 0).stats
        acc += [s.ms_trace, s.ms_shadow, s.ms_queue, s.ms_total]
    acc /= 5
    n = (W - 1) * (H - 1)
    print(f"{W}x{H}: {n:8d} rays  trace {acc[0]*1e3:7.1f} us  shadow {acc[1]*1e3:7.1f} us  queue+finish {acc[2]*1e3:6.1f} us  total {acc[3]*1e3:7.1f} us")
