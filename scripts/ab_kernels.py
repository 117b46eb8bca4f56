"""Per-kernel-class times of the headline frame (rt_stats, HIP events; mean of 8 frames) for the library RT_AMD_LIB names
(default: the in-tree build).  For same-box A/B runs of build variants of one kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rt_amd as rt, scenes
for name, build, size, depth in (("C3", scenes.teapot_scene, (1920, 1080), 5), ("C4", scenes.mixed_scene, (2560, 1440), 5)):
    w, cam = build(*size)
    for _ in range(3): cam.render(w, depth)
    acc = np.zeros(5)
    for _ in range(8):
        s = cam.render(w, depth).stats
        acc += [s.ms_trace, s.ms_walk, s.ms_queue, s.ms_kshade, s.ms_total]
    acc /= 8
    print(f"{os.environ.get('RT_AMD_LIB', 'in-tree')[-16:]:16s} {name}: k_trace {acc[0]*1e3:7.1f}  k_walk {acc[1]*1e3:6.1f}  k_queue(+k_spawn) {acc[2]*1e3:6.1f}  k_shade {acc[3]*1e3:6.1f}  frame {acc[4]*1e3:7.1f} us", flush=True)
