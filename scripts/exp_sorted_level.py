"""What would sorting a level's rays by "meets the mesh's world box" buy?  750k incoherent rays (origins in the scene's
volume, random directions) through rt_color_at with remaining = 0 (one k_trace + shadow launch, no children), in
random order and sorted by class; ms_trace of rt_stats.  usage: python scripts/exp_sorted_level.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rt_amd as rt, scenes
w, cam = scenes.teapot_scene(64, 36)
cam.render(w)
# world box of the teapot: its vertices through rotate(PI/7, Y)
v = np.array([[float(x) for x in l.split()[1:4]] for l in scenes.read_asset("teapot.obj").splitlines() if l.startswith("v ")])
a = math.pi / 7
R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
vw = v @ R.T
lo, hi = vw.min(0) - 1e-6, vw.max(0) + 1e-6
rng = np.random.default_rng(3)
n = 750000
o = rng.uniform([-8, 0.01, -8], [8, 4, 8], size=(n, 3))
d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
with np.errstate(divide="ignore", invalid="ignore"):
    t0, t1 = (lo - o) / d, (hi - o) / d
tmin, tmax = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
meets = (tmax >= np.maximum(tmin, 0.))
print(f"{meets.mean()*100:.1f} % of the rays meet the mesh's world box")
order = np.argsort(~meets, kind="stable")
for name, oo, dd in (("random order", o, d), ("sorted by class", o[order], d[order])):
    best = 1e9
    for _ in range(6):
        rgb, ids, fl, st = w.color_at_batch(oo, dd, remaining=0, want_stats=True)
        best = min(best, st.ms_trace)
    print(f"{name:16s}: k_trace (closest hit launch + shadow launch) {best*1e3:8.1f} us, launches {st.n_trace_launches}")
