"""Instrumented work counters per ray (want_counters = 1) for a BASELINE config: what a ray costs on average.
usage: python scripts/work_per_ray.py [c3|c4|c5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rt_amd as rt, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
w, cam, depth = {"c3": lambda: (*scenes.teapot_scene(1920, 1080), 5), "c4": lambda: (*scenes.mixed_scene(2560, 1440), 5),
                 "c5": lambda: (*scenes.instanced_grid(3840, 2160, grid=4), 8)}[which]()
img = cam.render(w, depth, want_counters=True)
st = img.stats
names = ["xforms", "box tests", "gates", "tri tests", "tri det exits", "tri u exits", "tri v exits", "spheres", "planes", "cubes", "cylinders", "cones", "bvh nodes"]
traced = st.rays_primary + st.rays_reflect + st.rays_refract
print(f"{which}: traced {traced}, shadow {st.rays_shadow}, walks {st.rays_container}")
for label, work, n in (("closest-hit", st.work_trace, traced), ("shadow", st.work_shadow, st.rays_shadow), ("n1/n2 walk", st.work_container, st.rays_container)):
    if n:
        print(f"  {label:12s} per ray: " + ", ".join(f"{names[k]} {work[k] / n:.2f}" for k in range(13) if work[k]))
