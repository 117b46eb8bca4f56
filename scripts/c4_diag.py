import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rt_amd as rt, scenes
w, cam = scenes.mixed_scene(2560, 1440)
cam.render(w, 5)
st = cam.render(w, 5, want_counters=True).stats
names = ["xforms", "box tests", "gates", "tri tests", "tri det exits", "tri u exits", "tri v exits", "spheres", "planes", "cubes", "cylinders", "cones", "bvh nodes"]
traced = st.rays_primary + st.rays_reflect + st.rays_refract
print(os.environ.get("RT_AMD_LIB", "current"), f"ms_trace {st.ms_trace:.2f} walk {st.ms_walk:.2f} total {st.ms_total:.2f}")
for label, work, k in (("closest", st.work_trace, traced), ("shadow", st.work_shadow, st.rays_shadow)):
    print("  ", label, ", ".join(f"{names[j]} {work[j]/k:.2f}" for j in range(13) if work[j]))
