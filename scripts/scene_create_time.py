"""Where scene creation goes for C5 at grid g (default 13 = 1.07 M triangles): OBJ parse + group build (host mirror),
flatten (host), rt_scene_create (validation, de-duplication, program + BVH build, upload), first frame.
usage: python scripts/scene_create_time.py [grid]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rt_amd as rt, scenes
g = int(sys.argv[1]) if len(sys.argv) > 1 else 13
_w, _c = scenes.book_world(32, 32); _c.render(_w)  # HIP start-up outside the timings
t0 = time.perf_counter(); w, cam = scenes.instanced_grid(1920, 1080, grid=g); t1 = time.perf_counter()
d = w.flatten(); t2 = time.perf_counter()
os.environ["RT_AMD_VERBOSE"] = "1"
sc = w.scene(0); t3 = time.perf_counter()
cam.render(w, want_stats=False, depth=8) if "depth" in cam.render.__code__.co_varnames else cam.render(w, want_stats=False); t4 = time.perf_counter()
print(f"grid {g}: {d.contents.n_triangles} triangles, {d.contents.n_nodes} nodes | parse + build groups {t1-t0:.2f} s | flatten {t2-t1:.2f} s | rt_scene_create {t3-t2:.2f} s | first 1080p frame {t4-t3:.2f} s")
