"""Wall time of building a scene (parse + host tree, flatten, rt_scene_create) and of its first and second frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rt_amd as rt, scenes
scenes.book_world(64, 32)[1].render(scenes.book_world(64, 32)[0])   # HIP initialised
for name, b, size, depth in (("C1 book world", scenes.book_world, (320, 160), 5), ("C3 teapot", scenes.teapot_scene, (1920, 1080), 5),
                             ("C4 mixed", scenes.mixed_scene, (2560, 1440), 5), ("C5 grid 4", lambda w, h: scenes.instanced_grid(w, h, grid=4), (7680, 4320), 8),
                             ("C5 grid 13", lambda w, h: scenes.instanced_grid(w, h, grid=13), (7680, 4320), 8)):
    t0 = time.perf_counter(); w, cam = b(*size); t1 = time.perf_counter()
    w.flatten(); t2 = time.perf_counter()
    w.scene(0); t3 = time.perf_counter()
    cam.render(w, depth, want_stats=False); t4 = time.perf_counter()
    cam.render(w, depth, want_stats=False); t5 = time.perf_counter()
    print(f"{name:12s} build {t1 - t0:6.3f} s  flatten {t2 - t1:6.3f} s  rt_scene_create {t3 - t2:6.3f} s  first rt_render {t4 - t3:6.3f} s  second {t5 - t4:6.3f} s", flush=True)
    w.invalidate()
