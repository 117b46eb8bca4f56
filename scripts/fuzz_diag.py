"""Diagnostics for tests/test_gpu_fuzz.py: per seed, which flag bits differ between the device and the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_bridge as ob
import scenes

seeds = [int(x) for x in sys.argv[1].split(",")]
W, H = 64, 40
for seed in seeds:
    mesh = len(sys.argv) > 2 and sys.argv[2] == "mesh"
    wild = len(sys.argv) > 2 and sys.argv[2] == "wild"
    depth = (3 if seed % 2 else 5) if mesh else 4 if wild else (4 if seed % 3 else 6)
    world, cam = scenes.random_scene(seed, W, H, wild=True) if wild else (scenes.random_mesh_scene if mesh else scenes.random_scene)(seed, W, H)
    g = cam.render(world, depth)
    o = ob.render(world, cam, depth)
    line = [f"seed {seed} depth {depth} objects {len(world.objects)}"]
    for b in range(3):
        gb, obit = (g.flags >> b) & 1, (o.flags >> b) & 1
        line.append(f"bit{b}: gpu-only {int((gb & ~obit & 1).sum())} oracle-only {int((obit & ~gb & 1).sum())} both {int((gb & obit).sum())}")
    line.append(f"ids differ {int((g.hit_id != o.hit_id).sum())}")
    err = np.abs(g.rgb - o.rgb)
    line.append(f"one-sided NaN {int((np.isnan(g.rgb) != np.isnan(o.rgb)).sum())}")
    # the same frame at depth 0 (primary rays only): where does the difference start?
    g0 = cam.render(world, 0)
    o0 = ob.render(world, cam, 0)
    line.append(f"depth0 flag diffs {int((g0.flags != o0.flags).sum())}")
    print("; ".join(line), flush=True)
    with np.errstate(invalid="ignore"):
        bad = (np.isnan(g.rgb) != np.isnan(o.rgb)).any(axis=-1) | (g.flags != o.flags) | (np.nan_to_num(np.abs(g.rgb - o.rgb)) > 1e-9).any(axis=-1)
    one = np.argwhere(bad)
    print(f"   stats gpu {g.stats.rays_reflect} {g.stats.rays_refract} {g.stats.rays_shadow} oracle {o.stats.rays_reflect} {o.stats.rays_refract} {o.stats.rays_shadow}; smooth triangles {world.flatten().contents.n_smooth}")
    for y, x in one[:4]:
        print(f"   pixel ({x},{y}): gpu {g.rgb[y, x]} oracle {o.rgb[y, x]} id {g.hit_id[y, x]}/{o.hit_id[y, x]} flags {g.flags[y, x]}/{o.flags[y, x]}")
        ro, rd = cam.rays_for_pixels(np.array([x], dtype=np.int32), np.array([y], dtype=np.int32))
        for rem in range(depth + 1):
            gc = world.color_at_batch(ro, rd, rem)
            oc = ob.color_at_batch(world, ro, rd, rem)
            print(f"      remaining {rem}: gpu {gc[0][0]} flags {gc[2][0]}  oracle {oc[0][0]} flags {oc[2][0]}")
