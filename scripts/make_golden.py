"""Regenerates tests/golden/render_*.npz: oracle renders of the BASELINE configs at reduced resolution.

These are REGRESSION fixtures of this repository's own oracle (oracle/rt_oracle.c), not outputs of the Rust reference
(which cannot run here: no rustc/cargo).  What pins the oracle to the reference is tests/test_oracle_kats.py (the
reference's own unit-test vectors); these files pin the oracle and the device path against drift between rounds:
rgb as f64 (bit patterns), primary hit ids, flags and the ray counts by class.

usage: python scripts/make_golden.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_bridge as ob
import scenes

CASES = {
    "c1_book_world": (scenes.book_world, (160, 80), 5),
    "c2_teapot_low": (scenes.teapot_low, (64, 36), 5),
    "c3_teapot_glass_mirror": (scenes.teapot_scene, (96, 54), 5),
    "c4_mixed_csg_nested_glass": (scenes.mixed_scene, (128, 72), 5),
    "c5_instanced_grid_depth8": (lambda w, h: scenes.instanced_grid(w, h, grid=2), (64, 36), 8),
}


def main():
    for name, (builder, (w, h), depth) in CASES.items():
        world, cam = builder(w, h)
        img = ob.render(world, cam, depth)
        st = img.stats
        counts = np.array([st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow, st.shaded_nodes], dtype=np.int64)
        path = os.path.join(ROOT, "tests", "golden", f"render_{name}.npz")
        np.savez_compressed(path, rgb=img.rgb, hit_id=img.hit_id, flags=img.flags, counts=counts,
                            size=np.array([w, h, depth], dtype=np.int32))
        print(name, img.rgb.shape, counts.tolist(), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
