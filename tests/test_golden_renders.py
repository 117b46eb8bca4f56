"""tests/golden/render_*.npz (scripts/make_golden.py): renders of the BASELINE configs at reduced resolution, frozen
as f64 bit patterns.  They are regression fixtures of this repository's oracle — the reference (Rust) cannot run in the
build image; what ties the oracle to the reference is test_oracle_kats.py.  CPU: the oracle still reproduces them bit
for bit.  GPU: the HIP path matches them to the parity bar (hit ids exact, colours < 1e-9, ray counts equal)."""
import glob
import os

import numpy as np
import pytest

import oracle_bridge as ob
import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BUILDERS = {
    "c1_book_world": scenes.book_world,
    "c2_teapot_low": scenes.teapot_low,
    "c3_teapot_glass_mirror": scenes.teapot_scene,
    "c4_mixed_csg_nested_glass": scenes.mixed_scene,
    "c5_instanced_grid_depth8": lambda w, h: scenes.instanced_grid(w, h, grid=2),
}
NAMES = sorted(BUILDERS)


def _same_ids(got, want):
    """Object ids come from a process-wide counter (src/lib.rs:46), so a scene built after other scenes carries the
    fixture's ids plus a constant: hits are compared after removing that constant (0 = miss stays 0)."""
    if not (got > 0).any() or not (want > 0).any():
        return np.array_equal(got, want)
    shift = int(got[got > 0].min()) - int(want[want > 0].min())
    return np.array_equal(np.where(got > 0, got - shift, 0), want)


def _load(name):
    z = np.load(os.path.join(GOLDEN, f"render_{name}.npz"), allow_pickle=False)
    w, h, depth = (int(x) for x in z["size"])
    return z, w, h, depth


def test_every_fixture_has_a_builder():
    found = sorted(os.path.basename(p)[len("render_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "render_*.npz")))
    assert found == NAMES


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_golden_render(name):
    z, w, h, depth = _load(name)
    world, cam = BUILDERS[name](w, h)
    img = ob.render(world, cam, depth)
    assert np.array_equal(img.rgb, z["rgb"], equal_nan=True)
    assert _same_ids(img.hit_id, z["hit_id"]) and np.array_equal(img.flags, z["flags"])
    st = img.stats
    assert [st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow, st.shaded_nodes] == z["counts"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_matches_the_golden_render(name):
    z, w, h, depth = _load(name)
    world, cam = BUILDERS[name](w, h)
    for want_stats in (True, False):
        img = cam.render(world, depth, want_stats=want_stats)
        assert _same_ids(img.hit_id, z["hit_id"]) and np.array_equal(img.flags, z["flags"])
        err = np.abs(img.rgb - z["rgb"])
        err[np.isnan(img.rgb) & np.isnan(z["rgb"])] = 0.
        assert float(err.max()) < 1e-9
        if want_stats:
            st = img.stats
            assert [st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow, st.shaded_nodes] == z["counts"].tolist()
