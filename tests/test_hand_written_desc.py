"""Scene tables written by hand (tests/hand_scene.py) against the flattener, the oracle and the device.

Every other GPU-vs-oracle test hands BOTH sides the output of ray-tracer_amd/host/flatten.cpp, so a flattener or
host-mirror bug could hide behind the shared input.  Here (1) the flattener's output must expand to exactly the
per-node content of the hand-written tables, (2) the oracle fed the hand-written tables reproduces the reference's
own known answers (camera.rs:164-178, world.rs:186-203) and renders the same image from either set of tables, and
(3) the device, created straight from the hand-written tables through rt_scene_create, matches the oracle."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import hand_scene as hs
import oracle_bridge as ob
import rt_amd as rt
from rt_amd import Camera, view_transform

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")) as _f:
    KAT = json.load(_f)
PI = math.pi
CASES = [(hs.default_world, hs.build_default_world), (hs.composite_world, hs.build_composite_world), (hs.mesh_world, hs.build_mesh_world)]


def oracle_render(desc_ptr, cam, depth=5):
    cd = cam.desc()
    rgb = np.zeros((cam.vsize, cam.hsize, 3))
    ids = np.zeros((cam.vsize, cam.hsize), dtype=np.int32)
    flags = np.zeros((cam.vsize, cam.hsize), dtype=np.uint8)
    st = ob.OrcStats()
    assert ob.olib.orc_render_rows(desc_ptr, C.byref(cd), depth, 1, 0, cam.vsize - 1, rgb.ctypes.data_as(rt.c_dp),
                                   ids.ctypes.data_as(rt.c_ip), flags.ctypes.data_as(rt.c_bp), C.byref(st)) >= 0
    return rgb, ids, flags, st


def camera_for(which):
    if which == 0:
        k = KAT["camera_render_default_world"]
        cam = Camera(k["hsize"], k["vsize"], k["fov"])
        cam.transform = view_transform(k["from"], k["to"], k["up"])
    elif which == 1:
        cam = Camera(72, 48, PI / 3)
        cam.transform = view_transform((0.5, 2.5, -7.), (-0.5, 0.3, 0.), (0., 1., 0.))
    else:
        cam = Camera(80, 60, PI / 3)
        cam.transform = view_transform((1.5, 2.5, -7.), (0.5, 0.3, 1.), (0., 1., 0.))
    return cam


@pytest.mark.parametrize("which", [0, 1, 2])
def test_flattener_output_has_the_hand_written_content(which):
    hand, build = CASES[which]
    h = hand()
    w = build()
    a, b = hs.canonical(h.pointer()), hs.canonical(w.flatten())
    assert len(a["nodes"]) == len(b["nodes"])
    for n, (x, y) in enumerate(zip(a["nodes"], b["nodes"])):
        assert x.keys() == y.keys(), n
        for key in x:
            assert x[key] == y[key], (n, key, x[key], y[key])
    assert a["lights"] == b["lights"]


def test_oracle_on_hand_written_default_world_reproduces_reference_kats():
    h = hs.default_world()
    k = KAT["camera_render_default_world"]
    rgb, ids, _, _ = oracle_render(h.pointer(), camera_for(0))
    x, y = k["pixel"]
    assert np.all(np.abs(rgb[y, x] - np.array(k["color"])) < 1e-5)          # camera.rs:164-178
    assert ids[y, x] == 11 and not rgb[-1].any() and not rgb[:, -1].any()    # F1
    o = np.array([KAT["world_intersect"]["ray"][0]], dtype=np.float64)
    d = np.array([KAT["world_intersect"]["ray"][1]], dtype=np.float64)
    cnt = np.zeros(1, dtype=np.int32)
    t = np.zeros((1, 8))
    oid = np.zeros((1, 8), dtype=np.int32)
    assert ob.olib.orc_intersect_world(h.pointer(), 1, o.ctypes.data_as(rt.c_dp), d.ctypes.data_as(rt.c_dp), 8,
                                       cnt.ctypes.data_as(rt.c_ip), t.ctypes.data_as(rt.c_dp), oid.ctypes.data_as(rt.c_ip)) >= 0
    assert cnt[0] == 4 and list(t[0, :4]) == KAT["world_intersect"]["t"] and list(oid[0, :4]) == [11, 12, 12, 11]   # world.rs:186-203


@pytest.mark.parametrize("which", [0, 1, 2])
def test_oracle_renders_the_same_image_from_either_set_of_tables(which):
    hand, build = CASES[which]
    h = hand()
    w = build()
    cam = camera_for(which)
    a = oracle_render(h.pointer(), cam)
    b = oracle_render(w.flatten(), cam)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[1] != 0, b[1] != 0)           # ids differ in value (hand-chosen), not in where something is hit
    assert (a[3].rays_primary, a[3].rays_reflect, a[3].rays_refract, a[3].rays_shadow) == \
        (b[3].rays_primary, b[3].rays_reflect, b[3].rays_refract, b[3].rays_shadow)
    if which == 1:
        assert a[3].rays_reflect > 0 and a[3].rays_refract > 0 and len(np.unique(a[1])) >= 6
    if which == 2:   # the mesh is seen (flat and smooth faces), refracts and reflects
        assert a[3].rays_reflect > 0 and a[3].rays_refract > 0 and {207, 211} & set(np.unique(a[1]).tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1, 2])
def test_device_from_hand_written_tables_matches_the_oracle(which):
    hand, build = CASES[which]
    h = hand()
    cam = camera_for(which)
    scene = C.c_void_p()
    rc = rt.lib.rt_scene_create(h.pointer(), 0, C.byref(scene))
    assert rc == 0, rt.last_error()
    try:
        cd = cam.desc()
        opts = rt.RenderOpts(max_depth=5, rank=0, n_ranks=1, want_counters=0, stream=None)
        rgb = np.zeros((cam.vsize, cam.hsize, 3))
        ids = np.zeros((cam.vsize, cam.hsize), dtype=np.int32)
        flags = np.zeros((cam.vsize, cam.hsize), dtype=np.uint8)
        st = rt.Stats()
        rc = rt.lib.rt_render(scene, C.byref(cd), C.byref(opts), rgb.ctypes.data_as(rt.c_dp), ids.ctypes.data_as(rt.c_ip),
                              flags.ctypes.data_as(rt.c_bp), C.byref(st))
        assert rc == 0, rt.last_error()
    finally:
        rt.lib.rt_scene_destroy(scene)
    o_rgb, o_ids, o_flags, o_st = oracle_render(h.pointer(), cam)
    assert np.array_equal(ids, o_ids) and np.array_equal(flags, o_flags)
    err = np.abs(rgb - o_rgb)
    err[np.isnan(rgb) & np.isnan(o_rgb)] = 0.
    assert float(err.max()) < 1e-9
    assert (st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow) == \
        (o_st.rays_primary, o_st.rays_reflect, o_st.rays_refract, o_st.rays_shadow)
    # and the same pixels as the device gets from the flattener's tables for the same scene
    w = build()
    g = cam.render(w)
    assert np.array_equal(g.rgb, rgb) and np.array_equal(g.hit_id != 0, ids != 0)
