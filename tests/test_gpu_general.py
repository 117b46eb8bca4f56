"""The 4-component path (ray-tracer_amd/csrc/rt_general.hip.h) against the oracle.

The kernels the library is built around drop the w of every tuple, exact while every inverse is affine.  What a user
of the reference can build beyond that — a camera whose transform has no inverse (src/camera.rs:44-46: every pixel's
ray is Ray(ORIGIN, ORIGIN), a point as direction), and Matrix::new(4, values) with a last row other than 0,0,0,1 as a
camera, shape or pattern transform — is rendered by a second path that carries all four components.  The oracle is
4-component throughout (oracle/rt_oracle.c), so it is the checker for both; and with RT_AMD_GENERAL=1 the ordinary
scenes of the suite go through the 4-component path too, which checks it where the answer is known three ways
(oracle, w-free kernels, reference KATs)."""
import math

import numpy as np
import pytest

import oracle_bridge as ob
import rt_amd as rt
import scenes
from rt_amd import Axis, Camera, Light, Material, Matrix, Pattern, Shape, World, rotate, scale, translate, view_transform
from test_gpu_fuzz import compare as fuzz_compare
from test_gpu_parity import COLOR_TOL, _random_rays, assert_image_parity

pytestmark = pytest.mark.gpu
PI = math.pi


def projective(m, row=(0.03, -0.02, 0.015, 1.0)):
    """`m` with its last row replaced: still invertible (the row is close to 0,0,0,1), no longer affine."""
    v = np.array(m.m, dtype=np.float64).copy()
    v[12:16] = row
    out = Matrix(v)
    inv = out.inverse()
    assert inv is not None and tuple(inv.m[12:16]) != (0., 0., 0., 1.)
    return out


def assert_batches_match(w, n=1500, seed=3, center=(0., 1., 0.), radius=9., depth=5):
    o, d = _random_rays(n, seed, center=center, radius=radius)
    gc, gt, gi = w.intersect_world_batch(o, d, 64)
    oc, ot, oi = ob.intersect_world_batch(w, o, d, 64)
    assert np.array_equal(gc, oc) and np.array_equal(gt, ot, equal_nan=True) and np.array_equal(gi, oi)
    g_rgb, g_id, g_fl, g_st = w.color_at_batch(o, d, depth, want_stats=True)
    o_rgb, o_id, o_fl = ob.color_at_batch(w, o, d, depth)
    assert np.array_equal(g_id, o_id) and np.array_equal(g_fl, o_fl)
    err = np.abs(g_rgb - o_rgb)
    err[np.isnan(g_rgb) & np.isnan(o_rgb)] = 0.
    assert float(err.max()) < COLOR_TOL
    assert g_st.rays_primary == n and g_st.shaded_nodes >= int((g_id != 0).sum())
    assert np.array_equal(w.is_shadowed_batch(o), ob.is_shadowed_batch(w, o))
    return gc


# ------------------------------------------------------------------ cameras
def _glass_and_mirror_world():
    """Objects placed so that Ray(ORIGIN, ORIGIN) — local origin = local direction = the translation column of each
    inverse, w = 1 — meets several of them and is reflected and refracted."""
    w = World()
    w.lights.append(Light.point_light((-6., 7., -8.), rt.WHITE))
    w.lights.append(Light.point_light((5., 3., -2.), (0.4, 0.4, 0.5)))
    w.objects.append(Shape.plane().set_transform(translate(0., -1.5, 0.)).set_material(
        Material(reflective=0.4).with_pattern(Pattern.checkered((0.9, 0.9, 0.9), (0.2, 0.2, 0.3)))))
    w.objects.append(Shape.glass_sphere().set_transform(translate(0.3, 0.2, 0.5) * scale(1.5, 1.5, 1.5)))
    w.objects.append(Shape.cube().set_transform(translate(-0.4, 0.5, 0.3) * rotate(0.4, Axis.Y) * scale(2., 2., 2.)).set_material(
        Material(color=(0.8, 0.3, 0.2), reflective=0.3, transparency=0.5, refractive_index=1.2)))
    cyl = Shape.cylinder().set_minimum(-1.).set_maximum(2.).set_closed(True)
    w.objects.append(cyl.set_transform(translate(0.2, 0., -0.3) * scale(3., 1., 3.)).set_material(Material(color=(0.2, 0.6, 0.9), reflective=0.2)))
    g = Shape.group().set_transform(translate(0.1, -0.2, 0.2) * rotate(0.3, Axis.Z))
    g.add_child(Shape.sphere().set_transform(scale(2.5, 2.5, 2.5)).set_material(Material(color=(0.3, 0.8, 0.3), transparency=0.6, refractive_index=1.1)))
    g.add_child(Shape.cone().set_minimum(-2.).set_maximum(2.).set_closed(True).set_transform(translate(0.2, 0.1, 0.)))
    w.objects.append(g)
    return w


@pytest.mark.parametrize("singular", [scale(0., 1., 1.), scale(1., 0., 1.) * translate(3., 2., 1.), Matrix(np.zeros(16)),
                                      Matrix([1., 2., 3., 4., 2., 4., 6., 8., 0., 1., 0., 0., 0., 0., 1., 0.])])
def test_camera_without_an_inverse_renders_the_reference_frame(singular):  # src/camera.rs:44-46
    assert singular.inverse() is None
    for w in (World.default(), _glass_and_mirror_world(), scenes.book_world(8, 8)[0], scenes.mixed_scene(8, 8)[0]):
        cam = Camera(24, 16, 1.0)
        cam.transform = singular
        g, o = assert_image_parity(w, cam, 5)
        # every rendered pixel carries the same ray, so the frame is flat
        inner = g.rgb[:-1, :-1].reshape(-1, 3)
        assert np.array_equal(inner, np.broadcast_to(inner[0], inner.shape), equal_nan=True)
        assert len(np.unique(g.hit_id[:-1, :-1])) == 1
        # shards of such a frame, and the stats-free (graph) path
        parts = [cam.render(w, 5, rank=r, n_ranks=3) for r in range(3)]
        rgb = np.zeros_like(g.rgb)
        for p in parts:
            m = ~np.isnan(p.rgb) & (p.rgb != 0.)
            rgb[m] = p.rgb[m]
        if not np.isnan(g.rgb).any():
            assert np.array_equal(rgb, g.rgb)
        assert sum(p.stats.rays_primary for p in parts) == g.stats.rays_primary
        assert sum(p.stats.rays_shadow for p in parts) == g.stats.rays_shadow
        q = cam.render(w, 5, want_stats=False)
        assert np.array_equal(q.rgb, g.rgb, equal_nan=True) and np.array_equal(q.hit_id, g.hit_id)
    # the reflected and refracted rays of that one ray really exist in the second world
    cam = Camera(9, 9, 1.0)
    cam.transform = scale(0., 1., 1.)
    w = _glass_and_mirror_world()
    w.objects = [w.objects[k] for k in (0, 2, 3)]   # floor, cube, cylinder: the ray meets the cube first
    g, _ = assert_image_parity(w, cam, 5)
    assert g.stats.rays_reflect > 0 and g.stats.rays_refract > 0 and g.hit_id[0, 0] != 0


def test_ray_for_pixel_cannot_hand_back_a_point_as_direction():
    """rt_ray_for_pixel returns 3-component directions; Ray(ORIGIN, ORIGIN) has w = 1 in its direction, which that
    signature cannot carry: an error there, while rt_render renders the frame."""
    cam = Camera(8, 8, 1.0)
    cam.transform = scale(0., 1., 1.)
    with pytest.raises(rt.RtError) as e:
        cam.ray_for_pixel(1, 1)
    assert e.value.code == rt.RT_ERR_UNSUPPORTED


def test_camera_with_a_projective_transform():
    for builder, size in ((scenes.book_world, (48, 32)), (scenes.mixed_scene, (40, 24)), (scenes.cover_like_scene, (32, 32))):
        w, cam = builder(*size)
        cam.transform = projective(cam.transform)
        g, _ = assert_image_parity(w, cam, 5)
        assert len(np.unique(g.hit_id)) > 2
        w2, cam2 = builder(*size)   # the same camera, affine: a different frame (the last row matters)
        assert not np.array_equal(cam2.render(w2, 5).rgb, g.rgb)


# ------------------------------------------------------------------ scenes with non-affine inverses
def _projective_world(seed=0):
    rng = np.random.RandomState(900 + seed)

    def pr(m):
        return projective(m, row=tuple(rng.uniform(-0.04, 0.04, 3)) + (float(rng.uniform(0.9, 1.1)),))

    w = World()
    w.lights.append(Light.point_light((-7., 9., -8.), rt.WHITE))
    w.lights.append(Light.point_light((6., 5., -7.), (0.35, 0.35, 0.4)))
    w.objects.append(Shape.plane().set_material(Material(reflective=0.3).with_pattern(
        Pattern.checkered((0.85, 0.85, 0.85), (0.25, 0.25, 0.3)).set_transform(pr(rotate(0.4, Axis.Y) * scale(0.7, 0.7, 0.7))))))
    w.objects.append(Shape.glass_sphere().set_transform(pr(translate(-1.5, 1.2, 0.5) * scale(1.2, 1.2, 1.2))))
    w.objects.append(Shape.sphere().set_transform(pr(translate(1.8, 0.9, 1.5) * scale(0.9, 0.9, 0.9))).set_material(
        Material(color=(0.9, 0.3, 0.2), reflective=0.5).with_pattern(Pattern.striped((0.9, 0.3, 0.2), (0.2, 0.3, 0.9)).set_transform(pr(scale(0.3, 0.3, 0.3))))))
    w.objects.append(Shape.cube().set_transform(pr(translate(3.5, 1., -1.) * rotate(0.5, Axis.Y))).set_material(
        Material(color=(0.3, 0.7, 0.3), transparency=0.5, reflective=0.2, refractive_index=1.3)))
    cyl = Shape.cylinder().set_minimum(0.).set_maximum(1.5).set_closed(True)
    w.objects.append(cyl.set_transform(pr(translate(-3.8, 0., 1.))).set_material(Material(color=(0.8, 0.7, 0.2)).with_pattern(
        Pattern.ring_gradient((0.8, 0.7, 0.2), (0.2, 0.2, 0.6)).set_transform(scale(0.4, 0.4, 0.4)))))
    cone = Shape.cone().set_minimum(-1.).set_maximum(0.).set_closed(True)
    w.objects.append(cone.set_transform(pr(translate(0.3, 1.6, 3.5) * scale(1., 1.6, 1.))).set_material(Material(color=(0.9, 0.5, 0.1))))
    csg = Shape.csg("difference", Shape.cube().set_material(Material(color=(0.9, 0.8, 0.2))),
                    Shape.sphere().set_transform(pr(scale(1.3, 1.3, 1.3))).set_material(Material(color=(0.9, 0.1, 0.1))))
    w.objects.append(csg.set_transform(pr(translate(0.2, 1., -2.) * rotate(0.6, Axis.Y))))
    g = Shape.group().set_transform(pr(translate(-0.5, 0.3, -4.) * rotate(-0.4, Axis.X)))
    inner = Shape.group().set_transform(pr(scale(0.8, 0.8, 0.8)))
    inner.add_child(Shape.sphere().set_transform(translate(0.9, 0., 0.) * scale(0.5, 0.5, 0.5)))
    inner.add_child(Shape.cylinder().set_minimum(-0.5).set_maximum(0.5).set_transform(pr(translate(-0.9, 0., 0.) * scale(0.3, 1., 0.3))))
    g.add_child(inner)
    g.add_child(Shape.sphere().set_transform(pr(translate(0., 1.1, 0.) * scale(0.4, 0.4, 0.4))))
    g.set_material(Material(color=(0.3, 0.4, 0.9), specular=0.7, reflective=0.15))
    w.objects.append(g)
    mesh, _ = rt.obj_to_group(scenes.random_mesh_obj(np.random.RandomState(77 + seed))[0])
    mesh.set_transform(pr(translate(2.5, 0.4, -3.5) * scale(0.8, 0.8, 0.8)))
    mesh.set_material(Material(color=(0.7, 0.6, 0.5), reflective=0.1))
    w.objects.append(mesh)
    cam = Camera(56, 36, PI / 3.)
    cam.transform = view_transform((0.5, 4.5, -11.), (0., 1., 0.), (0., 1., 0.))
    return w, cam


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_scene_with_projective_transforms(seed):
    w, cam = _projective_world(seed)
    g, _ = assert_image_parity(w, cam, 5)
    assert len(np.unique(g.hit_id)) >= 6 and g.stats.rays_reflect > 0
    assert_batches_match(w, n=1200, seed=seed, center=(0., 1., -1.), radius=10.)
    if seed == 0:   # a larger frame afterwards: the per-lane slab grows with the batch, the grid reaches its 256 workgroups
        big = Camera(224, 144, PI / 3.)
        big.transform = cam.transform
        assert_image_parity(w, big, 4)
        assert_image_parity(w, cam, 5)
    cam.transform = projective(cam.transform)        # and a projective camera on top
    assert_image_parity(w, cam, 3)
    cam.transform = scale(1., 0., 1.)                # and one without an inverse
    assert_image_parity(w, cam, 4)


# ------------------------------------------------------------------ ordinary scenes through the 4-component path
@pytest.mark.parametrize("builder,size", [(scenes.book_world, (64, 32)), (scenes.mixed_scene, (64, 36)), (scenes.cover_like_scene, (40, 40)),
                                          (scenes.smooth_teapot_alone, (40, 24)), (scenes.teapot_low, (40, 24))])
def test_ordinary_scenes_give_the_same_frame_on_either_path(builder, size, monkeypatch):
    w0, cam0 = builder(*size)
    fast = cam0.render(w0, 5)
    monkeypatch.setenv("RT_AMD_GENERAL", "1")
    w, cam = builder(*size)
    g, _ = assert_image_parity(w, cam, 5)
    # (two worlds built one after the other draw different object ids from the global counter: compare their ranks)
    assert np.array_equal(np.unique(g.hit_id, return_inverse=True)[1], np.unique(fast.hit_id, return_inverse=True)[1])
    assert np.array_equal(g.flags, fast.flags)
    assert float(np.nanmax(np.abs(g.rgb - fast.rgb))) < COLOR_TOL
    assert (g.stats.rays_reflect, g.stats.rays_refract, g.stats.rays_shadow, g.stats.shaded_nodes) == \
        (fast.stats.rays_reflect, fast.stats.rays_refract, fast.stats.rays_shadow, fast.stats.shaded_nodes)
    if builder in (scenes.book_world, scenes.mixed_scene):
        assert_batches_match(w, n=1500, seed=5)


@pytest.mark.parametrize("first", [0, 6, 12, 18])
def test_random_worlds_on_the_four_component_path(first, monkeypatch):
    """The fuzz suite's worlds (tests/scenes.py: every primitive, nested groups, CSG, meshes, ties, wild parameters)
    through the 4-component path: ids and flags exact, colours 1e-9, every ray class counted equal."""
    monkeypatch.setenv("RT_AMD_GENERAL", "1")
    failures = {}
    for seed in range(first, first + 6):
        make = (scenes.random_scene, lambda s, h, v: scenes.random_scene(s, h, v, wild=True), scenes.random_mesh_scene, scenes.random_tie_scene)[seed % 4]
        problems = fuzz_compare(seed, depth=4, builder=make)
        if problems:
            failures[seed] = problems
    assert not failures, failures


def test_reference_unit_test_tables_on_the_four_component_path(monkeypatch):
    """Every table of tests/golden/reference_kats.json that tests/test_gpu_kats.py replays through the C ABI, replayed once
    more with each scene forced onto the 4-component path: pins that path to the reference's own unit tests directly."""
    import test_gpu_kats as kats
    monkeypatch.setenv("RT_AMD_GENERAL", "1")
    ran = 0
    for name in sorted(dir(kats)):
        if name.startswith("test_"):
            getattr(kats, name)()
            ran += 1
    assert ran >= 20
