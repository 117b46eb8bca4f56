"""Pins the oracle (oracle/rt_oracle.c) — and with it the host mirror + flattener it is fed by — to the reference's
own unit-test vectors.  Every test names the reference test it transcribes
(file:line under /root/reference/src).  Tolerance 1e-5 = the reference's near_eq / PartialEq for Tuple and Color.
CPU only."""
import math

import numpy as np
import pytest

import oracle_bridge as ob
import rt_amd as rt
from rt_amd import Axis, Camera, Light, Material, Pattern, Shape, World, rotate, scale, translate, view_transform

PI = math.pi
S2 = math.sqrt(2.)
EPS = 1e-5


def near(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return bool(np.all(np.abs(a - b) < EPS))


def world_of(*shapes, lights=None):
    w = World()
    w.objects += list(shapes)
    w.lights += lights if lights is not None else [Light.point_light((-10., 10., -10.), rt.WHITE)]
    return w


# ------------------------------------------------------------------ camera.rs
def test_pixel_size():  # camera.rs:108-127
    assert abs(Camera(200, 125, PI / 2).pixel_size - 0.01) < EPS
    assert abs(Camera(125, 200, PI / 2).pixel_size - 0.01) < EPS


def test_ray_for_pixel():  # camera.rs:130-161
    cam = Camera(201, 101, PI / 2)
    o, d = ob.rays_for_pixels(cam, [100, 0], [50, 0])
    assert near(o[0], (0, 0, 0, 1)) and near(d[0], (0, 0, -1, 0))
    assert near(d[1], (0.66519, 0.33259, -0.66851, 0))
    cam.transform = rotate(PI / 4, Axis.Y) * translate(0., -2., 5.)
    o, d = ob.rays_for_pixels(cam, [100], [50])
    assert near(o[0], (0, 2, -5, 1)) and near(d[0], (S2 / 2, 0, -S2 / 2, 0))


def test_render_default_world():  # camera.rs:164-178
    w = World.default()
    cam = Camera(11, 11, PI / 2)
    cam.transform = view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0))
    img = ob.render(w, cam)
    assert near(img.pixel_at(5, 5), (0.38066, 0.47583, 0.2855))
    # F1 (camera.rs:65-66): the last row and column are never written
    assert not img.rgb[10, :, :].any() and not img.rgb[:, 10, :].any()


# ------------------------------------------------------------------ world.rs
def test_intersect_world():  # world.rs:186-203
    w = World.default()
    xs = ob.intersect_world(w, ((0, 0, -5), (0, 0, 1)))
    assert [t for t, _ in xs] == [4., 4.5, 5.5, 6.]


def test_shading_intersection():  # world.rs:206-218
    w = World.default()
    ray = ((0, 0, -5), (0, 0, 1))
    _, c = ob.eval_list(w, ray, [(4., w.objects[0])], 0, ob.SHADE_HIT)
    assert near(c, (0.38066, 0.47583, 0.2855))


def test_shading_from_inside():  # world.rs:221-234
    w = World.default()
    w.lights[0] = Light.point_light((0, 0.25, 0), rt.WHITE)
    _, c = ob.eval_list(w, ((0, 0, 0), (0, 0, 1)), [(0.5, w.objects[1])], 0, ob.SHADE_HIT)
    assert near(c, (0.90498, 0.90498, 0.90498))


def test_color_at():  # world.rs:237-274
    w = World.default()
    assert near(ob.color_at(w, ((0, 0, -5), (0, 1, 0))), (0, 0, 0))
    assert near(ob.color_at(w, ((0, 0, -5), (0, 0, 1))), (0.38066, 0.47583, 0.2855))
    w2 = World.default()
    for o in w2.objects:
        m = o.get_material()
        m.ambient = 1.
        o.set_material(m)
    inner = w2.objects[1].get_material().color
    assert near(ob.color_at(w2, ((0, 0, 0.75), (0, 0, -1))), inner)


def test_is_shadowed():  # world.rs:277-306
    w = World.default()
    assert ob.is_shadowed(w, (0, 10, 0)) == [False]
    assert ob.is_shadowed(w, (10, -10, 10)) == [True]
    assert ob.is_shadowed(w, (-20, 20, -20)) == [False]
    assert ob.is_shadowed(w, (-2, 2, -2)) == [False]


def test_shade_hit_in_shadow():  # world.rs:309-326
    s1 = Shape.sphere()
    s2 = Shape.sphere().set_transform(translate(0, 0, 10))
    w = world_of(s1, s2, lights=[Light.point_light((0, 0, -10), rt.WHITE)])
    _, c = ob.eval_list(w, ((0, 0, 5), (0, 0, 1)), [(4., s2)], 0, ob.SHADE_HIT)
    assert near(c, (0.1, 0.1, 0.1))


def test_no_shadow_when_not_casting():  # world.rs:575-588 (F8)
    sphere = Shape.sphere().set_transform(translate(0, 3, 0)).set_casts_shadow(False)
    w = world_of(sphere, Shape.plane(), lights=[Light.point_light((0, 10, 0), rt.WHITE)])
    assert ob.is_shadowed(w, (0, 0, 0)) == [False]


def reflective_plane_world(**mat):
    w = World.default()
    plane = Shape.plane().set_material(Material(**mat)).set_transform(translate(0, -1, 0))
    w.objects.append(plane)
    return w, plane


RAY45 = ((0, 0, -3), (0, -S2 / 2, S2 / 2))


def test_reflected_color():  # world.rs:329-364, 412-429
    w = World.default()
    shape = w.objects[1]
    m = shape.get_material()
    m.ambient = 1.
    shape.set_material(m)
    _, c = ob.eval_list(w, ((0, 0, 0), (0, 0, 1)), [(1., shape)], 0, ob.REFLECTED)
    assert near(c, (0, 0, 0))
    w, plane = reflective_plane_world(reflective=0.5)
    _, c = ob.eval_list(w, RAY45, [(S2, plane)], 0, ob.REFLECTED)
    assert near(c, (0.19033, 0.23791, 0.14274))
    _, c = ob.eval_list(w, RAY45, [(S2, plane)], 0, ob.REFLECTED, remaining=0)
    assert near(c, (0, 0, 0))


def test_shade_hit_reflective():  # world.rs:367-384
    w, plane = reflective_plane_world(reflective=0.5)
    _, c = ob.eval_list(w, RAY45, [(S2, plane)], 0, ob.SHADE_HIT)
    assert near(c, (0.87675, 0.92434, 0.82917))


def test_mutually_reflective_surfaces_terminate():  # world.rs:387-409
    lower = Shape.plane().set_material(Material(reflective=1.)).set_transform(translate(0, -1, 0))
    upper = Shape.plane().set_material(Material(reflective=1.)).set_transform(translate(0, 1, 0))
    w = world_of(lower, upper, lights=[Light.point_light((0, 0, 0), rt.WHITE)])
    c = ob.color_at(w, ((0, 0, 0), (0, 1, 0)))
    assert all(math.isfinite(x) for x in c)


def test_refracted_color_cases():  # world.rs:432-485
    w = World.default()
    shape = w.objects[0]
    ray = ((0, 0, -5), (0, 0, 1))
    _, c = ob.eval_list(w, ray, [(4., shape), (6., shape)], 0, ob.REFRACTED, remaining=5)
    assert near(c, (0, 0, 0))  # opaque
    w = World.default()
    w.objects[0].set_material(Material(color=(0.8, 1., 0.6), diffuse=0.7, specular=0.2, transparency=1., refractive_index=1.5))
    shape = w.objects[0]
    _, c = ob.eval_list(w, ray, [(4., shape), (6., shape)], 0, ob.REFRACTED, remaining=0)
    assert near(c, (0, 0, 0))  # max depth
    _, c = ob.eval_list(w, ((0, 0, S2 / 2), (0, 1, 0)), [(-S2 / 2, shape), (S2 / 2, shape)], 1, ob.REFRACTED)
    assert near(c, (0, 0, 0))  # total internal reflection


def test_refracted_color_with_refracted_ray():  # world.rs:488-511
    w = World.default()
    w.objects[0].set_material(Material(ambient=1.).with_pattern(Pattern.test()))
    w.objects[1].set_material(Material(transparency=1., refractive_index=1.5))
    a, b = w.objects
    xs = [(-0.9899, a), (-0.4899, b), (0.4899, b), (0.9899, a)]
    _, c = ob.eval_list(w, ((0, 0, 0.1), (0, 1, 0)), xs, 2, ob.REFRACTED)
    assert near(c, (0, 0.99887, 0.04722))


def floor_ball_world(**floor_mat):
    w = World.default()
    floor = Shape.plane().set_transform(translate(0, -1, 0)).set_material(Material(**floor_mat))
    ball = Shape.sphere().set_material(Material(color=(1, 0, 0), ambient=0.5)).set_transform(translate(0, -3.5, -0.5))
    w.objects += [floor, ball]
    return w, floor


def test_shade_hit_transparent():  # world.rs:514-541
    w, floor = floor_ball_world(transparency=0.5, refractive_index=1.5)
    _, c = ob.eval_list(w, RAY45, [(S2, floor)], 0, ob.SHADE_HIT)
    assert near(c, (0.93642, 0.68642, 0.68642))
    # the same through color_at on the real intersection list
    assert near(ob.color_at(w, RAY45), (0.93642, 0.68642, 0.68642))


def test_shade_hit_reflective_transparent_schlick():  # world.rs:544-572
    w, floor = floor_ball_world(reflective=0.5, transparency=0.5, refractive_index=1.5)
    _, c = ob.eval_list(w, RAY45, [(S2, floor)], 0, ob.SHADE_HIT)
    assert near(c, (0.93391, 0.69643, 0.69243))


# ------------------------------------------------------------------ intersection.rs
def test_prepare_computations_outside_inside():  # intersection.rs:235-294
    s = Shape.sphere()
    w = world_of(s)
    c, _ = ob.eval_list(w, ((0, 0, -5), (0, 0, 1)), [(4., s)], 0)
    assert c.t == 4. and near(c.point, (0, 0, -1)) and near(c.eye_vector, (0, 0, -1)) and near(c.normal_vector, (0, 0, -1))
    assert not c.inside and c.n1 == 1. and c.n2 == 1.
    c, _ = ob.eval_list(w, ((0, 0, 0), (0, 0, 1)), [(1., s)], 0)
    assert near(c.point, (0, 0, 1)) and near(c.eye_vector, (0, 0, -1)) and near(c.normal_vector, (0, 0, -1)) and c.inside


def test_over_and_under_point():  # intersection.rs:297-306, 356-366
    s = Shape.glass_sphere().set_transform(translate(0, 0, 1))
    c, _ = ob.eval_list(world_of(s), ((0, 0, -5), (0, 0, 1)), [(5., s)], 0)
    assert c.over_point[2] < -EPS / 2 and c.point[2] > c.over_point[2]
    assert c.under_point[2] > EPS / 2 and c.point[2] < c.under_point[2]


def test_reflect_vector():  # intersection.rs:309-320
    p = Shape.plane()
    c, _ = ob.eval_list(world_of(p), ((0, 1, -1), (0, -S2 / 2, S2 / 2)), [(S2, p)], 0)
    assert near(c.reflect_vector, (0, S2 / 2, S2 / 2))


def test_n1_n2():  # intersection.rs:323-353
    a = Shape.glass_sphere().set_transform(scale(2, 2, 2))
    b = Shape.glass_sphere().set_transform(translate(0, 0, -0.25)).set_material(Material(refractive_index=2.))
    c_ = Shape.glass_sphere().set_transform(scale(2, 2, 2)).set_material(Material(refractive_index=2.5))
    # NB: the reference test gives b and c default materials with only refractive_index changed
    w = world_of(a, b, c_)
    xs = [(2., a), (2.75, b), (3.25, c_), (4.75, b), (5.25, c_), (6., a)]
    expected = [(1., 1.5), (1.5, 2.), (2., 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.)]
    for i, (n1, n2) in enumerate(expected):
        c, _ = ob.eval_list(w, ((0, 0, -4), (0, 0, -1)), xs, i)
        assert abs(c.n1 - n1) < EPS and abs(c.n2 - n2) < EPS


def test_schlick():  # intersection.rs:369-410
    s = Shape.glass_sphere()
    w = world_of(s)
    _, r = ob.eval_list(w, ((0, 0, S2), (0, 1, 0)), [(-S2 / 2, s), (S2 / 2, s)], 1, ob.SCHLICK)
    assert abs(r - 1.) < EPS
    _, r = ob.eval_list(w, ((0, 0, 0), (0, 1, 0)), [(-1., s), (1., s)], 1, ob.SCHLICK)
    assert abs(r - 0.04) < EPS
    _, r = ob.eval_list(w, ((0, 0.99, -2), (0, 0, 1)), [(1.8589, s)], 0, ob.SCHLICK)
    assert abs(r - 0.48873) < EPS


# ------------------------------------------------------------------ primitives
def test_sphere_intersections():  # sphere.rs:107-196
    s = Shape.sphere()
    w = world_of(s)
    cases = [((0, 0, -5), (4., 6.)), ((0, 1, -5), (5., 5.)), ((0, 0, 0), (-1., 1.)), ((0, 0, 5), (-6., -4.))]
    for o, ts in cases:
        t, _ = ob.local_intersect(w, s, o, (0, 0, 1))
        assert near(t, ts)
    assert ob.local_intersect(w, s, (0, 2, -5), (0, 0, 1))[0] == []


def test_sphere_scaled_translated():  # sphere.rs (intersecting scaled / translated sphere)
    s = Shape.sphere().set_transform(scale(2, 2, 2))
    assert near([t for t, _ in ob.intersect_world(world_of(s), ((0, 0, -5), (0, 0, 1)))], (3., 7.))
    s = Shape.sphere().set_transform(translate(5, 0, 0))
    assert ob.intersect_world(world_of(s), ((0, 0, -5), (0, 0, 1))) == []


def test_plane():  # plane.rs:93-144
    p = Shape.plane()
    w = world_of(p)
    assert ob.local_intersect(w, p, (0, 10, 0), (0, 0, 1))[0] == []
    assert ob.local_intersect(w, p, (0, 0, 0), (0, 0, 1))[0] == []
    assert near(ob.local_intersect(w, p, (0, 1, 0), (0, -1, 0))[0], [1.])
    assert near(ob.local_intersect(w, p, (0, -1, 0), (0, 1, 0))[0], [1.])
    assert near(ob.local_normal_at(w, p, (10, 0, -10)), (0, 1, 0))


def test_cube():  # cube.rs:108-185
    c = Shape.cube()
    w = world_of(c)
    hits = [((5, 0.5, 0), (-1, 0, 0), (4, 6)), ((-5, 0.5, 0), (1, 0, 0), (4, 6)), ((0.5, 5, 0), (0, -1, 0), (4, 6)),
            ((0.5, -5, 0), (0, 1, 0), (4, 6)), ((0.5, 0, 5), (0, 0, -1), (4, 6)), ((0.5, 0, -5), (0, 0, 1), (4, 6)),
            ((0, 0.5, 0), (0, 0, 1), (-1, 1))]
    for o, d, ts in hits:
        assert near(ob.local_intersect(w, c, o, d)[0], ts)
    misses = [((-2, 0, 0), (0.2673, 0.5345, 0.8018)), ((0, -2, 0), (0.8018, 0.2673, 0.5345)),
              ((0, 0, -2), (0.5345, 0.8018, 0.2673)), ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)), ((2, 2, 0), (-1, 0, 0))]
    for o, d in misses:
        assert ob.local_intersect(w, c, o, d)[0] == []
    normals = [((1, 0.5, -0.8), (1, 0, 0)), ((-1, -0.2, 0.9), (-1, 0, 0)), ((-0.4, 1, -0.1), (0, 1, 0)),
               ((0.3, -1, -0.7), (0, -1, 0)), ((-0.6, 0.3, 1), (0, 0, 1)), ((0.4, 0.4, -1), (0, 0, -1)),
               ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))]
    for p, n in normals:
        assert near(ob.local_normal_at(w, c, p), n)


def norm(v):
    v = np.asarray(v, dtype=float)
    return tuple(v / np.sqrt((v * v).sum()))


def test_cylinder():  # cylinder.rs:145-311
    cyl = Shape.cylinder()
    w = world_of(cyl)
    for o, d in [((1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, -5), (1, 1, 1))]:
        assert ob.local_intersect(w, cyl, o, norm(d))[0] == []
    for o, d, ts in [((1, 0, -5), (0, 0, 1), (5, 5)), ((0, 0, -5), (0, 0, 1), (4, 6)), ((0.5, 0, -5), (0.1, 1, 1), (6.80798, 7.08872))]:
        assert near(ob.local_intersect(w, cyl, o, norm(d))[0], ts)
    for p, n in [((1, 0, 0), (1, 0, 0)), ((0, 5, -1), (0, 0, -1)), ((0, -2, 1), (0, 0, 1)), ((-1, 1, 0), (-1, 0, 0))]:
        assert near(ob.local_normal_at(w, cyl, p), n)
    trunc = Shape.cylinder().set_minimum(1.).set_maximum(2.)
    w = world_of(trunc)
    rays = [((0, 1.5, 0), (0.1, 1, 0)), ((0, 3, -5), (0, 0, 1)), ((0, 0, -5), (0, 0, 1)), ((0, 2, -5), (0, 0, 1)),
            ((0, 1, -5), (0, 0, 1)), ((0, 1.5, -2), (0, 0, 1))]
    assert [len(ob.local_intersect(w, trunc, o, norm(d))[0]) for o, d in rays] == [0, 0, 0, 0, 0, 2]
    capped = Shape.cylinder().set_minimum(1.).set_maximum(2.).set_closed(True)
    w = world_of(capped)
    rays = [((0, 3, 0), (0, -1, 0)), ((0, 3, -2), (0, -1, 2)), ((0, 4, -2), (0, -1, 1)), ((0, 0, -2), (0, 1, 2)), ((0, -1, -2), (0, 1, 1))]
    assert [len(ob.local_intersect(w, capped, o, norm(d))[0]) for o, d in rays] == [2] * 5
    caps = [((0, 1, 0), (0, -1, 0)), ((0.5, 1, 0), (0, -1, 0)), ((0, 1, 0.5), (0, -1, 0)), ((0, 2, 0), (0, 1, 0)),
            ((0.5, 2, 0), (0, 1, 0)), ((0, 2, 0.5), (0, 1, 0))]
    for p, n in caps:
        assert near(ob.local_normal_at(w, capped, p), n)


def test_cone():  # cone.rs:156-236
    cone = Shape.cone()
    w = world_of(cone)
    for o, d, ts in [((0, 0, -5), (0, 0, 1), (5, 5)), ((0, 0, -5), (1, 1, 1), (8.66025, 8.66025)),
                     ((1, 1, -5), (-0.5, -1, 1), (4.55006, 49.44994))]:
        assert near(ob.local_intersect(w, cone, o, norm(d))[0], ts)
    t, _ = ob.local_intersect(w, cone, (0, 0, -1), norm((0, 1, 1)))
    assert len(t) == 1 and abs(t[0] - 0.35355) < EPS
    capped = Shape.cone().set_minimum(-0.5).set_maximum(0.5).set_closed(True)
    w2 = world_of(capped)
    rays = [((0, 0, -5), (0, 1, 0)), ((0, 0, -0.25), (0, 1, 1)), ((0, 0, -0.25), (0, 1, 0))]
    assert [len(ob.local_intersect(w2, capped, o, norm(d))[0]) for o, d in rays] == [0, 2, 4]
    for p, n in [((0, 0, 0), (0, 0, 0)), ((1, 1, 1), (1, -S2, 1)), ((-1, -1, 0), (-1, 1, 0))]:
        assert near(ob.local_normal_at(w, cone, p), n)


def test_triangle():  # triangle.rs:105-195
    t = Shape.triangle((0, 1, 0), (-1, 0, 0), (1, 0, 0))
    data = t.triangle_data()
    assert near(data[3], (-1, -1, 0)) and near(data[4], (1, -1, 0)) and near(data[5], (0, 0, -1))
    w = world_of(t)
    for p in [(0, 0.5, 0), (-0.5, 0.75, 0), (0.5, 0.25, 0)]:
        assert near(ob.local_normal_at(w, t, p), (0, 0, -1))
    assert ob.local_intersect(w, t, (0, -1, -2), (0, 1, 0))[0] == []   # parallel
    assert ob.local_intersect(w, t, (1, 1, -2), (0, 0, 1))[0] == []    # p1-p3 edge
    assert ob.local_intersect(w, t, (-1, 1, -2), (0, 0, 1))[0] == []   # p1-p2 edge
    assert ob.local_intersect(w, t, (0, -1, -2), (0, 0, 1))[0] == []   # p2-p3 edge
    assert near(ob.local_intersect(w, t, (0, 0.5, -2), (0, 0, 1))[0], [2.])


def smooth_tri():
    return Shape.smooth_triangle((0, 1, 0), (-1, 0, 0), (1, 0, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0))


def test_smooth_triangle():  # smooth_triangle.rs:151-196
    t = smooth_tri()
    w = world_of(t)
    ts, uv = ob.local_intersect(w, t, (-0.2, 0.3, -2), (0, 0, 1))
    assert len(ts) == 1 and abs(uv[0][0] - 0.45) < EPS and abs(uv[0][1] - 0.25) < EPS
    assert near(ob.normal_at(w, t, (0, 0, 0), 0.45, 0.25), (-0.5547, 0.83205, 0))
    c, _ = ob.eval_list(w, ((-0.2, 0.3, -2), (0, 0, 1)), [(1., t, 0.45, 0.25)], 0)
    assert near(c.normal_vector, (-0.5547, 0.83205, 0))


def test_smooth_triangle_uses_first_intersections_uv():  # F5: intersection.rs:40 passes intersections[0]
    t = smooth_tri()
    w = world_of(t)
    ray = ((-0.2, 0.3, -2), (0, 0, 1))
    c_first, _ = ob.eval_list(w, ray, [(-1., t, 0.1, 0.2), (1., t, 0.45, 0.25)], 1)
    n_of_first = ob.normal_at(w, t, (0, 0, 0), 0.1, 0.2)
    assert near(np.abs(c_first.normal_vector), np.abs(n_of_first))
    assert not near(np.abs(c_first.normal_vector), (0.5547, 0.83205, 0))


def test_smooth_triangle_first_without_uv_is_flagged():  # F5b: reference unwrap()s a None
    t = smooth_tri()
    s = Shape.sphere().set_transform(translate(0, 0, -6))
    w = world_of(s, t)
    c, _ = ob.eval_list(w, ((-0.2, 0.3, -2), (0, 0, 1)), [(-5., s), (2., t, 0.45, 0.25)], 1)
    assert c.flags & rt.RT_FLAG_REF_PANIC_UV


# ------------------------------------------------------------------ shape.rs parent chain, group.rs
def test_world_to_object_and_normal_to_world():  # shape.rs:678-730
    g1 = Shape.group().set_transform(rotate(PI / 2, Axis.Y))
    g2 = Shape.group().set_transform(scale(2, 2, 2))
    s = Shape.sphere().set_transform(translate(5, 0, 0))
    g2.add_child(s)
    g1.add_child(g2)
    w = world_of(g1)
    assert near(ob.world_to_object(w, s, (-2, 0, -10)), (0, 0, -1))
    g1 = Shape.group().set_transform(rotate(PI / 2, Axis.Y))
    g2 = Shape.group().set_transform(scale(1, 2, 3))
    s = Shape.sphere().set_transform(translate(5, 0, 0))
    g2.add_child(s)
    g1.add_child(g2)
    w = world_of(g1)
    r3 = math.sqrt(3) / 3
    assert near(ob.normal_to_world(w, s, (r3, r3, r3)), (0.28571, 0.42857, -0.85714))
    assert near(ob.normal_at(w, s, (1.7321, 1.1547, -5.5774)), (0.2857, 0.42854, -0.85716))


def test_group_intersections():  # group.rs:192-241
    assert ob.intersect_world(world_of(Shape.group()), ((0, 0, 0), (0, 0, 1))) == []
    g = Shape.group()
    s1 = Shape.sphere()
    s2 = Shape.sphere().set_transform(translate(0, 0, -3))
    s3 = Shape.sphere().set_transform(translate(5, 0, 0))
    for s in (s1, s2, s3):
        g.add_child(s)
    xs = ob.intersect_world(world_of(g), ((0, 0, -5), (0, 0, 1)))
    assert [i for _, i in xs] == [s2.get_id(), s2.get_id(), s1.get_id(), s1.get_id()]
    g = Shape.group().set_transform(scale(2, 2, 2))
    s = Shape.sphere().set_transform(translate(5, 0, 0))
    g.add_child(s)
    assert len(ob.intersect_world(world_of(g), ((10, 0, -10), (0, 0, 1)))) == 2


def test_group_box_gate_skips_children():  # group.rs:265-300 (TestShape replaced by a sphere the box excludes)
    g = Shape.group()
    s = Shape.sphere()
    g.add_child(s)
    w = world_of(g)
    assert ob.intersect_world(w, ((0, 0, -5), (0, 1, 0))) == []
    assert len(ob.intersect_world(w, ((0, 0, -5), (0, 0, 1)))) == 2


def test_material_comes_from_root_ancestor():  # F7: shape.rs:163-184
    g = Shape.group().set_material(Material(color=(1, 0, 0), ambient=1., diffuse=0., specular=0.))
    s = Shape.sphere().set_material(Material(color=(0, 1, 0), ambient=1., diffuse=0., specular=0.))
    g.add_child(s)
    assert near(ob.color_at(world_of(g), ((0, 0, -5), (0, 0, 1))), (1, 0, 0))


def test_hexagon_parity_of_gates():  # lib.rs:91-126: finite group boxes on two levels
    w = world_of(Shape.hexagon(), lights=[Light.point_light((0, 10, 0), rt.WHITE)])
    xs = ob.intersect_world(w, ((0, 3, -3), norm((0, -2.7, 3))))
    assert len(xs) % 2 == 0


# ------------------------------------------------------------------ bound.rs
BOX_DIRS = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1), (0, 0, 1), (2, 4, 6), (6, 2, 4), (4, 6, 2),
            (0, 0, -1), (0, -1, 0), (-1, 0, 0)]
BOX_EXPECT = [True] * 7 + [False] * 6


def test_bound_intersects_unit_box():  # bound.rs:279-309
    origins = [(5, 0.5, 0), (-5, 0.5, 0), (0.5, 5, 0), (0.5, -5, 0), (0.5, 0, 5), (0.5, 0, -5), (0, 0.5, 0), (-2, 0, 0),
               (0, -2, 0), (0, 0, -2), (2, 0, 2), (0, 2, 2), (2, 2, 0)]
    got = [ob.bound_intersects((-1, -1, -1), (1, 1, 1), o, norm(d)) for o, d in zip(origins, BOX_DIRS)]
    assert got == BOX_EXPECT


def test_bound_intersects_noncubic_box():  # bound.rs:312-342
    origins = [(1.5, 1, 2), (-5, -1, 4), (7, 6, 5), (9, -5, 6), (8, 2, 1.2), (6, 0, -5), (8, 1, 3.5), (9, -1, -8), (8, 3, -4),
               (9, -1, -2), (4, 0, 9), (8, 6, -1), (1.2, 5, 4)]
    got = [ob.bound_intersects((5, -2, 0), (11, 4, 7), o, norm(d)) for o, d in zip(origins, BOX_DIRS)]
    assert got == BOX_EXPECT


# ------------------------------------------------------------------ csg.rs
def test_csg_truth_table():  # csg.rs:176-220
    table = {"union": [0, 1, 0, 1, 0, 0, 1, 1], "intersection": [1, 0, 1, 0, 1, 1, 0, 0], "difference": [0, 1, 0, 1, 1, 1, 0, 0]}
    for op, results in table.items():
        i = 0
        for lhit in (True, False):
            for inl in (True, False):
                for inr in (True, False):
                    assert ob.intersection_allowed(op, lhit, inl, inr) == bool(results[i])
                    i += 1
    assert not ob.intersection_allowed("bogus", True, False, False)  # csg.rs:102


def test_csg_ray_misses_and_hits():  # csg.rs:242-273
    s, c = Shape.sphere(), Shape.cube()
    csg = Shape.csg("union", s, c)
    assert ob.intersect_world(world_of(csg), ((0, 2, -5), (0, 0, 1))) == []
    s, c = Shape.sphere(), Shape.cube().set_transform(translate(0, 0, 0.5))
    csg = Shape.csg("union", s, c)
    xs = ob.intersect_world(world_of(csg), ((0, 0, -5), (0, 0, 1)))
    assert len(xs) == 2 and near(xs[0][0], 4.) and xs[0][1] == s.get_id() and near(xs[1][0], 6.5) and xs[1][1] == c.get_id()


def test_csg_filtering_through_real_rays():  # csg.rs:223-240 (lists (1,2,3,4) -> union 0,3 / intersection 1,2 / difference 0,1)
    for op, keep in (("union", (0, 3)), ("intersection", (1, 2)), ("difference", (0, 1))):
        a = Shape.sphere()                                   # t = 4, 6
        b = Shape.sphere().set_transform(translate(0, 0, 0.5))  # t = 4.5, 6.5
        csg = Shape.csg(op, a, b)
        xs = ob.intersect_world(world_of(csg), ((0, 0, -5), (0, 0, 1)))
        full = [4., 4.5, 6., 6.5]
        assert near([t for t, _ in xs], [full[keep[0]], full[keep[1]]])


# ------------------------------------------------------------------ material.rs lighting
def lighting_world(light_pos, material=None):
    s = Shape.sphere()
    if material:
        s.set_material(material)
    return world_of(s, lights=[Light.point_light(light_pos, rt.WHITE)]), s


def test_lighting():  # material.rs:161-260
    P, N = (0, 0, 0), (0, 0, -1)
    w, s = lighting_world((0, 0, -10))
    assert near(ob.lighting(w, s, 0, P, (0, 0, -1), N, False), (1.9, 1.9, 1.9))
    assert near(ob.lighting(w, s, 0, P, (0, S2 / 2, -S2 / 2), N, False), (1., 1., 1.))
    w, s = lighting_world((0, 10, -10))
    assert near(ob.lighting(w, s, 0, P, (0, 0, -1), N, False), (0.7364, 0.7364, 0.7364))
    assert near(ob.lighting(w, s, 0, P, (0, -S2 / 2, -S2 / 2), N, False), (1.6364, 1.6364, 1.6364))
    w, s = lighting_world((0, 0, 10))
    assert near(ob.lighting(w, s, 0, P, (0, 0, -1), N, False), (0.1, 0.1, 0.1))
    w, s = lighting_world((0, 0, -10))
    assert near(ob.lighting(w, s, 0, P, (0, 0, -1), N, True), (0.1, 0.1, 0.1))


def test_lighting_with_pattern():  # material.rs:262-277
    m = Material(ambient=1., diffuse=0., specular=0.).with_pattern(Pattern.striped(rt.WHITE, rt.BLACK))
    w, s = lighting_world((0, 0, -10), m)
    assert near(ob.lighting(w, s, 0, (0.9, 0, 0), (0, 0, -1), (0, 0, -1), False), (1, 1, 1))
    assert near(ob.lighting(w, s, 0, (1.1, 0, 0), (0, 0, -1), (0, 0, -1), False), (0, 0, 0))


# ------------------------------------------------------------------ pattern.rs
def pat_world(pattern, transform=None):
    s = Shape.sphere().set_material(Material().with_pattern(pattern))
    if transform:
        s.set_transform(transform)
    return world_of(s), s


def test_stripe_pattern():  # pattern.rs:226-290
    w, s = pat_world(Pattern.striped(rt.WHITE, rt.BLACK))
    for p, c in [((0, 0, 0), 1), ((0, 1, 0), 1), ((0, 2, 0), 1), ((0, 0, 1), 1), ((0, 0, 2), 1), ((0.9, 0, 0), 1), ((1, 0, 0), 0),
                 ((-0.1, 0, 0), 0), ((-1, 0, 0), 0), ((-1.1, 0, 0), 1)]:
        assert near(ob.pattern_at_shape(w, s, p), (c, c, c))


def test_pattern_transforms():  # pattern.rs:292-350
    w, s = pat_world(Pattern.test(), scale(2, 2, 2))
    assert near(ob.pattern_at_shape(w, s, (2, 3, 4)), (1, 1.5, 2))
    w, s = pat_world(Pattern.test().set_transform(scale(2, 2, 2)))
    assert near(ob.pattern_at_shape(w, s, (2, 3, 4)), (1, 1.5, 2))
    w, s = pat_world(Pattern.test().set_transform(translate(0.5, 1, 1.5)), scale(2, 2, 2))
    assert near(ob.pattern_at_shape(w, s, (2.5, 3, 3.5)), (0.75, 0.5, 0.25))


def test_gradient_ring_checkers():  # pattern.rs:352-426
    w, s = pat_world(Pattern.gradient(rt.WHITE, rt.BLACK))
    for x, c in [(0, 1), (0.25, 0.75), (0.5, 0.5), (0.75, 0.25)]:
        assert near(ob.pattern_at_shape(w, s, (x, 0, 0)), (c, c, c))
    w, s = pat_world(Pattern.ring(rt.WHITE, rt.BLACK))
    for p, c in [((0, 0, 0), 1), ((1, 0, 0), 0), ((0, 0, 1), 0), ((0.708, 0, 0.708), 0)]:
        assert near(ob.pattern_at_shape(w, s, p), (c, c, c))
    w, s = pat_world(Pattern.checkered(rt.WHITE, rt.BLACK))
    for p, c in [((0, 0, 0), 1), ((0.99, 0, 0), 1), ((1.01, 0, 0), 0), ((0, 0.99, 0), 1), ((0, 1.01, 0), 0), ((0, 0, 0.99), 1),
                 ((0, 0, 1.01), 0)]:
        assert near(ob.pattern_at_shape(w, s, p), (c, c, c))


# ------------------------------------------------------------------ reference quirks no reference test covers (follow the code text)
def test_f6_double_match_last_wins():
    # two entries of the hit object within 1e-5 of the hit's t: n1/n2 are overwritten by the second (intersection.rs:54-80)
    a = Shape.glass_sphere()
    w = world_of(a)
    c, _ = ob.eval_list(w, ((0, 0, -5), (0, 0, 1)), [(4., a), (4.000001, a)], 0)
    # first match: n1 = 1, push a -> n2 = 1.5; second match: n1 = 1.5, remove a -> n2 = 1
    assert c.n1 == 1.5 and c.n2 == 1.


def test_f14_unbounded_cylinder_is_not_equal_to_itself():
    cyl = Shape.cylinder().set_material(Material(transparency=1., refractive_index=1.5))
    assert not (cyl == cyl.clone())   # inf - inf = NaN in near_eq (cylinder.rs:26-33)
    w = world_of(cyl)
    c, _ = ob.eval_list(w, ((0, 0, -5), (0, 0, 1)), [(4., cyl), (6., cyl)], 0)
    assert c.n1 == 0. and c.n2 == 0.   # intersection.rs:54-55 initial values survive
    bounded = Shape.cylinder().set_minimum(0.).set_maximum(1.)
    assert bounded == bounded.clone()
