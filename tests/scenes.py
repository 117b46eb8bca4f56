"""Scene builders for the configurations of BASELINE.json / SURVEY.md section 8(d), written against the
reference-shaped API of rt_amd.  All deterministic (no RNG).  Mesh assets live in tests/golden/ (they are data
files the reference ships: teapot.obj, teapot-low.obj, test.obj, triangles.obj)."""
import math
import os

import numpy as np

import rt_amd as rt
from rt_amd import (Axis, Camera, Light, Material, Pattern, Shape, World, rotate, scale, translate, view_transform)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PI = math.pi


def read_asset(name):
    with open(os.path.join(GOLDEN, name), "r") as f:
        return f.read()


def book_world(hsize=320, vsize=160):
    """C1: the Ray Tracer Challenge sphere-and-plane world — three spheres of src/main.rs:220-236 on a checkered
    plane, patterns on every sphere, one light, camera of src/main.rs:248-249 at pi/3."""
    w = World()
    w.lights.append(Light.point_light((-10., 10., -10.), rt.WHITE))
    floor = Shape.plane().set_material(Material(specular=0.).with_pattern(Pattern.checkered(rt.WHITE, rt.BLACK)))
    middle = Shape.sphere().set_transform(translate(-0.5, 1., 0.5))
    middle.set_material(Material(color=(0.1, 1., 0.5), diffuse=0.7, specular=0.3).with_pattern(
        Pattern.striped((0.1, 1., 0.5), (0.9, 0.2, 0.3)).set_transform(scale(0.25, 0.25, 0.25))))
    right = Shape.sphere().set_transform(translate(1.5, 0.5, -0.5) * scale(0.5, 0.5, 0.5))
    right.set_material(Material(color=(0.5, 1., 0.1), diffuse=0.7, specular=0.3).with_pattern(
        Pattern.gradient((0.5, 1., 0.1), (0.1, 0.2, 1.))))
    left = Shape.sphere().set_transform(translate(-1.5, 0.33, -0.75) * scale(0.33, 0.33, 0.33))
    left.set_material(Material(color=(1., 0.8, 0.1), diffuse=0.7, specular=0.3, reflective=0.3).with_pattern(
        Pattern.ring((1., 0.8, 0.1), (0.2, 0.2, 0.2)).set_transform(scale(0.2, 0.2, 0.2))))
    w.objects += [floor, middle, right, left]
    cam = Camera(hsize, vsize, PI / 3.)
    cam.transform = view_transform((0., 1.5, -5.), (0., 1., 0.), (0., 1., 0.))
    return w, cam


def teapot_low(hsize=1920, vsize=1080):
    """C2: teapot-low.obj through the unchanged parser (16 + 112 x 135 = 15,136 flat triangles, F3) in its Group
    (infinite box, F4), Z-up mesh rotated upright and scaled to 0.1 (det >= 1e-5, F11), one light."""
    w = World()
    w.lights.append(Light.point_light((-5., 10., -10.), rt.WHITE))
    teapot, counts = rt.obj_to_group(read_asset("teapot-low.obj"))
    teapot.set_transform(rotate(-PI / 2., Axis.X) * scale(0.1, 0.1, 0.1))
    teapot.set_material(Material(color=(0.8, 0.3, 0.2), diffuse=0.7, specular=0.4, shininess=50.))
    w.objects.append(teapot)
    cam = Camera(hsize, vsize, PI / 3.)
    cam.transform = view_transform((0., 2.5, -5.), (0., 0.6, 0.), (0., 1., 0.))
    return w, cam


def glass_material():
    # blue glass sphere of src/main.rs:511-516
    return Material(color=(0., 0., 0.2), ambient=0., diffuse=0.4, specular=0.9, shininess=300., reflective=0.9,
                    transparency=0.9, refractive_index=1.5)


def mirror_material():
    # mirror of src/main.rs:658-659
    return Material(color=rt.BLACK, diffuse=0., ambient=0., specular=1., shininess=300., reflective=1.)


def teapot_scene(hsize=1920, vsize=1080, obj_text=None):
    """C3 (the headline workload): teapot.obj (6320 flat triangles as the reference parser yields them, F2) with a
    glass sphere, a mirror sphere, a checkered floor and one light."""
    w = World()
    w.lights.append(Light.point_light((-6., 9., -8.), rt.WHITE))
    floor = Shape.plane().set_material(
        Material(specular=0., reflective=0.1).with_pattern(Pattern.checkered((0.85, 0.85, 0.85), (0.25, 0.25, 0.25))))
    teapot, counts = rt.obj_to_group(obj_text if obj_text is not None else read_asset("teapot.obj"))
    teapot.set_transform(rotate(PI / 7., Axis.Y))
    teapot.set_material(Material(color=(0.9, 0.55, 0.2), diffuse=0.7, specular=0.5, shininess=80.))
    glass = Shape.sphere().set_transform(translate(-3.6, 1.2, -2.6) * scale(1.2, 1.2, 1.2)).set_material(glass_material())
    mirror = Shape.sphere().set_transform(translate(4.2, 1.5, 0.8) * scale(1.5, 1.5, 1.5)).set_material(mirror_material())
    w.objects += [floor, teapot, glass, mirror]
    cam = Camera(hsize, vsize, PI / 3.)
    cam.transform = view_transform((0.5, 4.2, -10.5), (0.2, 1.2, 0.), (0., 1., 0.))
    return w, cam


def smooth_teapot_obj():
    """teapot.obj with generated area-weighted vertex normals and `a//a b//b c//c` faces (F2: no bundled OBJ
    yields SmoothTriangles, so the smooth kernel gets a generated asset).  Pure text transformation of data."""
    verts, faces = [], []
    for line in read_asset("teapot.obj").splitlines():
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            verts.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            faces.append([int(x) for x in p[1:4]])
    v = np.array(verts)
    n = np.zeros_like(v)
    for a, b, c in faces:
        fn = np.cross(v[b - 1] - v[a - 1], v[c - 1] - v[a - 1])
        for i in (a, b, c):
            n[i - 1] += fn
    ln = np.linalg.norm(n, axis=1)
    n[ln > 0] /= ln[ln > 0][:, None]
    n[ln == 0] = (0., 1., 0.)
    out = [f"v {x:.6f} {y:.6f} {z:.6f}" for x, y, z in v]
    out += [f"vn {x:.6f} {y:.6f} {z:.6f}" for x, y, z in n]
    out += [f"f {a}//{a} {b}//{b} {c}//{c}" for a, b, c in faces]
    return "\n".join(out) + "\n"


def smooth_teapot_alone(hsize=240, vsize=135):
    """Smooth teapot with nothing else in the world: xs[0] is always a smooth-triangle hit, so the reference does
    not panic (F5b) and its `u,v of xs[0]` rule (F5a) is exercised on every multi-hit ray."""
    w = World()
    w.lights.append(Light.point_light((-6., 9., -8.), rt.WHITE))
    teapot, _ = rt.obj_to_group(smooth_teapot_obj())
    teapot.set_material(Material(color=(0.9, 0.55, 0.2), diffuse=0.7, specular=0.5, shininess=80., reflective=0.2))
    w.objects.append(teapot)
    cam = Camera(hsize, vsize, PI / 3.)
    cam.transform = view_transform((0.5, 4.2, -9.5), (0.2, 1.4, 0.), (0., 1., 0.))
    return w, cam


def mixed_scene(hsize=2560, vsize=1440):
    """C4: cube, closed cylinder (0..1), closed cone (-1..0), CSG difference(cube, sphere), a glass sphere holding
    an air bubble (src/main.rs:411-421) inside a larger glass cube (3-deep n1/n2), hexagon group with finite
    gate boxes (src/lib.rs:91-126), two lights."""
    w = World()
    w.lights.append(Light.point_light((-8., 9., -9.), (0.8, 0.8, 0.8)))
    w.lights.append(Light.point_light((9., 7., -6.), (0.4, 0.4, 0.45)))
    floor = Shape.plane().set_material(
        Material(specular=0.1, reflective=0.15).with_pattern(
            Pattern.checkered((0.8, 0.8, 0.8), (0.3, 0.3, 0.35)).set_transform(rotate(0.3, Axis.Y))))
    cube = Shape.cube().set_transform(translate(-4., 1., 2.) * rotate(0.5, Axis.Y)).set_material(
        Material(color=(0.8, 0.2, 0.2), reflective=0.2).with_pattern(
            Pattern.gradient((0.8, 0.2, 0.2), (0.2, 0.2, 0.8)).set_transform(scale(2., 2., 2.) * translate(0.5, 0., 0.))))
    cyl = Shape.cylinder().set_minimum(0.).set_maximum(1.).set_closed(True)
    cyl.set_transform(translate(3.5, 0., 2.5) * scale(1., 2.5, 1.)).set_material(
        Material(color=(0.2, 0.7, 0.3), specular=0.6).with_pattern(
            Pattern.ring_gradient((0.2, 0.7, 0.3), (0.9, 0.9, 0.2)).set_transform(scale(0.4, 0.4, 0.4))))
    cone = Shape.cone().set_minimum(-1.).set_maximum(0.).set_closed(True)
    cone.set_transform(translate(0.5, 2., 4.5) * scale(1.2, 2., 1.2)).set_material(Material(color=(0.9, 0.6, 0.1), shininess=30.))
    c_left = Shape.cube().set_material(Material(color=(0.9, 0.8, 0.2)))
    c_right = Shape.sphere().set_transform(scale(1.3, 1.3, 1.3)).set_material(Material(color=(0.9, 0.1, 0.1)))
    csg = Shape.csg("difference", c_left, c_right)
    csg.set_transform(translate(-1.2, 1., -1.5) * rotate(0.6, Axis.Y) * rotate(0.3, Axis.X))
    csg.set_material(Material(color=(0.9, 0.75, 0.2), diffuse=0.8, specular=0.4, reflective=0.1))
    glass = Material(diffuse=0.1, shininess=300., reflective=1., refractive_index=1.52, transparency=1.)
    air = Material(diffuse=0.1, shininess=300., reflective=1., refractive_index=1., transparency=1.)
    glass_cube = Shape.cube().set_transform(translate(2.2, 1.3, -2.2) * rotate(0.4, Axis.Y) * scale(1.3, 1.3, 1.3))
    glass_cube.set_material(Material(color=(0.9, 1., 0.9), ambient=0., diffuse=0.2, specular=0.9, shininess=300.,
                                     reflective=0.6, transparency=0.8, refractive_index=1.3))
    glass_cube.set_casts_shadow(False)
    ball = Shape.glass_sphere().set_transform(translate(2.2, 1.3, -2.2)).set_material(glass)
    bubble = Shape.sphere().set_transform(translate(2.2, 1.3, -2.2) * scale(0.5, 0.5, 0.5)).set_material(air)
    hexa = Shape.hexagon().set_transform(translate(-2.5, 0.6, -4.) * rotate(-PI / 6., Axis.X) * scale(1.4, 1.4, 1.4))
    hexa.set_material(Material(color=(0.2, 0.4, 0.9), specular=0.7))
    w.objects += [floor, cube, cyl, cone, csg, glass_cube, ball, bubble, hexa]
    cam = Camera(hsize, vsize, PI / 3.)
    cam.transform = view_transform((0.5, 5., -11.), (0., 1., 0.), (0., 1., 0.))
    return w, cam


def instanced_grid(hsize=7680, vsize=4320, grid=4):
    """C5: grid x grid separately parsed teapot.obj + triangles.obj groups, each with its own translate*scale
    (scale >= 0.03, F11), mirror floor.  Depth 8 is chosen by the caller (render(..., depth=8))."""
    w = World()
    w.lights.append(Light.point_light((-10., 20., -15.), rt.WHITE))
    floor = Shape.plane().set_material(Material(color=(0.3, 0.3, 0.35), specular=0.2, reflective=0.6))
    w.objects.append(floor)
    tea, tri = read_asset("teapot.obj"), read_asset("triangles.obj")
    span = 8.0
    for gz in range(grid):
        for gx in range(grid):
            x = (gx - (grid - 1) / 2.) * span
            z = gz * span
            t, _ = rt.obj_to_group(tea)
            t.set_transform(translate(x, 0., z) * rotate(0.4 * (gx + 2 * gz), Axis.Y) * scale(0.9, 0.9, 0.9))
            t.set_material(Material(color=(0.3 + 0.6 * gx / max(grid - 1, 1), 0.5, 0.9 - 0.6 * gz / max(grid - 1, 1)),
                                    diffuse=0.7, specular=0.5, shininess=60., reflective=0.15))
            w.objects.append(t)
            g, _ = rt.obj_to_group(tri)
            g.set_transform(translate(x, 3.4, z) * scale(1.5, 1.5, 1.5))
            g.set_material(Material(color=(0.9, 0.9, 0.3), reflective=0.3))
            w.objects.append(g)
    cam = Camera(hsize, vsize, PI / 3.)
    cam.transform = view_transform((0., 9., -16.), (0., 1.5, span * (grid - 1) / 2.), (0., 1., 0.))
    return w, cam


def many_objects_scene(seed=0, hsize=96, vsize=54, n=40):
    """A world of many BOUNDED top-level objects (spheres, closed / truncated cylinders, CSG of those, groups of those —
    glass, mirrors, patterns, a non-casting one) between a few unbounded ones (floor plane, a cube, an open cone): what
    the top-level tree of rt_program.cpp is built for.  Two lights."""
    rng = np.random.RandomState(4200 + seed)
    w = World()
    w.lights.append(Light.point_light((-9., 14., -11.), rt.WHITE))
    w.lights.append(Light.point_light((12., 9., -4.), (0.35, 0.3, 0.3)))
    w.objects.append(Shape.plane().set_material(Material(color=(0.8, 0.8, 0.75), reflective=0.25, pattern=Pattern.checkered((0.9, 0.9, 0.9), (0.2, 0.2, 0.25)))))
    w.objects.append(Shape.cube().set_transform(translate(-7., 1., 9.) * scale(1., 1., 1.)).set_material(Material(color=(0.8, 0.3, 0.2), reflective=0.3)))
    w.objects.append(Shape.cone().set_minimum(-1.).set_maximum(0.).set_transform(translate(7., 2., 10.) * scale(1., 2., 1.)).set_material(Material(color=(0.2, 0.7, 0.3))))

    def material():
        r = rng.rand()
        if r < 0.2:
            return Material(color=(1., 1., 1.), diffuse=0.1, ambient=0.05, reflective=0.7, transparency=0.85, refractive_index=1.2 + 0.5 * rng.rand())
        if r < 0.35:
            return Material(color=tuple(rng.uniform(0.5, 1., 3)), diffuse=0.3, reflective=0.8)
        m = Material(color=tuple(rng.uniform(0.1, 1., 3)), diffuse=0.7, specular=0.4, shininess=float(rng.uniform(10., 200.)))
        if r < 0.5:
            m.pattern = Pattern.striped(tuple(rng.uniform(0., 1., 3)), tuple(rng.uniform(0., 1., 3))).set_transform(scale(0.3, 0.3, 0.3))
        return m

    def place(shape, size):
        x, z = rng.uniform(-9., 9.), rng.uniform(-2., 16.)
        return shape.set_transform(translate(x, size + rng.uniform(0., 2.5), z) * rotate(rng.uniform(0., PI), Axis.Y) *
                                   rotate(rng.uniform(-0.6, 0.6), Axis.X) * scale(size, size * rng.uniform(0.6, 1.4), size))

    def prim():
        if rng.rand() < 0.55:
            return Shape.sphere()
        c = Shape.cylinder().set_minimum(float(rng.uniform(-1.5, -0.2))).set_maximum(float(rng.uniform(0.2, 1.5)))
        return c.set_closed(bool(rng.rand() < 0.7))

    for k in range(n):
        size = float(rng.uniform(0.3, 1.1))
        r = rng.rand()
        if r < 0.6:
            o = place(prim(), size).set_material(material())
        elif r < 0.8:
            a = prim()
            b = prim().set_transform(translate(rng.uniform(-0.6, 0.6), rng.uniform(-0.4, 0.4), rng.uniform(-0.6, 0.6)) * scale(0.8, 0.8, 0.8))
            a.set_material(material())
            b.set_material(material())
            o = place(Shape.csg(("union", "intersection", "difference")[rng.randint(3)], a, b), size)
        else:
            g = Shape.group()
            for _ in range(int(rng.randint(2, 5))):
                g.add_child(prim().set_transform(translate(rng.uniform(-1.2, 1.2), rng.uniform(-0.5, 0.5), rng.uniform(-1.2, 1.2)) * scale(0.5, 0.5, 0.5)))
            o = place(g, size).set_material(material())
        if k == 3:
            o.set_casts_shadow(False)
        w.objects.append(o)
    cam = Camera(hsize, vsize, PI / 3.)
    cam.transform = view_transform((0., 7., -14.), (0., 1., 6.), (0., 1., 0.))
    return w, cam


def instanced_small_meshes(seed=0, hsize=96, vsize=54, n=80, smooth=False, tiny=0):
    """n separately parsed copies of one small OBJ mesh (a 4 x 4 height field: 32 triangles, optionally with vertex
    normals) scattered with their own rotations and scales over a mirror floor, a few of them glass: a world of mesh
    INSTANCES small enough for the brute-force oracle.  Every copy is an object of its own (one ITEM_MESH).  `tiny`: as many
    copies of triangles.obj on top — two named groups of ONE triangle each under the object's group, which the program
    compiler turns into one two-triangle mesh under a top-level tree (rt_program.cpp: only_plain_triangles)."""
    rng = np.random.RandomState(5100 + seed)
    hgt = rng.uniform(-0.3, 0.5, size=(5, 5))
    lines = []
    for j in range(5):
        for i in range(5):
            lines.append("v %r %r %r" % (i / 2. - 1., float(hgt[j, i]), j / 2. - 1.))
    if smooth:
        for j in range(5):
            for i in range(5):
                nv = np.array([float(hgt[j, max(i - 1, 0)] - hgt[j, min(i + 1, 4)]), 1., float(hgt[max(j - 1, 0), i] - hgt[min(j + 1, 4), i])])
                nv /= np.linalg.norm(nv)
                lines.append("vn %r %r %r" % tuple(float(x) for x in nv))
    vid = lambda i, j: j * 5 + i + 1
    for j in range(4):
        for i in range(4):
            for tri in ((vid(i, j), vid(i + 1, j), vid(i + 1, j + 1)), (vid(i, j), vid(i + 1, j + 1), vid(i, j + 1))):
                lines.append("f " + " ".join(("%d//%d" % (v, v)) if smooth else str(v) for v in tri))
    text = "\n".join(lines) + "\n"
    w = World()
    w.lights.append(Light.point_light((-10., 14., -12.), rt.WHITE))
    w.objects.append(Shape.plane().set_material(Material(color=(0.4, 0.4, 0.45), specular=0.2, reflective=0.5)))
    for k in range(n):
        g, _ = rt.obj_to_group(text)
        size = float(rng.uniform(0.5, 1.3))
        g.set_transform(translate(rng.uniform(-12., 12.), rng.uniform(0.4, 3.), rng.uniform(-3., 22.)) * rotate(rng.uniform(0., 2. * PI), Axis.Y) *
                        rotate(rng.uniform(-0.8, 0.8), Axis.Z) * scale(size, size, size))
        if k % 7 == 3:
            g.set_material(Material(color=(1., 1., 1.), diffuse=0.2, reflective=0.4, transparency=0.7, refractive_index=1.3))
        else:
            g.set_material(Material(color=tuple(rng.uniform(0.2, 1., 3)), diffuse=0.7, specular=0.4, reflective=0.25 if k % 3 == 0 else 0.))
        w.objects.append(g)
    for k in range(tiny):
        g, _ = rt.obj_to_group(read_asset("triangles.obj"))
        size = float(rng.uniform(0.8, 2.))
        g.set_transform(translate(rng.uniform(-10., 10.), rng.uniform(0.2, 3.5), rng.uniform(-2., 20.)) * rotate(rng.uniform(0., 2. * PI), Axis.Y) *
                        rotate(rng.uniform(-1.2, 1.2), Axis.X) * scale(size, size, size))
        g.set_material(Material(color=(0.9, 0.9, 0.3), reflective=0.3) if k % 3 else
                       Material(color=(1., 1., 1.), diffuse=0.2, reflective=0.3, transparency=0.8, refractive_index=1.4))
        w.objects.append(g)
    cam = Camera(hsize, vsize, PI / 3.)
    cam.transform = view_transform((0., 8., -15.), (0., 1., 8.), (0., 1., 0.))
    return w, cam


def cover_like_scene(hsize=200, vsize=200):
    """A many-cube/sphere scene in the spirit of render_cover_image (src/main.rs:876): exercises cubes with
    reflective and transparent materials plus a non-shadow-casting plane backdrop."""
    w = World()
    w.lights.append(Light.point_light((50., 100., -50.), rt.WHITE))
    w.lights.append(Light.point_light((-400., 50., -10.), (0.2, 0.2, 0.2)))
    white = Material(color=rt.WHITE, diffuse=0.7, ambient=0.1, specular=0., reflective=0.1)
    blue = Material(color=(0.537, 0.831, 0.914), diffuse=0.7, ambient=0.1, specular=0., reflective=0.1)
    red = Material(color=(0.941, 0.322, 0.388), diffuse=0.7, ambient=0.1, specular=0., reflective=0.1)
    backdrop = Shape.plane().set_material(Material(color=rt.WHITE, ambient=1., diffuse=0., specular=0.))
    backdrop.set_transform(translate(0., 0., 500.) * rotate(PI / 2., Axis.X))
    ball = Shape.sphere().set_material(Material(color=(0.373, 0.404, 0.550), diffuse=0.2, ambient=0., specular=1.,
                                                shininess=200., reflective=0.7, transparency=0.7, refractive_index=1.5))
    ball.set_transform(translate(1., -1., 1.) * scale(0.5, 0.5, 0.5) * scale(3.5, 3.5, 3.5))
    w.objects += [backdrop, ball]
    spots = [(4., 0., 0., 3., white), (8.5, 1.5, -0.5, 3.5, blue), (0., 0., 4., 3.5, red), (4., 0., 4., 2., white),
             (7.5, 0.5, 4., 3., red), (-0.25, 0.25, 8., 3.5, white), (4., 1., 7.5, 3.5, blue), (10., 2., 7.5, 3., red),
             (8., 2., 12., 2., white), (20., 1., 9., 2., white), (-0.5, -5., 0.25, 3.5, blue), (4., -4., 0., 3.5, red)]
    for x, y, z, s, m in spots:
        c = Shape.cube().set_material(m)
        c.set_transform(translate(x, y, z) * translate(1., -1., 1.) * scale(0.5, 0.5, 0.5) * scale(s, s, s))
        w.objects.append(c)
    cam = Camera(hsize, vsize, 0.785)
    cam.transform = view_transform((-6., 6., -10.), (6., 0., 6.), (-0.45, 1., 0.))
    return w, cam


def random_scene(seed, hsize=64, vsize=40, wild=False):
    """A seeded random world for the fuzz parity tests: every primitive kind, nested groups, CSG trees of every
    operation, small flat / smooth triangle fans, patterns with their own transforms, glass, mirrors, objects that
    cast no shadow, truncated and capped cylinders and cones, sheared transforms, one to three lights.  Transforms
    keep |det| well above 1e-5 (F11: the reference panics on a non-invertible matrix).  `wild`: scales from 0.05 to
    20, objects and camera ten times farther out, material parameters outside [0, 1], refractive indices of 0, -1
    and infinity, lights inside the scene."""
    rng = np.random.RandomState(seed + (500_000 if wild else 0))
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    pick = lambda seq: seq[int(rng.randint(len(seq)))]
    colour = lambda: (u(0., 1.), u(0., 1.), u(0., 1.))

    def transform(spread):
        m = translate(u(-spread, spread), u(-0.5, 2.5) if spread > 1. else u(-spread, spread), u(-spread, spread))
        m = m * rotate(u(-PI, PI), pick([Axis.X, Axis.Y, Axis.Z]))
        if rng.rand() < 0.25:
            m = m * rt.shear(u(-0.3, 0.3), 0., 0., u(-0.3, 0.3), 0., 0.)
        lo, hi = (0.05, 20.) if wild and rng.rand() < 0.3 else (0.4, 1.6)
        if rng.rand() < 0.5:
            s = u(lo, hi)
            return m * scale(s, s, s)
        return m * scale(u(lo, hi), u(lo, hi), u(lo, hi))

    def pattern():
        kind = pick([Pattern.striped, Pattern.gradient, Pattern.ring, Pattern.checkered, Pattern.ring_gradient])
        p = kind(colour(), colour())
        if rng.rand() < 0.7:
            s = u(0.2, 1.5)
            p.set_transform(rotate(u(-PI, PI), pick([Axis.X, Axis.Y, Axis.Z])) * scale(s, s, s))
        return p

    def material():
        m = Material(color=colour(), ambient=u(0., 0.3), diffuse=u(0.2, 0.9), specular=u(0., 1.),
                     shininess=pick([10., 50., 200., 300.]), reflective=pick([0., 0., 0.3, 1.]),
                     transparency=pick([0., 0., 0., 0.6, 1.]), refractive_index=pick([1., 1.3, 1.5, 2.417]))
        if wild and rng.rand() < 0.4:
            m = m.with_reflective(pick([1.5, -0.3, 1e-6, 0.999999])).with_transparency(pick([-0.2, 1.7, 1e-6, 0.5]))
            m = m.with_refractive_index(pick([0., -1., float("inf"), 1e-3, 50.]))
        return m.with_pattern(pattern()) if rng.rand() < 0.4 else m

    def conic(shape):
        if rng.rand() < 0.8:
            lo = u(-1.5, 0.)
            shape.set_minimum(lo).set_maximum(lo + u(0.3, 2.))
            shape.set_closed(bool(rng.rand() < 0.6))
        return shape

    def primitive():
        k = int(rng.randint(5))
        s = (Shape.sphere() if k == 0 else Shape.cube() if k == 1 else conic(Shape.cylinder()) if k == 2
             else conic(Shape.cone()) if k == 3 else Shape.glass_sphere())
        return s

    def fan(smooth):
        g = Shape.group()
        n = int(rng.randint(3, 9))
        apex = (u(-0.2, 0.2), u(0.8, 1.4), u(-0.2, 0.2))
        ring = [(math.cos(2. * PI * i / n) * u(0.8, 1.2), u(-0.1, 0.1), math.sin(2. * PI * i / n) * u(0.8, 1.2)) for i in range(n)]
        for i in range(n):
            a, b = ring[i], ring[(i + 1) % n]
            if smooth:
                nrm = lambda p: tuple(np.asarray(p) / np.linalg.norm(p))
                g.add_child(Shape.smooth_triangle(apex, a, b, (0., 1., 0.), nrm(a), nrm(b)))
            else:
                g.add_child(Shape.triangle(apex, a, b))
        return g

    def compound(depth):
        r = rng.rand()
        if depth <= 0 or r < 0.45:
            s = primitive()
            s.set_transform(transform(0.8))
            return s
        if r < 0.7:
            s = Shape.csg(pick(["union", "intersection", "difference"]), compound(depth - 1), compound(depth - 1))
        elif r < 0.9:
            s = Shape.group()
            for _ in range(int(rng.randint(1, 4))):
                s.add_child(compound(depth - 1))
        else:
            s = fan(False)
        s.set_transform(transform(0.5))
        return s

    w = World()
    for _ in range(int(rng.randint(1, 4))):
        w.lights.append(Light.point_light((u(-10., 10.), u(4., 12.), u(-12., -2.)), (u(0.3, 1.), u(0.3, 1.), u(0.3, 1.))))
    if rng.rand() < 0.8:
        w.objects.append(Shape.plane().set_transform(translate(0., -1.2, 0.)).set_material(material()))
    smooth_world = rng.rand() < 0.1  # a world of smooth fans only (next to other shapes the reference panics, F5b)
    for _ in range(int(rng.randint(3, 11))):
        s = fan(True) if smooth_world else compound(2)
        s.set_transform(transform(40. if wild and rng.rand() < 0.3 else 4.))
        s.set_material(material())
        if rng.rand() < 0.15:
            s.set_casts_shadow(False)
        w.objects.append(s)
    if wild:
        w.lights.append(Light.point_light((u(-2., 2.), u(0., 2.), u(-2., 2.)), (1., 1., 1.)))
    cam = Camera(hsize, vsize, u(0.6, 1.4) if not wild else pick([0.05, 0.8, 2.8]))
    a = u(0., 2. * PI)
    r = 11. if not wild else pick([0.5, 11., 110.])
    cam.transform = view_transform((r * math.cos(a), u(1., 7.), r * math.sin(a)), (u(-1., 1.), u(0., 1.5), u(-1., 1.)), (0., 1., 0.))
    return w, cam


def random_mesh_obj(rng):
    """OBJ text of a random height field (two triangles per cell), some faces repeated, up to three polygon faces (the
    parser fans those over the whole vertex list, F3), an optional second named group of loose triangles, optionally
    with vertex normals on every face."""
    nx, nz = int(rng.randint(5, 15)), int(rng.randint(5, 15))
    smooth = rng.rand() < 0.3
    lines = []
    for iz in range(nz):
        for ix in range(nx):
            lines.append(f"v {ix - (nx - 1) / 2.:.4f} {rng.uniform(0., 1.2):.4f} {iz - (nz - 1) / 2.:.4f}")
    n_grid = nx * nz
    n_loose = int(rng.randint(0, 40)) * 3
    for _ in range(n_loose // 3):
        c = rng.uniform(-4., 4., size=3) + np.array([0., 3., 0.])
        for _ in range(3):
            p = c + rng.uniform(-0.8, 0.8, size=3)
            lines.append(f"v {p[0]:.4f} {p[1]:.4f} {p[2]:.4f}")
    if smooth:
        for _ in range(n_grid + n_loose):
            n = rng.normal(size=3) * 0.3 + np.array([0., 1., 0.])
            n /= np.linalg.norm(n)
            lines.append(f"vn {n[0]:.4f} {n[1]:.4f} {n[2]:.4f}")
    ref = (lambda i: f"{i}//{i}") if smooth else (lambda i: f"{i}")
    faces = []
    for iz in range(nz - 1):
        for ix in range(nx - 1):
            a = iz * nx + ix + 1
            faces.append("f " + " ".join(ref(i) for i in (a, a + 1, a + nx + 1)))
            faces.append("f " + " ".join(ref(i) for i in (a, a + nx + 1, a + nx)))
    for k in range(len(faces)):
        if rng.rand() < 0.15:
            faces.append(faces[k])  # a repeated face
    for _ in range(int(rng.randint(0, 4))):  # a polygon face fans over the WHOLE vertex list (F3), each time the same triangles
        a = int(rng.randint(1, n_grid - nx - 2))
        faces.insert(int(rng.randint(len(faces))), "f " + " ".join(ref(i) for i in (a, a + 1, a + nx + 1, a + nx)))
    lines += faces
    if n_loose:
        lines.append("g loose")
        for k in range(n_loose // 3):
            a = n_grid + 3 * k + 1
            lines.append("f " + " ".join(ref(i) for i in (a, a + 1, a + 2)))
    return "\n".join(lines) + "\n", smooth


def random_mesh_scene(seed, hsize=64, vsize=40):
    """A seeded random world around one or two parsed meshes (BVH traversal, repeated triangles, mesh boxes, any-hit
    shadows), optionally next to glass / mirror primitives and a floor."""
    rng = np.random.RandomState(10_000 + seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    pick = lambda seq: seq[int(rng.randint(len(seq)))]
    w = World()
    for _ in range(int(rng.randint(1, 3))):
        w.lights.append(Light.point_light((u(-10., 10.), u(5., 12.), u(-12., -3.)), (u(0.4, 1.), u(0.4, 1.), u(0.4, 1.))))
    alone = False
    for k in range(int(rng.randint(1, 3))):
        text, smooth = random_mesh_obj(rng)
        alone = alone or (smooth and rng.rand() < 0.5)
        g, _ = rt.obj_to_group(text)
        s = u(0.3, 0.9)
        g.set_transform(translate(u(-2., 2.) + 5. * k, u(-0.5, 0.5), u(-2., 2.)) * rotate(u(-PI, PI), Axis.Y) *
                        rotate(u(-0.4, 0.4), Axis.X) * scale(s, s * u(0.6, 1.6), s))
        g.set_material(Material(color=(u(0., 1.), u(0., 1.), u(0., 1.)), diffuse=0.7, specular=u(0., 0.8), shininess=pick([20., 80., 200.]),
                                reflective=pick([0., 0., 0.3]), transparency=pick([0., 0., 0., 0.5]), refractive_index=pick([1., 1.5])))
        if rng.rand() < 0.3:
            g.divide(int(rng.randint(2, 12)))
        w.objects.append(g)
    if not alone:  # smooth meshes next to other shapes make the reference panic (F5b): half of them stay alone
        if rng.rand() < 0.7:
            w.objects.append(Shape.plane().set_transform(translate(0., -1., 0.)).set_material(
                Material(specular=0., reflective=pick([0., 0.2])).with_pattern(Pattern.checkered((0.8, 0.8, 0.8), (0.2, 0.2, 0.2)))))
        for _ in range(int(rng.randint(0, 4))):
            s = pick([Shape.sphere, Shape.cube, Shape.glass_sphere])()
            r = u(0.5, 1.5)
            s.set_transform(translate(u(-5., 8.), u(0.5, 3.), u(-4., 4.)) * scale(r, r, r))
            s.set_material(pick([glass_material(), mirror_material(), Material(color=(u(0., 1.), u(0., 1.), u(0., 1.)))]))
            w.objects.append(s)
    cam = Camera(hsize, vsize, u(0.7, 1.3))
    a = u(0., 2. * PI)
    cam.transform = view_transform((2.5 + 12. * math.cos(a), u(2., 8.), 12. * math.sin(a)), (2.5 + u(-1., 1.), u(0., 1.5), u(-1., 1.)), (0., 1., 0.))
    return w, cam


def random_tie_scene(seed, hsize=64, vsize=40):
    """Coincident surfaces on purpose: several shapes under the SAME transform (translations and power-of-two scales,
    so the arithmetic is exact), CSG of shapes that share faces, a plane through a cube face.  Equal t values must come
    out in the reference's order: stable sorts of the depth-first emission order (src/world.rs:46-55, group.rs:52-65)."""
    rng = np.random.RandomState(900_000 + seed)
    pick = lambda seq: seq[int(rng.randint(len(seq)))]
    colour = lambda: (float(rng.uniform(0., 1.)), float(rng.uniform(0., 1.)), float(rng.uniform(0., 1.)))

    def material():
        return Material(color=colour(), reflective=pick([0., 0.4]), transparency=pick([0., 0., 0.7]), refractive_index=pick([1., 1.5]))

    def shape():
        k = int(rng.randint(4))
        if k == 0:
            return Shape.sphere()
        if k == 1:
            return Shape.cube()
        if k == 2:
            return Shape.cylinder().set_minimum(-1.).set_maximum(1.).set_closed(True)
        return Shape.glass_sphere()

    w = World()
    w.lights.append(Light.point_light((-4., 8., -8.), rt.WHITE))
    if rng.rand() < 0.5:
        w.lights.append(Light.point_light((6., 6., -6.), (0.5, 0.5, 0.5)))
    for cluster in range(int(rng.randint(1, 4))):
        s = pick([0.5, 1., 2.])
        base = translate(float(rng.randint(-3, 4)) * 2., float(rng.randint(0, 2)), float(rng.randint(-2, 3)) * 2.) * scale(s, s, s)
        members = []
        for _ in range(int(rng.randint(2, 5))):
            r = rng.rand()
            if r < 0.6:
                m = shape()
            elif r < 0.85:
                m = Shape.csg(pick(["union", "intersection", "difference"]), shape(), shape())
            else:
                m = Shape.group()
                for _ in range(2):
                    m.add_child(shape())
            m.set_material(material())
            members.append(m)
        if rng.rand() < 0.5:
            g = Shape.group()
            for m in members:
                g.add_child(m)
            g.set_transform(base)
            w.objects.append(g)
        else:
            for m in members:
                m.set_transform(base)
                w.objects.append(m)
        if rng.rand() < 0.5:  # a plane through the unit shapes' bottom face
            w.objects.append(Shape.plane().set_transform(base * translate(0., -1., 0.)).set_material(material()))
    cam = Camera(hsize, vsize, PI / 3.)
    eye = pick([(0., 1., -12.), (0., 12., -3.), (-12., 1., 0.), (7., 5., -9.)])
    cam.transform = view_transform(eye, (0., 0.5, 0.), (0., 1., 0.))
    return w, cam
