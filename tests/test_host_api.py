"""Host mirror (ray-tracer_amd/host) against the reference's unit tests for the construction-side modules, plus the
C-ABI export check and descriptor validation.  CPU only: no compute call is made."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import rt_amd as rt
import scenes
from rt_amd import Axis, Camera, Light, Material, Matrix, Pattern, Shape, World, rotate, scale, shear, translate, view_transform

PI = math.pi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def near(a, b, eps=1e-5):
    return bool(np.all(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float)) < eps))


# ------------------------------------------------------------------ exports
def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared("rt_abi.h") + _declared("rt_host.h")
    assert "rt_render" in names and "rth_world_flatten" in names and len(names) > 50
    missing = [n for n in names if not hasattr(rt.lib, n)]
    assert not missing, missing
    assert sorted(rt.ABI_SYMBOLS) == _declared("rt_abi.h")
    assert sorted(rt.HOST_SYMBOLS) == _declared("rt_host.h")


def test_integration_md_binds_every_declared_symbol_and_builds_every_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    bound = sorted(set(re.findall(r"pub fn (rt_[a-z0-9_]+)\(", text)))
    assert bound == _declared("rt_abi.h")
    # the build.rs recipe must name every translation unit of csrc/ (a missing one is an undefined symbol at load time)
    csrc = os.path.join(ROOT, "ray-tracer_amd", "csrc")
    units = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".cpp")))
    recipe = text[text.index("build.rs"):text.index("## 2.")]
    assert all(("csrc/" + u) in recipe for u in units), units


def test_struct_sizes_match_the_header():
    # rt_camera_desc: 4 int32 + 16 + 3 doubles; rt_render_opts: 4 int32 + pointer; rt_stats: see rt_abi.h
    assert C.sizeof(rt.CameraDesc) == 16 + 19 * 8
    assert C.sizeof(rt.RenderOpts) == 24
    assert C.sizeof(rt.Stats) == 6 * 8 + 8 + 7 * 8 + 8 + 39 * 8


# ------------------------------------------------------------------ matrix.rs / transformation.rs
def test_matrix_inverse_kats():  # matrix.rs:523-605
    m = Matrix([-5., 2., 6., -8., 1., -5., 1., 8., 7., 7., -6., -7., 1., -3., 7., 4.])
    inv = m.inverse()
    assert near(inv.m, [0.21805, 0.45113, 0.24060, -0.04511, -0.80827, -1.45677, -0.44361, 0.52068, -0.07895, -0.22368,
                        -0.05263, 0.19737, -0.52256, -0.81391, -0.30075, 0.30639])
    m = Matrix([8., -5., 9., 2., 7., 5., 6., 1., -6., 0., 9., 6., -3., 0., -9., -4.])
    assert near(m.inverse().m, [-0.15385, -0.15385, -0.28205, -0.53846, -0.07692, 0.12308, 0.02564, 0.03077, 0.35897,
                                0.35897, 0.43590, 0.92308, -0.69231, -0.69231, -0.76923, -1.92308])
    m = Matrix([9., 3., 0., 9., -5., -2., -6., -3., -4., 9., 6., 4., -7., 6., 6., 2.])
    assert near(m.inverse().m, [-0.04074, -0.07778, 0.14444, -0.22222, -0.07778, 0.03333, 0.36667, -0.33333, -0.02901,
                                -0.14630, -0.10926, 0.12963, 0.17778, 0.06667, -0.26667, 0.33333])
    assert Matrix([-4., 2., -2., -3., 9., 6., 2., 6., 0., -5., 1., -5., 0., 0., 0., 0.]).inverse() is None  # :510-521
    a = Matrix([3., -9., 7., 3., 3., -8., 2., -9., -4., 4., 4., 1., -6., 5., -1., 1.])
    b = Matrix([8., 2., 2., 2., 3., -1., 7., 0., 7., 0., 5., 4., 6., -2., 0., 5.])
    assert near(((a * b) * b.inverse()).m, a.m)


def test_inverse_refuses_small_determinant():  # F11: matrix.rs:133-136
    assert scale(0.02, 0.02, 0.02).inverse() is None        # det 8e-6 < 1e-5
    assert scale(0.03, 0.03, 0.03).inverse() is not None
    s = Shape.sphere().set_transform(scale(0.02, 0.02, 0.02))
    w = World()
    w.objects.append(s)
    with pytest.raises(rt.RtError) as e:  # the reference panics at shape.rs:75; here: an error code
        w.flatten()
    assert e.value.code == rt.RT_ERR_UNSUPPORTED and "not invertible" in str(e.value)


def test_view_transform():  # transformation.rs:344-397
    assert near(view_transform((0, 0, 0), (0, 0, -1), (0, 1, 0)).m, Matrix().m)
    assert near(view_transform((0, 0, 0), (0, 0, 1), (0, 1, 0)).m, scale(-1, 1, -1).m)
    assert near(view_transform((0, 0, 8), (0, 0, 0), (0, 1, 0)).m, translate(0, 0, -8).m)
    assert near(view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0)).m,
                [-0.50709, 0.50709, 0.67612, -2.36643, 0.76772, 0.60609, 0.12122, -2.82843, -0.35857, 0.59761, -0.71714, 0., 0., 0., 0., 1.])


def test_rotation_and_shear_entries():  # transformation.rs:28-65
    r = rotate(PI / 2, Axis.X)
    assert near(r.m, [1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, 1])
    sh = shear(1, 2, 3, 4, 5, 6)
    assert near(sh.m, [1, 1, 2, 0, 3, 1, 4, 0, 5, 6, 1, 0, 0, 0, 0, 1])


# ------------------------------------------------------------------ bounds (bound.rs, group.rs, shape.rs)
def test_primitive_bounds():
    assert near(Shape.sphere().bounds_of(), ((-1, -1, -1), (1, 1, 1)))
    mn, mx = Shape.plane().bounds_of()
    assert mn[0] == -np.inf and mn[1] == 0 and mx[2] == np.inf
    mn, mx = Shape.cylinder().bounds_of()
    assert mn[1] == -np.inf and mx[1] == np.inf and mn[0] == -1
    mn, mx = Shape.cone().set_minimum(-5).set_maximum(3).bounds_of()  # cone.rs:257-272
    assert near(mn, (-5, -5, -5)) and near(mx, (5, 3, 5))
    mn, mx = Shape.triangle((0, 1, 0), (-1, 0, 0), (1, 0, 0)).bounds_of()  # F4: triangle.rs:93-95
    assert np.all(mn == np.inf) and np.all(mx == -np.inf)


def test_group_bounds():  # group.rs:244-262, shape.rs:748-759
    s = Shape.sphere().set_transform(translate(2, 5, -3) * scale(2, 2, 2))
    c = Shape.cylinder().set_minimum(-2).set_maximum(2).set_transform(translate(-4, -1, 4) * scale(0.5, 1, 0.5))
    g = Shape.group().add_child(s).add_child(c)
    mn, mx = g.bounds_of()
    assert near(mn, (-4.5, -3, -5)) and near(mx, (4, 7, 4.5))
    s = Shape.sphere().set_transform(translate(1, -3, 5) * scale(0.5, 2, 4))
    mn, mx = s.parent_space_bounds_of()
    assert near(mn, (0.5, -5, 1)) and near(mx, (1.5, -1, 9))


def test_triangle_group_box_is_infinite():  # F4
    t = Shape.triangle((0, 1, 0), (-1, 0, 0), (1, 0, 0))
    g = Shape.group().add_child(t)
    mn, mx = g.bounds_of()
    assert np.all(mn == -np.inf) and np.all(mx == np.inf)


def test_divide():  # group.rs:345-373
    s1 = Shape.sphere().set_transform(translate(-2, -2, 0))
    s2 = Shape.sphere().set_transform(translate(-2, 2, 0))
    s3 = Shape.sphere().set_transform(scale(4, 4, 4))
    g = Shape.group().add_child(s1).add_child(s2).add_child(s3)
    g.divide(1)
    kids = g.get_shapes()
    assert kids[0].get_id() == s3.get_id()
    sub = kids[1].get_shapes()
    assert len(sub) == 2
    assert sub[0].get_shapes()[0].get_id() == s1.get_id() and sub[1].get_shapes()[0].get_id() == s2.get_id()


def test_add_child_stores_a_clone_and_sets_parent():  # group.rs:45-50
    g = Shape.group()
    s = Shape.sphere()
    g.add_child(s)
    assert s.get_parent() == g.get_id()
    s.set_transform(translate(1, 0, 0))           # after the fact: the group's copy keeps the old transform
    w = World()
    w.objects.append(g)
    d = w.flatten().contents
    assert d.n_nodes == 2 and d.node_xform[1] == 0  # identity


def test_csg_sets_parents_and_includes():  # csg.rs:150-173, shape.rs:422-435
    a, b = Shape.sphere(), Shape.cube()
    c = Shape.csg("union", a, b)
    assert a.get_parent() == c.get_id() and b.get_parent() == c.get_id()
    assert c.includes(a) and c.includes(b) and not c.includes(Shape.sphere())


def test_wrong_variant_accessors_fail():  # shape.rs:247-292 panics -> error codes
    with pytest.raises(rt.RtError):
        Shape.sphere().set_minimum(1.)
    with pytest.raises(rt.RtError):
        Shape.sphere().add_child(Shape.sphere())


# ------------------------------------------------------------------ obj_file.rs
def test_obj_ignored_lines():  # obj_file.rs:180-192
    text = ("There was a young lady named Bright\nwho traveled much faster than light.\nShe set out one day\n"
            "in a relative way,\nand came back the previous night.\n")
    _, counts = rt.obj_to_group(text)
    assert counts["lines_ignored"] == 5


def test_obj_faces_and_fan():  # obj_file.rs:214-252
    g, counts = rt.obj_to_group("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3\nf 1 3 4\n")
    tris = g.get_shapes()[0].get_shapes()
    assert counts["vertices"] == 5 and len(tris) == 2
    assert near(tris[0].triangle_data()[:3], [(-1, 1, 0), (-1, 0, 0), (1, 0, 0)])
    assert near(tris[1].triangle_data()[:3], [(-1, 1, 0), (1, 0, 0), (1, 1, 0)])
    g, _ = rt.obj_to_group("v -1 1 0\nv -1 0 0\nv 1 0 0\nv 1 1 0\nv 0 2 0\nf 1 2 3 4 5\n")
    tris = g.get_shapes()[0].get_shapes()
    assert len(tris) == 3
    assert near(tris[2].triangle_data()[:3], [(-1, 1, 0), (1, 1, 0), (0, 2, 0)])


def test_obj_fan_bug_uses_the_whole_vertex_list():  # F3: obj_file.rs:143-151
    # a quad that names only 4 of 6 vertices still fans over all 6
    g, _ = rt.obj_to_group("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 5 5 5\nv 6 6 6\nf 1 2 3 4\n")
    tris = g.get_shapes()[0].get_shapes()
    assert len(tris) == 4
    assert near(tris[3].triangle_data()[:3], [(0, 0, 0), (5, 5, 5), (6, 6, 6)])


def test_obj_groups_and_assets():  # obj_file.rs:254-284 + asset shapes of SURVEY F2/F3
    g, counts = rt.obj_to_group(scenes.read_asset("triangles.obj"))
    assert counts["groups"] == 3 and g.child_count() == 3
    _, counts = rt.obj_to_group(scenes.read_asset("teapot.obj"))
    assert counts["triangles"] == 6320 and counts["normals"] == 1 and counts["groups"] == 1
    _, counts = rt.obj_to_group(scenes.read_asset("teapot-low.obj"))
    assert counts["triangles"] == 16 + 112 * 135 and counts["groups"] == 2


def test_obj_normals():  # obj_file.rs:286-333
    g, counts = rt.obj_to_group("v 0 1 0\nv -1 0 0\nv 1 0 0\nvn -1 0 0\nvn 1 0 0\nvn 0 1 0\nf 1//3 2//1 3//2\nf 1/0/3 2/102/1 3/14/2\n")
    assert counts["normals"] == 4
    for t in g.get_shapes()[0].get_shapes():
        d = t.triangle_data()
        assert t.kind() == rt.RT_SMOOTH_TRIANGLE
        assert near(d[:3], [(0, 1, 0), (-1, 0, 0), (1, 0, 0)]) and near(d[6:], [(0, 1, 0), (-1, 0, 0), (1, 0, 0)])


# ------------------------------------------------------------------ canvas.rs
def test_canvas_to_ppm():  # canvas.rs:136-200
    rgb = np.zeros((3, 5, 3))
    rgb[0, 0] = (1.5, 0, 0)
    rgb[1, 2] = (0, 0.5, 0)
    rgb[2, 4] = (-0.5, 0, 1)
    ppm = rt.Image(rgb, None, None, None).to_ppm()
    assert ppm.splitlines()[:3] == ["P3", "5 3", "255"]
    assert ppm.splitlines()[3:6] == ["255 0 0 0 0 0 0 0 0 0 0 0 0 0 0", "0 0 0 0 0 0 0 128 0 0 0 0 0 0 0", "0 0 0 0 0 0 0 0 0 0 0 0 0 0 255"]
    rgb = np.zeros((2, 10, 3))
    rgb[:, :] = (1.0, 0.8, 0.6)
    ppm = rt.Image(rgb, None, None, None).to_ppm()
    assert ppm.splitlines()[3:7] == ["255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204",
                                     "153 255 204 153 255 204 153 255 204 153", ] * 2 or \
        ppm.splitlines()[3] == "255 204 153 255 204 153 255 204 153 255 204 153 255 204 153 255 204"
    assert ppm.splitlines()[4] == "153 255 204 153 255 204 153 255 204 153 255 204 153"
    assert ppm.endswith("\n")
    q = rt.Image(np.array([[[1.5, 0.5, -0.5]]]), None, None, None).quantised()
    assert q.tolist() == [[[255, 128, 0]]]


# ------------------------------------------------------------------ flattening
def test_flatten_teapot_layout():
    w, _ = scenes.teapot_scene(64, 36)
    d = w.flatten().contents
    assert d.n_nodes == 1 + 2 + 6320 + 2          # plane, main group + obj group, triangles, two spheres
    kinds = np.ctypeslib.as_array(d.node_kind, shape=(d.n_nodes,))
    assert kinds[0] == rt.RT_PLANE and kinds[1] == rt.RT_GROUP and kinds[2] == rt.RT_GROUP and kinds[3] == rt.RT_TRIANGLE
    ends = np.ctypeslib.as_array(d.node_subtree_end, shape=(d.n_nodes,))
    assert ends[1] == 3 + 6320 and ends[2] == 3 + 6320
    box = np.ctypeslib.as_array(d.group_box, shape=(d.n_groups, 6))
    assert np.all(box[:, :3] == -np.inf) and np.all(box[:, 3:] == np.inf)   # F4
    mats = np.ctypeslib.as_array(d.node_material, shape=(d.n_nodes,))
    assert len(set(mats[3:3 + 6320])) == 1                                   # F7: the main group's material for every triangle
    assert d.n_xforms <= 8 and d.n_ancestors == 2


def test_flatten_equality_classes():
    cyl = Shape.cylinder()                      # unbounded: not equal to itself (F14)
    s = Shape.sphere()
    w = World()
    w.objects += [cyl, s, s.clone()]            # the same shape twice: equal to each other
    d = w.flatten().contents
    assert list(np.ctypeslib.as_array(d.node_eq_class, shape=(3,))) == [-1, 1, 1]


def test_shard_pixel_counts_cover_the_frame():
    cam = Camera(1920, 1080, PI / 3)
    cd = cam.desc()
    tiles = ((1919 + 7) // 8) * ((1079 + 7) // 8)
    for n in (1, 2, 3, 4, 8):
        counts = [rt.lib.rt_shard_pixel_count(C.byref(cd), r, n) for r in range(n)]
        assert sum(counts) == tiles * 64 and max(counts) - min(counts) <= 64
    assert rt.lib.rt_shard_pixel_count(C.byref(cd), 2, 2) < 0


def test_no_gpu_means_a_clear_error_not_a_fallback():
    if rt.device_count() > 0:
        pytest.skip("a GPU is present")
    w = World.default()
    with pytest.raises(rt.RtError) as e:
        w.scene()
    assert "no HIP device" in str(e.value) or "no CPU path" in str(e.value)


# ---------------------------------------------------------------------------------------------------------------
# OBJ -> flattened tables without a Shape per triangle (rth_obj_file_to_group)

_TABLE_SIZES = {
    "node_kind": ("n_nodes", 1), "node_object_id": ("n_nodes", 1), "node_parent": ("n_nodes", 1), "node_subtree_end": ("n_nodes", 1),
    "node_xform": ("n_nodes", 1), "node_ancestor": ("n_nodes", 1), "node_material": ("n_nodes", 1), "node_casts_shadow": ("n_nodes", 1),
    "node_eq_class": ("n_nodes", 1), "node_param": ("n_nodes", 1), "xform_inverse": ("n_xforms", 16), "xform_inverse_transpose": ("n_xforms", 16),
    "ancestor_xform": ("n_ancestors", 1), "ancestor_parent": ("n_ancestors", 1), "group_box": ("n_groups", 6), "csg_op": ("n_csg", 1),
    "conic_minimum": ("n_conics", 1), "conic_maximum": ("n_conics", 1), "conic_closed": ("n_conics", 1), "tri_p1": ("n_triangles", 3),
    "tri_e1": ("n_triangles", 3), "tri_e2": ("n_triangles", 3), "tri_normal": ("n_triangles", 3), "tri_smooth": ("n_triangles", 1),
    "smooth_n1": ("n_smooth", 3), "smooth_n2": ("n_smooth", 3), "smooth_n3": ("n_smooth", 3), "material_params": ("n_materials", 10),
    "material_pattern": ("n_materials", 1), "pattern_kind": ("n_patterns", 1), "pattern_a": ("n_patterns", 3), "pattern_b": ("n_patterns", 3),
    "pattern_inverse": ("n_patterns", 16), "light_position": ("n_lights", 3), "light_intensity": ("n_lights", 3)}


def _tables(world):
    d = world.flatten().contents
    out = {}
    for name, (count, width) in _TABLE_SIZES.items():
        n = getattr(d, count) * width
        out[name] = np.ctypeslib.as_array(getattr(d, name), shape=(n,)).copy() if n else np.zeros(0)
    return out


def _same_tables(a, b, what):
    """Bit for bit, except that object ids (drawn from one global counter) may differ by renaming: the mapping between
    the two id columns must be one-to-one."""
    for name in _TABLE_SIZES:
        x, y = a[name], b[name]
        assert x.shape == y.shape, (what, name, x.shape, y.shape)
        if name == "node_object_id":
            fwd, back = {}, {}
            for p, q in zip(x.tolist(), y.tolist()):
                assert fwd.setdefault(p, q) == q and back.setdefault(q, p) == p, (what, "object ids are not a renaming")
        else:
            assert x.tobytes() == y.tobytes(), (what, name)


def _obj_worlds(text, packed):
    """The same constructions over a mesh parsed either way; each returns a World."""
    from rt_amd import Light, Material, Shape, World, rotate, scale, translate
    def mesh():
        g, counts = rt.obj_to_group(text, packed=packed)
        return g, counts
    worlds = []
    # 1: the mesh alone, transformed and with a material set on the main group
    g, counts = mesh()
    g.set_transform(translate(0.5, 0., -1.) * rotate(0.3, rt.Axis.Y) * scale(0.5, 0.5, 0.5))
    g.set_material(Material(color=(0.2, 0.7, 0.4), reflective=0.3))
    w = World(); w.lights.append(Light.point_light((-5., 8., -6.), rt.WHITE)); w.objects.append(g)
    worlds.append(("alone", w))
    # 2: between other shapes, and the SAME group twice in one world (equal ids: Shape::eq classes must be shared)
    g, _ = mesh()
    g.set_material(Material(color=(0.9, 0.1, 0.1), transparency=0.5, refractive_index=1.3))
    w = World(); w.lights.append(Light.point_light((-5., 8., -6.), rt.WHITE))
    w.objects += [Shape.sphere().set_transform(translate(3., 0., 0.)), g, Shape.plane(), g]
    worlds.append(("twice", w))
    # 3: as a CSG operand and inside another group
    g, _ = mesh()
    c = Shape.csg("difference", g, Shape.cube())
    outer = Shape.group()
    g2, _ = mesh()
    g2.set_casts_shadow(False)
    outer.add_child(g2)
    outer.set_transform(scale(2., 2., 2.))
    w = World(); w.lights.append(Light.point_light((-5., 8., -6.), rt.WHITE)); w.objects += [c, outer]
    worlds.append(("csg+nested", w))
    # 4: divided (needs the children as shapes: the packed form is materialised)
    g, _ = mesh()
    g.divide(4)
    w = World(); w.lights.append(Light.point_light((-5., 8., -6.), rt.WHITE)); w.objects.append(g)
    worlds.append(("divided", w))
    return worlds, counts


@pytest.mark.parametrize("asset", ["triangles.obj", "teapot-low.obj", "teapot.obj", "smooth", "random0", "random1", "random2", "quirks"])
def test_packed_obj_groups_flatten_to_the_same_tables_as_shape_trees(asset):
    import scenes
    if asset == "smooth":
        text = scenes.smooth_teapot_obj()
    elif asset.startswith("random"):
        text, _ = scenes.random_mesh_obj(np.random.RandomState(100 + int(asset[-1])))
    elif asset == "quirks":  # several groups, an empty one, polygon faces (F3), ignored lines, a bad vertex, a vn-less slash form
        text = "\n".join(["# comment", "v 0 0 0", "v 1 0 0", "v 1 1 0", "v x y z", "v 0 1 0", "v 0.5 2 0", "vn 0 0 1", "vn 0 1 0", "vn 1 0 0", "vn 0 0 -1", "vn 0 -1 0", "vn -1 0 0",
                          "g first", "f 1 2 3", "f 1/5 2/6 3/7", "g empty", "g second", "f 1 2 3 4 5", "f 1//1 2//2 3//3",
                          "f 1//1 2//2 3//3 4//1", "junk line", "f 2 3 4"]) + "\n"
    else:
        text = scenes.read_asset(asset)
    shapes, counts_s = _obj_worlds(text, packed=False)
    packed, counts_p = _obj_worlds(text, packed=True)
    assert counts_s == counts_p
    for (name, ws), (_, wp) in zip(shapes, packed):
        _same_tables(_tables(ws), _tables(wp), (asset, name))


def test_packed_obj_group_behaves_like_a_group_of_shapes():
    """Everything the mirror lets a caller do with a parsed group works on the packed form (it is unpacked on demand)."""
    import scenes
    text = scenes.read_asset("triangles.obj")
    a, _ = rt.obj_to_group(text, packed=False)
    b, _ = rt.obj_to_group(text, packed=True)
    assert np.array_equal(np.array(a.bounds_of()), np.array(b.bounds_of()), equal_nan=True)
    b2, _ = rt.obj_to_group(text, packed=True)
    first = lambda g: next(g.child(k).child(0) for k in range(g.child_count()) if g.child(k).child_count())
    assert b2.includes(first(b2)) and not b2.includes(first(a))
    assert b.child_count() == a.child_count()
    for k in range(a.child_count()):
        ga, gb = a.child(k), b.child(k)
        assert ga.child_count() == gb.child_count()
        for j in range(ga.child_count()):
            assert np.array_equal(ga.child(j).triangle_data(), gb.child(j).triangle_data())
