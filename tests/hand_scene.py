"""rt_scene_desc tables written BY HAND (numpy literals), never produced by ray-tracer_amd/host/flatten.cpp.

Every GPU-vs-oracle parity test feeds both sides the flattener's output, so a flattener bug would be invisible to them.
The two scenes here are spelled out table by table from the reference's definitions (file:line below) and serve two
purposes (tests/test_hand_written_desc.py): the flattener's output for the same scene, built through the host mirror,
must expand to the same per-node content (`canonical`), and oracle and device must agree on the hand tables too.

TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import math

import numpy as np

import rt_amd as rt

INF = math.inf
IDENT = [1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.]


def translation(x, y, z):  # transformation.rs:10-17
    return [1., 0., 0., x, 0., 1., 0., y, 0., 0., 1., z, 0., 0., 0., 1.]


def scaling(x, y, z):  # transformation.rs:19-26
    return [x, 0., 0., 0., 0., y, 0., 0., 0., 0., z, 0., 0., 0., 0., 1.]


def transposed(m):
    return [m[4 * c + r] for r in range(4) for c in range(4)]


DEFAULT_MATERIAL = [1., 1., 1., 0.1, 0.9, 0.9, 200., 0., 0., 1.]  # material.rs:23-37: color, ka, kd, ks, shininess, refl, transp, ior


class HandDesc:
    """Keeps the numpy arrays alive and exposes a ctypes rt_scene_desc over them."""

    def __init__(self, **t):
        self.arrays = {}
        d = rt.SceneDesc()
        d.abi_version = 1

        def put(name, values, dtype):
            a = np.ascontiguousarray(np.asarray(values, dtype=dtype).reshape(-1))
            if a.size == 0:
                a = np.zeros(1, dtype=dtype)   # never a null pointer; the count says it is empty
            self.arrays[name] = a
            ptr = {np.float64: rt.c_dp, np.int32: rt.c_ip, np.uint8: rt.c_bp}[dtype]
            setattr(d, name, a.ctypes.data_as(ptr))

        i32, f64, u8 = np.int32, np.float64, np.uint8
        d.n_nodes = len(t["node_kind"])
        for k in ("node_kind", "node_object_id", "node_parent", "node_subtree_end", "node_xform", "node_ancestor", "node_material",
                  "node_eq_class", "node_param"):
            put(k, t[k], i32)
        put("node_casts_shadow", t["node_casts_shadow"], u8)
        d.n_xforms = len(t["xform_inverse"])
        put("xform_inverse", t["xform_inverse"], f64)
        put("xform_inverse_transpose", [transposed(m) for m in t["xform_inverse"]], f64)
        d.n_ancestors = len(t["ancestor_xform"])
        put("ancestor_xform", t["ancestor_xform"], i32)
        put("ancestor_parent", t["ancestor_parent"], i32)
        d.n_groups = len(t["group_box"])
        put("group_box", t["group_box"], f64)
        d.n_csg = len(t["csg_op"])
        put("csg_op", t["csg_op"], i32)
        d.n_conics = len(t["conic_minimum"])
        put("conic_minimum", t["conic_minimum"], f64)
        put("conic_maximum", t["conic_maximum"], f64)
        put("conic_closed", t["conic_closed"], u8)
        d.n_triangles = len(t["tri_smooth"])
        for k in ("tri_p1", "tri_e1", "tri_e2", "tri_normal"):
            put(k, t[k], f64)
        put("tri_smooth", t["tri_smooth"], i32)
        d.n_smooth = len(t["smooth_n1"])
        for k in ("smooth_n1", "smooth_n2", "smooth_n3"):
            put(k, t[k], f64)
        d.n_materials = len(t["material_params"])
        put("material_params", t["material_params"], f64)
        put("material_pattern", t["material_pattern"], i32)
        d.n_patterns = len(t["pattern_kind"])
        put("pattern_kind", t["pattern_kind"], i32)
        put("pattern_a", t["pattern_a"], f64)
        put("pattern_b", t["pattern_b"], f64)
        put("pattern_inverse", t["pattern_inverse"], f64)
        d.n_lights = len(t["light_position"])
        put("light_position", t["light_position"], f64)
        put("light_intensity", t["light_intensity"], f64)
        self.desc = d

    def pointer(self):
        return C.pointer(self.desc)


EMPTY = dict(ancestor_xform=[], ancestor_parent=[], group_box=[], csg_op=[], conic_minimum=[], conic_maximum=[], conic_closed=[],
             tri_p1=[], tri_e1=[], tri_e2=[], tri_normal=[], tri_smooth=[], smooth_n1=[], smooth_n2=[], smooth_n3=[],
             pattern_kind=[], pattern_a=[], pattern_b=[], pattern_inverse=[])


def default_world():
    """impl Default for World, src/world.rs:20-39: light (-10,10,-10) white; s1 with color (0.8,1,0.6), diffuse 0.7,
    specular 0.2; s2 scaled by 0.5 (inverse: scaling(2,2,2)).  Object ids are arbitrary distinct positive numbers."""
    t = dict(EMPTY)
    t.update(
        node_kind=[rt.RT_SPHERE, rt.RT_SPHERE], node_object_id=[11, 12], node_parent=[-1, -1], node_subtree_end=[1, 2],
        node_xform=[0, 1], node_ancestor=[-1, -1], node_material=[0, 1], node_casts_shadow=[1, 1], node_eq_class=[0, 1],
        node_param=[-1, -1],
        xform_inverse=[IDENT, scaling(2., 2., 2.)],
        material_params=[[0.8, 1., 0.6, 0.1, 0.7, 0.2, 200., 0., 0., 1.], DEFAULT_MATERIAL], material_pattern=[-1, -1],
        light_position=[[-10., 10., -10.]], light_intensity=[[1., 1., 1.]])
    return HandDesc(**t)


def build_default_world():
    return rt.World.default()


def composite_world():
    """Nine nodes covering every table:
      0 plane, translate(0,-1,0), reflective 0.3 + checkered pattern scaled by 2
      1 group G translate(1,0,0), material red        (children take the ROOT ancestor's material, shape.rs:163-184)
      2   sphere S scale(2,2,2), own material green   (ignored: F7)
      3   cylinder C min 0 max 1 closed, translate(0,0,2), casts no shadow (the leaf's own flag, shape.rs:209-222)
      4 csg D difference, translate(-3,0,0)
      5   cube
      6   sphere translate(0.5,0.5,-0.5)
      7 group T translate(-1,2,-2): its box is (-inf, +inf) because triangles have empty bounds (F4: triangle.rs:93-95, bound.rs:35-47)
      8   triangle (0,1,0) (-1,0,0) (1,0,0)
      9   smooth triangle, same points, normals (0,1,0) (-1,0,0) (1,0,0)
    Group G's box: S -> +-2 on every axis; C -> (-1,0,-1)..(1,1,1) moved by (0,0,2) = (-1,0,1)..(1,1,3); union."""
    red = [1., 0.2, 0.2, 0.1, 0.9, 0.9, 200., 0., 0., 1.]
    floor = [1., 1., 1., 0.1, 0.9, 0.9, 200., 0.3, 0., 1.]
    glass = [0.9, 1., 0.9, 0.1, 0.3, 0.9, 200., 0.4, 0.7, 1.5]
    e1, e2 = [-1., -1., 0.], [1., -1., 0.]          # p2 - p1, p3 - p1 (triangle.rs:38-56)
    normal = [0., 0., -1.]                          # normalize(e2 x e1) = (0, 0, -2) / 2
    t = dict(
        node_kind=[rt.RT_PLANE, rt.RT_GROUP, rt.RT_SPHERE, rt.RT_CYLINDER, rt.RT_CSG, rt.RT_CUBE, rt.RT_SPHERE, rt.RT_GROUP,
                   rt.RT_TRIANGLE, rt.RT_SMOOTH_TRIANGLE],
        node_object_id=[101, 102, 103, 104, 105, 106, 107, 108, 109, 110],
        node_parent=[-1, -1, 1, 1, -1, 4, 4, -1, 7, 7],
        node_subtree_end=[1, 4, 3, 4, 7, 6, 7, 10, 9, 10],
        node_xform=[1, 2, 3, 4, 5, 0, 6, 7, 0, 0],
        node_ancestor=[-1, -1, 0, 0, -1, 1, 1, -1, 2, 2],
        node_material=[0, -1, 1, 1, -1, 2, 2, -1, 3, 3],
        node_casts_shadow=[1, 1, 1, 0, 1, 1, 1, 1, 1, 1],
        node_eq_class=[0, -1, 2, 3, -1, 5, 6, -1, 8, 9],
        node_param=[-1, 0, -1, 0, 0, -1, -1, 1, 0, 1],
        xform_inverse=[IDENT, translation(0., 1., 0.), translation(-1., 0., 0.), scaling(0.5, 0.5, 0.5), translation(0., 0., -2.),
                       translation(3., 0., 0.), translation(-0.5, -0.5, 0.5), translation(1., -2., 2.)],
        ancestor_xform=[2, 5, 7], ancestor_parent=[-1, -1, -1],
        group_box=[[-2., -2., -2., 2., 2., 3.], [-INF, -INF, -INF, INF, INF, INF]],
        csg_op=[2],  # RT_CSG_DIFFERENCE
        conic_minimum=[0.], conic_maximum=[1.], conic_closed=[1],
        tri_p1=[[0., 1., 0.], [0., 1., 0.]], tri_e1=[e1, e1], tri_e2=[e2, e2], tri_normal=[normal, normal], tri_smooth=[-1, 0],
        smooth_n1=[[0., 1., 0.]], smooth_n2=[[-1., 0., 0.]], smooth_n3=[[1., 0., 0.]],
        material_params=[floor, red, DEFAULT_MATERIAL, glass], material_pattern=[0, -1, -1, -1],
        pattern_kind=[rt.RT_PAT_CHECKERED], pattern_a=[[1., 1., 1.]], pattern_b=[[0., 0., 0.]],
        pattern_inverse=[scaling(0.5, 0.5, 0.5)],
        light_position=[[-10., 10., -10.], [5., 8., -3.]], light_intensity=[[1., 1., 1.], [0.4, 0.4, 0.6]])
    return HandDesc(**t)


def build_composite_world():
    """The same scene through the host mirror of the crate's API (what the flattener then sees)."""
    from rt_amd import Light, Material, Pattern, Shape, World, scale, translate
    w = World()
    w.lights += [Light.point_light((-10., 10., -10.), rt.WHITE), Light.point_light((5., 8., -3.), (0.4, 0.4, 0.6))]
    pat = Pattern.checkered(rt.WHITE, rt.BLACK).set_transform(scale(2., 2., 2.))
    plane = Shape.plane().set_transform(translate(0., -1., 0.)).set_material(Material(reflective=0.3).with_pattern(pat))
    g = Shape.group().set_transform(translate(1., 0., 0.)).set_material(Material(color=(1., 0.2, 0.2)))
    s = Shape.sphere().set_transform(scale(2., 2., 2.)).set_material(Material(color=(0.2, 1., 0.2)))
    c = Shape.cylinder().set_minimum(0.).set_maximum(1.).set_closed(True).set_transform(translate(0., 0., 2.)).set_casts_shadow(False)
    g.add_child(s)
    g.add_child(c)
    cube = Shape.cube()
    sph = Shape.sphere().set_transform(translate(0.5, 0.5, -0.5))
    d = Shape.csg("difference", cube, sph).set_transform(translate(-3., 0., 0.))
    tg = Shape.group().set_transform(translate(-1., 2., -2.)).set_material(Material(color=(0.9, 1., 0.9), diffuse=0.3, reflective=0.4, transparency=0.7, refractive_index=1.5))
    tg.add_child(Shape.triangle((0., 1., 0.), (-1., 0., 0.), (1., 0., 0.)))
    tg.add_child(Shape.smooth_triangle((0., 1., 0.), (-1., 0., 0.), (1., 0., 0.), (0., 1., 0.), (-1., 0., 0.), (1., 0., 0.)))
    w.objects += [plane, g, d, tg]
    return w


def _triangle_record(p1, p2, p3):
    """Triangle::new in plain Python floats (src/triangle.rs:38-56): e1 = p2 - p1, e2 = p3 - p1, normal = normalize(e2 x e1)
    with Tuple::cross (tuple.rs:49-52) and Tuple::normalize (tuple.rs:32-43: one sqrt, then a division per component).
    Independent of the host mirror: nothing here calls into the library."""
    e1 = [p2[k] - p1[k] for k in range(3)]
    e2 = [p3[k] - p1[k] for k in range(3)]
    c = [e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]]
    m = math.sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + 0. * 0.)
    return e1, e2, [c[0] / m, c[1] / m, c[2] / m]


# an octahedron: six vertices (1-based, as an OBJ file numbers them), eight faces; two faces carry vertex normals
OCTA_V = [None, [1., 0., 0.], [-1., 0., 0.], [0., 1., 0.], [0., -1., 0.], [0., 0., 1.], [0., 0., -1.]]
OCTA_F = [(1, 3, 5, False), (3, 2, 5, False), (2, 4, 5, True), (4, 1, 5, False), (3, 1, 6, False), (2, 3, 6, False), (4, 2, 6, True), (1, 4, 6, False)]
OCTA_OBJ = "".join("v %r %r %r\n" % tuple(v) for v in OCTA_V[1:]) + "".join("vn %r %r %r\n" % tuple(v) for v in OCTA_V[1:]) + \
    "".join(("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c)) if smooth else ("f %d %d %d\n" % (a, b, c)) for a, b, c, smooth in OCTA_F)


def mesh_world():
    """A parsed mesh under a two-level TRANSFORMED ancestor chain, tables typed in from the reference's definitions:
      0 plane translate(0,-1.5,0), reflective 0.2
      1 group OUTER translate(0.5,0.25,1), glass-like material (the ROOT ancestor's material is every triangle's, F7)
      2   group WRAP scale(2,2,2)            = obj_to_group's main_group (src/obj_file.rs:164-172)
      3     group DEFAULT (identity)         = the parser's default group (src/obj_file.rs:33-35)
      4..11   the octahedron's eight faces in file order; faces 2 and 6 are SmoothTriangles (`f a//a b//b c//c`,
              src/obj_file.rs:75-100) whose vertex normals are the vertex positions
    Every group box is (-inf, +inf): a triangle's box is EMPTY (triangle.rs:93-95), Bound::transform of a box with an infinite
    corner yields NaN points (0 * inf in Matrix * Tuple) that add_point ignores, so the transformed box is EMPTY again, and
    add_box(EMPTY) turns the group box into (-inf, +inf) (bound.rs:35-47) — at every level.
    world_to_object / normal_to_world walk DEFAULT -> WRAP -> OUTER (shape.rs:394-420): three ancestor entries, the
    identity one included."""
    floor = [1., 1., 1., 0.1, 0.9, 0.9, 200., 0.2, 0., 1.]
    glass = [0.8, 0.9, 1., 0.1, 0.4, 0.9, 200., 0.3, 0.6, 1.4]
    p1s, e1s, e2s, ns, smooth, sn = [], [], [], [], [], [[], [], []]
    for a, b, c, is_smooth in OCTA_F:
        e1, e2, n = _triangle_record(OCTA_V[a], OCTA_V[b], OCTA_V[c])
        p1s.append(OCTA_V[a]); e1s.append(e1); e2s.append(e2); ns.append(n)
        if is_smooth:
            smooth.append(len(sn[0]))
            for k, v in enumerate((a, b, c)):
                sn[k].append(OCTA_V[v])
        else:
            smooth.append(-1)
    box = [-INF, -INF, -INF, INF, INF, INF]
    kinds = [rt.RT_SMOOTH_TRIANGLE if f[3] else rt.RT_TRIANGLE for f in OCTA_F]
    t = dict(EMPTY)
    t.update(
        node_kind=[rt.RT_PLANE, rt.RT_GROUP, rt.RT_GROUP, rt.RT_GROUP] + kinds,
        node_object_id=[201, 202, 203, 204] + list(range(205, 213)),
        node_parent=[-1, -1, 1, 2] + [3] * 8,
        node_subtree_end=[1, 12, 12, 12] + list(range(5, 13)),
        node_xform=[1, 2, 3, 0] + [0] * 8,
        node_ancestor=[-1, -1, -1, -1] + [0] * 8,
        node_material=[0, -1, -1, -1] + [1] * 8,
        node_casts_shadow=[1] * 12,
        node_eq_class=[0, -1, -1, -1] + list(range(1, 9)),
        node_param=[-1, 0, 1, 2] + list(range(8)),
        xform_inverse=[IDENT, translation(0., 1.5, 0.), translation(-0.5, -0.25, -1.), scaling(0.5, 0.5, 0.5)],
        ancestor_xform=[0, 3, 2], ancestor_parent=[1, 2, -1],      # DEFAULT -> WRAP -> OUTER
        group_box=[box, box, box],
        tri_p1=p1s, tri_e1=e1s, tri_e2=e2s, tri_normal=ns, tri_smooth=smooth,
        smooth_n1=sn[0], smooth_n2=sn[1], smooth_n3=sn[2],
        material_params=[floor, glass], material_pattern=[-1, -1],
        light_position=[[-8., 9., -10.], [6., 4., -6.]], light_intensity=[[1., 1., 1.], [0.3, 0.3, 0.4]])
    return HandDesc(**t)


def build_mesh_world():
    """The same scene through parse_obj_file + obj_to_group (by default the packed OBJ path, rth_obj_file_to_group)."""
    from rt_amd import Light, Material, Shape, World, scale, translate
    w = World()
    w.lights += [Light.point_light((-8., 9., -10.), rt.WHITE), Light.point_light((6., 4., -6.), (0.3, 0.3, 0.4))]
    plane = Shape.plane().set_transform(translate(0., -1.5, 0.)).set_material(Material(reflective=0.2))
    wrap, counts = rt.obj_to_group(OCTA_OBJ)
    assert counts["triangles"] == 8 and counts["lines_ignored"] == 0
    wrap.set_transform(scale(2., 2., 2.))
    outer = Shape.group().set_transform(translate(0.5, 0.25, 1.))
    outer.add_child(wrap)
    outer.set_material(Material(color=(0.8, 0.9, 1.), diffuse=0.4, reflective=0.3, transparency=0.6, refractive_index=1.4))
    w.objects += [plane, outer]
    return w


def canonical(desc_ptr):
    """Expands a descriptor into per-node content that does not depend on how the tables are organised (index
    assignment, de-duplication, id values): what the renderers actually consume."""
    d = desc_ptr.contents if hasattr(desc_ptr, "contents") else desc_ptr

    def arr(p, n):
        return [p[i] for i in range(n)]

    def m16(p, i):
        return tuple(p[16 * i + k] + 0. for k in range(16))   # + 0.: -0.0 -> 0.0

    def v3(p, i):
        return tuple(p[3 * i + k] + 0. for k in range(3))

    id_rank, cls_rank = {}, {}
    nodes = []
    for n in range(d.n_nodes):
        kind = d.node_kind[n]
        e = {"kind": kind, "id": id_rank.setdefault(d.node_object_id[n], len(id_rank)), "parent": d.node_parent[n],
             "end": d.node_subtree_end[n], "inverse": m16(d.xform_inverse, d.node_xform[n]),
             "inverse_transpose": m16(d.xform_inverse_transpose, d.node_xform[n]), "casts_shadow": int(d.node_casts_shadow[n])}
        if kind == rt.RT_GROUP:
            e["box"] = tuple(d.group_box[6 * d.node_param[n] + k] for k in range(6))
        elif kind == rt.RT_CSG:
            e["op"] = d.csg_op[d.node_param[n]]
        else:
            chain, a = [], d.node_ancestor[n]
            while a >= 0:
                chain.append((m16(d.xform_inverse, d.ancestor_xform[a]), m16(d.xform_inverse_transpose, d.ancestor_xform[a])))
                a = d.ancestor_parent[a]
            e["ancestors"] = chain
            mi = d.node_material[n]
            mat = tuple(d.material_params[10 * mi + k] for k in range(10))
            pi = d.material_pattern[mi]
            pat = None if pi < 0 else (d.pattern_kind[pi], v3(d.pattern_a, pi), v3(d.pattern_b, pi), m16(d.pattern_inverse, pi))
            e["material"] = (mat, pat)
            c = d.node_eq_class[n]
            e["eq_class"] = -1 if c < 0 else cls_rank.setdefault(c, len(cls_rank))
            if kind in (rt.RT_CYLINDER, rt.RT_CONE):
                q = d.node_param[n]
                e["conic"] = (d.conic_minimum[q], d.conic_maximum[q], int(d.conic_closed[q]))
            if kind in (rt.RT_TRIANGLE, rt.RT_SMOOTH_TRIANGLE):
                q = d.node_param[n]
                e["tri"] = (v3(d.tri_p1, q), v3(d.tri_e1, q), v3(d.tri_e2, q), v3(d.tri_normal, q))
                if kind == rt.RT_SMOOTH_TRIANGLE:
                    sidx = d.tri_smooth[q]
                    e["smooth"] = (v3(d.smooth_n1, sidx), v3(d.smooth_n2, sidx), v3(d.smooth_n3, sidx))
        nodes.append(e)
    lights = [(v3(d.light_position, i), v3(d.light_intensity, i)) for i in range(d.n_lights)]
    return {"nodes": nodes, "lights": lights}
