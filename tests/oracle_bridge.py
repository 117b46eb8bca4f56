"""ctypes bridge to oracle/librt_oracle.so — the CPU restatement of the reference.  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os

import numpy as np

import rt_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "librt_oracle.so")
if not os.path.exists(ORACLE_PATH):
    raise ImportError(f"{ORACLE_PATH} is missing: run `python __graft_entry__.py` first")
olib = C.CDLL(ORACLE_PATH)

c_dp, c_ip, c_bp = rt.c_dp, rt.c_ip, rt.c_bp
DESC = C.POINTER(rt.SceneDesc)


class OrcStats(C.Structure):
    _fields_ = [("rays_primary", C.c_int64), ("rays_reflect", C.c_int64), ("rays_refract", C.c_int64),
                ("rays_shadow", C.c_int64), ("shaded_nodes", C.c_int64)]

    @property
    def total_rays(self):
        return self.rays_primary + self.rays_reflect + self.rays_refract + self.rays_shadow


olib.orc_render.argtypes = [DESC, C.POINTER(rt.CameraDesc), C.c_int, C.c_int, c_dp, c_ip, c_bp, C.POINTER(OrcStats)]
olib.orc_render_rows.argtypes = [DESC, C.POINTER(rt.CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_ip, c_bp, C.POINTER(OrcStats)]
olib.orc_color_at.argtypes = [DESC, C.c_int64, c_dp, c_dp, C.c_int, c_dp, c_ip, c_bp]
olib.orc_intersect_world.argtypes = [DESC, C.c_int64, c_dp, c_dp, C.c_int, c_ip, c_dp, c_ip]
olib.orc_is_shadowed.argtypes = [DESC, C.c_int64, c_dp, c_bp]
olib.orc_ray_for_pixel.argtypes = [C.POINTER(rt.CameraDesc), C.c_int64, c_ip, c_ip, c_dp, c_dp]
olib.orc_eval_list.argtypes = [DESC, c_dp, c_dp, C.c_int, c_dp, c_ip, c_dp, c_dp, C.c_int, C.c_int, C.c_int, c_dp, c_dp]
olib.orc_lighting.argtypes = [DESC, C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_int, c_dp]
olib.orc_normal_at.argtypes = [DESC, C.c_int, c_dp, C.c_double, C.c_double, c_dp]
olib.orc_world_to_object.argtypes = [DESC, C.c_int, c_dp, c_dp]
olib.orc_normal_to_world.argtypes = [DESC, C.c_int, c_dp, c_dp]
olib.orc_pattern_at_shape.argtypes = [DESC, C.c_int, C.c_int, c_dp, c_dp]
olib.orc_bound_intersects.argtypes = [c_dp, c_dp, c_dp]
olib.orc_intersection_allowed.argtypes = [C.c_int] * 4
olib.orc_local_intersect.argtypes = [DESC, C.c_int, c_dp, c_dp, c_dp, c_dp]
olib.orc_local_normal_at.argtypes = [DESC, C.c_int, c_dp, C.c_double, C.c_double, c_dp]


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _v(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))


def _ok(rc):
    if rc < 0:
        raise RuntimeError(f"oracle error {rc}")
    return rc


def render(world, camera, depth=rt.DEFAULT_RECURSION, stride=1, rows=None):
    """Camera::render by the oracle (rows = (y0, y1) restricts it to a band).  Returns an rt.Image whose stats is
    an OrcStats."""
    cd = camera.desc()
    rgb = np.zeros((camera.vsize, camera.hsize, 3))
    ids = np.zeros((camera.vsize, camera.hsize), dtype=np.int32)
    flags = np.zeros((camera.vsize, camera.hsize), dtype=np.uint8)
    st = OrcStats()
    y0, y1 = rows if rows is not None else (0, camera.vsize - 1)
    _ok(olib.orc_render_rows(world.flatten(), C.byref(cd), depth, stride, int(y0), int(y1), _dp(rgb), ids.ctypes.data_as(c_ip),
                             flags.ctypes.data_as(c_bp), C.byref(st)))
    return rt.Image(rgb, ids, flags, st)


def color_at_batch(world, origins, directions, remaining=rt.DEFAULT_RECURSION):
    o = np.ascontiguousarray(np.asarray(origins, dtype=np.float64).reshape(-1, 3))
    d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).reshape(-1, 3))
    n = len(o)
    rgb = np.zeros((n, 3))
    ids = np.zeros(n, dtype=np.int32)
    flags = np.zeros(n, dtype=np.uint8)
    _ok(olib.orc_color_at(world.flatten(), n, _dp(o), _dp(d), remaining, _dp(rgb), ids.ctypes.data_as(c_ip),
                          flags.ctypes.data_as(c_bp)))
    return rgb, ids, flags


def color_at(world, ray, remaining=rt.DEFAULT_RECURSION):
    return tuple(color_at_batch(world, [ray[0]], [ray[1]], remaining)[0][0])


def intersect_world_batch(world, origins, directions, max_hits=64):
    o = np.ascontiguousarray(np.asarray(origins, dtype=np.float64).reshape(-1, 3))
    d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).reshape(-1, 3))
    n = len(o)
    cnt = np.zeros(n, dtype=np.int32)
    t = np.zeros((n, max_hits))
    ids = np.zeros((n, max_hits), dtype=np.int32)
    _ok(olib.orc_intersect_world(world.flatten(), n, _dp(o), _dp(d), max_hits, cnt.ctypes.data_as(c_ip), _dp(t),
                                 ids.ctypes.data_as(c_ip)))
    return cnt, t, ids


def intersect_world(world, ray):
    cnt, t, ids = intersect_world_batch(world, [ray[0]], [ray[1]])
    return [(float(t[0, k]), int(ids[0, k])) for k in range(int(cnt[0]))]


def is_shadowed_batch(world, points):
    p = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
    nl = len(world.lights)
    out = np.zeros((len(p), max(nl, 1)), dtype=np.uint8)
    _ok(olib.orc_is_shadowed(world.flatten(), len(p), _dp(p), out.ctypes.data_as(c_bp)))
    return out[:, :nl].astype(bool)


def is_shadowed(world, point):
    return list(is_shadowed_batch(world, [point])[0])


def rays_for_pixels(camera, px, py):
    px = np.ascontiguousarray(np.asarray(px, dtype=np.int32))
    py = np.ascontiguousarray(np.asarray(py, dtype=np.int32))
    o = np.zeros((len(px), 4))
    d = np.zeros((len(px), 4))
    cd = camera.desc()
    _ok(olib.orc_ray_for_pixel(C.byref(cd), len(px), px.ctypes.data_as(c_ip), py.ctypes.data_as(c_ip), _dp(o), _dp(d)))
    return o, d


class Comps:
    def __init__(self, a):
        self.t = a[0]
        self.point, self.eye_vector, self.normal_vector = a[1:4], a[4:7], a[7:10]
        self.inside = bool(a[10])
        self.over_point, self.under_point, self.reflect_vector = a[11:14], a[14:17], a[17:20]
        self.n1, self.n2, self.flags = a[20], a[21], int(a[22])


PREPARE, SHADE_HIT, REFLECTED, REFRACTED, SCHLICK = range(5)


def eval_list(world, ray, xs, hit, what=PREPARE, remaining=rt.DEFAULT_RECURSION):
    """xs: list of (t, shape) or (t, shape, u, v), ALREADY SORTED by t as intersections! leaves it; hit: index.
    Returns (Comps, value) with value an rgb tuple (SHADE_HIT/REFLECTED/REFRACTED) or a float (SCHLICK)."""
    n = len(xs)
    t = np.array([x[0] for x in xs], dtype=np.float64)
    node = np.array([world.node_of(x[1]) for x in xs], dtype=np.int32)
    u = np.array([x[2] if len(x) > 2 else np.nan for x in xs], dtype=np.float64)
    v = np.array([x[3] if len(x) > 3 else np.nan for x in xs], dtype=np.float64)
    comps = np.zeros(27)
    val = np.zeros(3)
    o, d = _v(ray[0]), _v(ray[1])
    _ok(olib.orc_eval_list(world.flatten(), _dp(o), _dp(d), n, _dp(t), node.ctypes.data_as(c_ip), _dp(u), _dp(v), hit,
                           remaining, what, _dp(comps), _dp(val)))
    return Comps(comps), (float(val[0]) if what == SCHLICK else tuple(val))


def lighting(world, shape, light_index, point, eye, normal, in_shadow):
    out = np.zeros(3)
    _ok(olib.orc_lighting(world.flatten(), world.node_of(shape), light_index, _dp(_v(point)), _dp(_v(eye)),
                          _dp(_v(normal)), int(in_shadow), _dp(out)))
    return tuple(out)


def normal_at(world, shape, world_point, u=np.nan, v=np.nan):
    out = np.zeros(3)
    _ok(olib.orc_normal_at(world.flatten(), world.node_of(shape), _dp(_v(world_point)), u, v, _dp(out)))
    return tuple(out)


def world_to_object(world, shape, p):
    out = np.zeros(3)
    _ok(olib.orc_world_to_object(world.flatten(), world.node_of(shape), _dp(_v(p)), _dp(out)))
    return tuple(out)


def normal_to_world(world, shape, n):
    out = np.zeros(3)
    _ok(olib.orc_normal_to_world(world.flatten(), world.node_of(shape), _dp(_v(n)), _dp(out)))
    return tuple(out)


def pattern_at_shape(world, shape, p):
    """The shape's (root-ancestor) material must carry a pattern."""
    d = world.flatten().contents
    node = world.node_of(shape)
    pat = d.material_pattern[d.node_material[node]]
    assert pat >= 0
    out = np.zeros(3)
    _ok(olib.orc_pattern_at_shape(world.flatten(), pat, node, _dp(_v(p)), _dp(out)))
    return tuple(out)


def bound_intersects(box_min, box_max, origin, direction):
    box = _v(list(box_min) + list(box_max))
    return bool(olib.orc_bound_intersects(_dp(box), _dp(_v(origin)), _dp(_v(direction))))


def intersection_allowed(op, lhit, inl, inr):
    code = {"union": 0, "intersection": 1, "difference": 2}.get(op, 3)
    return bool(olib.orc_intersection_allowed(code, int(lhit), int(inl), int(inr)))


def local_intersect(world, shape, origin, direction):
    """Primitive::intersect on a local-space ray.  Returns (list of t in push order, list of (u, v))."""
    t = np.zeros(4)
    uv = np.zeros(8)
    n = olib.orc_local_intersect(world.flatten(), world.node_of(shape), _dp(_v(origin)), _dp(_v(direction)), _dp(t), _dp(uv))
    return [float(x) for x in t[:n]], [(float(uv[2 * i]), float(uv[2 * i + 1])) for i in range(n)]


def local_normal_at(world, shape, p, u=np.nan, v=np.nan):
    out = np.zeros(3)
    _ok(olib.orc_local_normal_at(world.flatten(), world.node_of(shape), _dp(_v(p)), u, v, _dp(out)))
    return tuple(out)
