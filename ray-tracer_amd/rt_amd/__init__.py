"""rt_amd — ctypes binding of libraytracer_amd.so (include/rt_abi.h + include/rt_host.h).

Python is plumbing here: scene construction goes through the C++ host mirror of the reference crate's API
(ray-tracer_amd/host), rendering goes through the C ABI into the HIP kernels.  Class and method names follow the
reference (World, Camera, Shape, Material, Pattern, Light, translate/scale/rotate/shear/view_transform,
obj_to_group) so tests read like the crate's own.  There is no CPU fallback: importing fails if the library is
missing, and every compute call fails if there is no GPU.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libraytracer_amd.so")  # RT_AMD_LIB: experiment builds
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` (build()) first; there is no fallback path")


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and ask for it by that file
    name, which the loader does not match against an already loaded /opt/rocm/lib/libamdhip64.so.7 — so `import rt_amd`
    followed by `import torch` put TWO runtimes into the process, and whichever of them touched the GPU second could
    find no device.  If a torch with a bundled runtime is installed (not necessarily imported), its copy is loaded first
    and this library binds to it by soname; without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch's runtime is in the process already: the soname matches
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(bundled):
        return
    # Bind to torch's copy only if it is the runtime generation this library was linked against (its DT_NEEDED names
    # libamdhip64.so.<major>): a bundled runtime of another major version would leave two runtimes in the process after
    # all, or bind the library to an ABI it was not built for.
    want, have = _needed_hip_major(LIB_PATH), _needed_hip_major(bundled)   # no HIP call: nothing touches the GPU at import
    global HIP_RUNTIME_BOUND
    if want is not None and have is not None and have != want:
        import warnings
        warnings.warn(f"rt_amd: torch bundles libamdhip64.so.{have}, libraytracer_amd.so was linked against libamdhip64.so.{want}: "
                      "not preloading it (the library binds the system runtime; import torch FIRST if both must share one process)")
        return
    try:
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)
    except OSError as e:
        import warnings
        warnings.warn(f"rt_amd: could not preload torch's HIP runtime ({e}); the system runtime is used")
        return
    HIP_RUNTIME_BOUND = f"{bundled} (soname libamdhip64.so.{have})"


def _needed_hip_major(path):
    """Major version in the file's DT_NEEDED / DT_SONAME text for libamdhip64 (None if it cannot be read)."""
    import re
    try:
        with open(path, "rb") as f:
            m = re.search(rb"libamdhip64\.so\.(\d+)", f.read())
        return int(m.group(1)) if m else None
    except OSError:
        return None


HIP_RUNTIME_BOUND = "system (the library's own DT_NEEDED)"
_share_torch_hip_runtime()
lib = C.CDLL(LIB_PATH)
if os.environ.get("RT_AMD_VERBOSE"):
    import sys as _sys
    print(f"rt_amd: HIP runtime bound: {HIP_RUNTIME_BOUND}", file=_sys.stderr)

EPSILON = 0.00001
DEFAULT_RECURSION = 5

RT_SPHERE, RT_PLANE, RT_CUBE, RT_CYLINDER, RT_CONE, RT_TRIANGLE, RT_SMOOTH_TRIANGLE, RT_GROUP, RT_CSG = range(9)
RT_PAT_STRIPED, RT_PAT_GRADIENT, RT_PAT_RING, RT_PAT_CHECKERED, RT_PAT_RING_GRADIENT, RT_PAT_TEST = range(6)
RT_FLAG_REF_PANIC_UV, RT_FLAG_NAN_T, RT_FLAG_LIST_OVERFLOW = 1, 2, 4
RT_OK, RT_ERR_BAD_ARG, RT_ERR_UNSUPPORTED, RT_ERR_HIP, RT_ERR_NO_DEVICE, RT_ERR_OVERFLOW = 0, -1, -2, -3, -4, -5

# columns of rt_prepare_computations' rows (rt_abi.h RT_COMP_*)
RT_COMP_COUNT = 25
COMP = {"t": slice(0, 1), "point": slice(1, 4), "eye": slice(4, 7), "normal": slice(7, 10), "inside": slice(10, 11), "over": slice(11, 14),
        "under": slice(14, 17), "reflect": slice(17, 20), "n1": slice(20, 21), "n2": slice(21, 22), "schlick": slice(22, 23),
        "u": slice(23, 24), "v": slice(24, 25)}

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_bp = C.POINTER(C.c_uint8)


class RtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"rt error {code}: {text}")
        self.code = code


class SceneDesc(C.Structure):  # rt_scene_desc
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_nodes", C.c_int32),
        ("node_kind", c_ip), ("node_object_id", c_ip), ("node_parent", c_ip), ("node_subtree_end", c_ip),
        ("node_xform", c_ip), ("node_ancestor", c_ip), ("node_material", c_ip), ("node_casts_shadow", c_bp),
        ("node_eq_class", c_ip), ("node_param", c_ip),
        ("n_xforms", C.c_int32), ("xform_inverse", c_dp), ("xform_inverse_transpose", c_dp),
        ("n_ancestors", C.c_int32), ("ancestor_xform", c_ip), ("ancestor_parent", c_ip),
        ("n_groups", C.c_int32), ("group_box", c_dp),
        ("n_csg", C.c_int32), ("csg_op", c_ip),
        ("n_conics", C.c_int32), ("conic_minimum", c_dp), ("conic_maximum", c_dp), ("conic_closed", c_bp),
        ("n_triangles", C.c_int32), ("tri_p1", c_dp), ("tri_e1", c_dp), ("tri_e2", c_dp), ("tri_normal", c_dp),
        ("tri_smooth", c_ip),
        ("n_smooth", C.c_int32), ("smooth_n1", c_dp), ("smooth_n2", c_dp), ("smooth_n3", c_dp),
        ("n_materials", C.c_int32), ("material_params", c_dp), ("material_pattern", c_ip),
        ("n_patterns", C.c_int32), ("pattern_kind", c_ip), ("pattern_a", c_dp), ("pattern_b", c_dp),
        ("pattern_inverse", c_dp),
        ("n_lights", C.c_int32), ("light_position", c_dp), ("light_intensity", c_dp),
    ]


class CameraDesc(C.Structure):  # rt_camera_desc
    _fields_ = [("hsize", C.c_int32), ("vsize", C.c_int32), ("invertible", C.c_int32), ("_pad", C.c_int32),
                ("inverse", C.c_double * 16), ("half_width", C.c_double), ("half_height", C.c_double),
                ("pixel_size", C.c_double)]


class RenderOpts(C.Structure):  # rt_render_opts
    _fields_ = [("max_depth", C.c_int32), ("rank", C.c_int32), ("n_ranks", C.c_int32), ("want_counters", C.c_int32),
                ("stream", C.c_void_p)]


class Stats(C.Structure):  # rt_stats
    _fields_ = [("rays_primary", C.c_int64), ("rays_reflect", C.c_int64), ("rays_refract", C.c_int64),
                ("rays_shadow", C.c_int64), ("rays_container", C.c_int64), ("shaded_nodes", C.c_int64),
                ("levels", C.c_int32), ("_pad", C.c_int32),
                ("ms_total", C.c_double), ("ms_raygen", C.c_double), ("ms_trace", C.c_double),
                ("ms_shadow", C.c_double), ("ms_walk", C.c_double), ("ms_queue", C.c_double),
                ("ms_kshade", C.c_double), ("n_trace_launches", C.c_int32), ("n_walk_launches", C.c_int32),
                ("work_trace", C.c_int64 * 13), ("work_shadow", C.c_int64 * 13),
                ("work_container", C.c_int64 * 13)]

    @property
    def total_rays(self):
        return self.rays_primary + self.rays_reflect + self.rays_refract + self.rays_shadow

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            if name == "_pad":
                continue
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


class CMaterial(C.Structure):  # rth_material
    _fields_ = [("color", C.c_double * 3), ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double), ("transparency", C.c_double),
                ("refractive_index", C.c_double), ("pattern_kind", C.c_int32), ("_pad", C.c_int32),
                ("pattern_a", C.c_double * 3), ("pattern_b", C.c_double * 3), ("pattern_transform", C.c_double * 16)]


# ---- prototypes: every symbol include/*.h declares -------------------------------------------------------
ABI_SYMBOLS = ["rt_device_count", "rt_scene_create", "rt_scene_destroy", "rt_render", "rt_render_begin", "rt_render_frame_begin", "rt_render_shard_begin", "rt_render_end", "rt_shard_pixel_count",
               "rt_render_shard", "rt_scene_meshes_built_on_gpu", "rt_scene_meshes_shared", "rt_render_frame", "rt_assemble", "rt_assemble_strided", "rt_quantise", "rt_assemble_quantised", "rt_color_at", "rt_intersect_world", "rt_is_shadowed", "rt_prepare_computations", "rt_normal_at", "rt_lighting",
               "rt_ray_for_pixel", "rt_canvas_to_ppm", "rt_canvas_to_ppm_noalloc", "rt_scene_create_multi", "rt_scene_create_opts", "rt_render_multi", "rt_host_alloc", "rt_host_free",
               "rt_scene_gpu_count", "rt_scene_gather_kind", "rt_last_error"]
HOST_SYMBOLS = ["rth_identity", "rth_translate", "rth_scale", "rth_rotate", "rth_shear", "rth_view_transform",
                "rth_matmul", "rth_inverse", "rth_material_default", "rth_shape_new", "rth_glass_sphere",
                "rth_triangle_new", "rth_smooth_triangle_new", "rth_csg_new", "rth_hexagon", "rth_shape_clone",
                "rth_shape_free", "rth_shape_kind", "rth_shape_id", "rth_shape_parent", "rth_shape_set_transform",
                "rth_shape_set_material", "rth_shape_get_material", "rth_shape_set_casts_shadow",
                "rth_shape_set_minimum", "rth_shape_set_maximum", "rth_shape_set_closed", "rth_group_add_child",
                "rth_shape_child_count", "rth_shape_child", "rth_shape_divide", "rth_shape_bounds_of",
                "rth_shape_parent_space_bounds_of", "rth_shape_includes", "rth_shape_eq", "rth_triangle_data",
                "rth_obj_to_group", "rth_obj_file_to_group", "rth_world_create", "rth_world_free", "rth_world_flatten", "rth_world_scene",
                "rth_camera_desc", "rth_canvas_to_ppm"]

lib.rt_last_error.restype = C.c_char_p
lib.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
lib.rt_scene_destroy.argtypes = [C.c_void_p]
lib.rt_scene_destroy.restype = None
lib.rt_render.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderOpts), c_dp, c_ip, c_bp, C.POINTER(Stats)]
lib.rt_render_begin.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderOpts), c_dp, c_ip, c_bp]
lib.rt_render_frame_begin.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p, C.c_void_p]
lib.rt_render_shard_begin.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p, C.c_void_p]
lib.rt_render_end.argtypes = [C.c_void_p]
lib.rt_shard_pixel_count.argtypes = [C.POINTER(CameraDesc), C.c_int32, C.c_int32]
lib.rt_shard_pixel_count.restype = C.c_int64
lib.rt_render_shard.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.POINTER(Stats)]
lib.rt_scene_meshes_built_on_gpu.argtypes = [C.c_void_p]
lib.rt_scene_meshes_built_on_gpu.restype = C.c_int32
lib.rt_scene_meshes_shared.argtypes = [C.c_void_p]
lib.rt_scene_meshes_shared.restype = C.c_int32
lib.rt_render_frame.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderOpts), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.POINTER(Stats)]
lib.rt_assemble.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p]
lib.rt_assemble_strided.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
lib.rt_quantise.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
lib.rt_assemble_quantised.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
lib.rt_color_at.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, C.c_int32, c_dp, c_ip, c_bp, C.POINTER(Stats)]
lib.rt_intersect_world.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, C.c_int32, c_ip, c_dp, c_ip]
lib.rt_is_shadowed.argtypes = [C.c_void_p, C.c_int64, c_dp, c_bp]
lib.rt_prepare_computations.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, c_ip, c_dp, c_bp]
lib.rt_normal_at.argtypes = [C.c_void_p, C.c_int32, C.c_int64, c_dp, c_dp, c_dp]
lib.rt_lighting.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, c_dp, c_dp, c_dp, c_bp, c_dp]
lib.rt_ray_for_pixel.argtypes = [C.POINTER(CameraDesc), C.c_int64, c_ip, c_ip, c_dp, c_dp]
lib.rt_canvas_to_ppm.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_char_p, C.c_int64]
lib.rt_canvas_to_ppm.restype = C.c_int64
lib.rt_canvas_to_ppm_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p]
lib.rt_canvas_to_ppm_noalloc.restype = C.c_int64
lib.rt_scene_create_multi.argtypes = [C.POINTER(SceneDesc), C.c_int32, c_ip, C.POINTER(C.c_void_p)]


lib.rt_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
lib.rt_host_free.argtypes = [C.c_void_p]
lib.rt_host_free.restype = None


def pinned_array(shape, dtype=np.float64):
    """A numpy array over page-locked host memory from rt_host_alloc (freed with the array): the fast target for
    rt_render's host output buffers."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    _chk(lib.rt_host_alloc(n, C.byref(p)))

    class _Owner:
        def __init__(self, ptr): self.ptr = ptr
        def __del__(self):
            try: lib.rt_host_free(self.ptr)
            except Exception: pass
    raw = (C.c_char * max(n, 1)).from_address(p.value)
    raw._owner = _Owner(p)   # keeps the allocation alive as long as any view of `raw` exists
    a = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    a[...] = 0
    return a


class SceneOpts(C.Structure):
    """rt_scene_opts (include/rt_abi.h): -1 = library default in every field."""
    _fields_ = [("struct_size", C.c_int32), ("cull_ranges", C.c_int32), ("passive_triangles", C.c_int32), ("frame_graph", C.c_int32),
                ("device_bvh", C.c_int32), ("gather", C.c_int32), ("verbose", C.c_int32),
                ("share_meshes", C.c_int32), ("fuse_output", C.c_int32), ("fuse_spawn", C.c_int32), ("top_level_tree", C.c_int32),
                ("four_components", C.c_int32), ("walk_bound", C.c_int32), ("render_lanes", C.c_int32), ("fuse_raygen", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        for name, _ in self._fields_:
            setattr(self, name, -1)
        self.struct_size = C.sizeof(SceneOpts)
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("rt_scene_opts has no field " + k)
            setattr(self, k, v)


lib.rt_scene_create_opts.argtypes = [C.POINTER(SceneDesc), C.c_int32, c_ip, C.POINTER(SceneOpts), C.POINTER(C.c_void_p)]
lib.rt_render_multi.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderOpts), c_dp, c_ip, c_bp, c_bp, C.POINTER(Stats)]
lib.rt_scene_gpu_count.argtypes = [C.c_void_p]
lib.rt_scene_gpu_count.restype = C.c_int32
lib.rt_scene_gather_kind.argtypes = [C.c_void_p]
lib.rt_scene_gather_kind.restype = C.c_int32

for _n in ("rth_identity", "rth_translate", "rth_scale", "rth_rotate", "rth_shear", "rth_view_transform", "rth_matmul",
           "rth_material_default", "rth_shape_free", "rth_world_free"):
    getattr(lib, _n).restype = None
lib.rth_identity.argtypes = [c_dp]
lib.rth_translate.argtypes = [C.c_double] * 3 + [c_dp]
lib.rth_scale.argtypes = [C.c_double] * 3 + [c_dp]
lib.rth_rotate.argtypes = [C.c_int, C.c_double, c_dp]
lib.rth_shear.argtypes = [C.c_double] * 6 + [c_dp]
lib.rth_view_transform.argtypes = [c_dp, c_dp, c_dp, c_dp]
lib.rth_matmul.argtypes = [c_dp, c_dp, c_dp]
lib.rth_inverse.argtypes = [c_dp, c_dp]
lib.rth_material_default.argtypes = [C.POINTER(CMaterial)]
lib.rth_triangle_new.argtypes = [c_dp] * 3
lib.rth_smooth_triangle_new.argtypes = [c_dp] * 6
lib.rth_csg_new.argtypes = [C.c_char_p, C.c_int, C.c_int]
lib.rth_shape_set_transform.argtypes = [C.c_int, c_dp]
lib.rth_shape_set_material.argtypes = [C.c_int, C.POINTER(CMaterial)]
lib.rth_shape_get_material.argtypes = [C.c_int, C.POINTER(CMaterial)]
lib.rth_shape_set_minimum.argtypes = [C.c_int, C.c_double]
lib.rth_shape_set_maximum.argtypes = [C.c_int, C.c_double]
lib.rth_shape_bounds_of.argtypes = [C.c_int, c_dp]
lib.rth_shape_parent_space_bounds_of.argtypes = [C.c_int, c_dp]
lib.rth_triangle_data.argtypes = [C.c_int, c_dp]
lib.rth_obj_to_group.argtypes = [C.c_char_p, C.c_int64, c_ip]
lib.rth_obj_file_to_group.argtypes = [C.c_char_p, C.c_int64, c_ip]
lib.rth_world_create.argtypes = [c_ip, C.c_int, c_dp, c_dp, C.c_int]
lib.rth_world_flatten.argtypes = [C.c_int]
lib.rth_world_flatten.restype = C.POINTER(SceneDesc)
lib.rth_world_scene.argtypes = [C.c_int, C.c_int]
lib.rth_world_scene.restype = C.c_void_p
lib.rth_camera_desc.argtypes = [C.c_int, C.c_int, C.c_double, c_dp, C.POINTER(CameraDesc)]
lib.rth_canvas_to_ppm.argtypes = [c_dp, C.c_int, C.c_int, C.c_char_p, C.c_int64]
lib.rth_canvas_to_ppm.restype = C.c_int64


def last_error():
    return lib.rt_last_error().decode("utf-8", "replace")


def _chk(rc):
    if rc < 0:
        raise RtError(rc, last_error())
    return rc


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _vec3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64)[:3])


def device_count():
    return int(lib.rt_device_count())


# ---- matrix.rs / transformation.rs ----------------------------------------------------------------------
class Matrix:
    def __init__(self, values=None):
        self.m = np.zeros(16, dtype=np.float64)
        if values is None:
            lib.rth_identity(_dp(self.m))
        else:
            self.m[:] = np.asarray(values, dtype=np.float64).reshape(16)

    @staticmethod
    def identity(size=4):
        assert size == 4
        return Matrix()

    def __mul__(self, other):
        out = Matrix()
        lib.rth_matmul(_dp(self.m), _dp(other.m), _dp(out.m))
        return out

    def inverse(self):
        out = Matrix()
        return out if lib.rth_inverse(_dp(self.m), _dp(out.m)) else None

    def __getitem__(self, rc):
        return float(self.m[rc[0] * 4 + rc[1]])


class Axis:
    X, Y, Z = 0, 1, 2


def translate(x, y, z):
    out = Matrix()
    lib.rth_translate(x, y, z, _dp(out.m))
    return out


def scale(x, y, z):
    out = Matrix()
    lib.rth_scale(x, y, z, _dp(out.m))
    return out


def rotate(radians, axis):
    out = Matrix()
    lib.rth_rotate(axis, radians, _dp(out.m))
    return out


def shear(xy, xz, yx, yz, zx, zy):
    out = Matrix()
    lib.rth_shear(xy, xz, yx, yz, zx, zy, _dp(out.m))
    return out


def view_transform(frm, to, up):
    out = Matrix()
    lib.rth_view_transform(_dp(_vec3(frm)), _dp(_vec3(to)), _dp(_vec3(up)), _dp(out.m))
    return out


# ---- color / pattern / material / light ------------------------------------------------------------------
BLACK = (0.0, 0.0, 0.0)
WHITE = (1.0, 1.0, 1.0)


def Color(r, g, b):
    return (float(r), float(g), float(b))


class Pattern:
    def __init__(self, kind, a=BLACK, b=BLACK):
        self.kind, self.a, self.b = kind, a, b
        self.transform = Matrix()

    def set_transform(self, m):
        self.transform = m
        return self

    @staticmethod
    def striped(a, b):
        return Pattern(RT_PAT_STRIPED, a, b)

    @staticmethod
    def gradient(a, b):
        return Pattern(RT_PAT_GRADIENT, a, b)

    @staticmethod
    def ring(a, b):
        return Pattern(RT_PAT_RING, a, b)

    @staticmethod
    def checkered(a, b):
        return Pattern(RT_PAT_CHECKERED, a, b)

    @staticmethod
    def ring_gradient(a, b):
        return Pattern(RT_PAT_RING_GRADIENT, a, b)

    @staticmethod
    def test():
        return Pattern(RT_PAT_TEST)


class Material:
    """struct Material (src/material.rs:11-21) with the with_* builders."""

    def __init__(self, **kw):
        self.color = WHITE
        self.ambient, self.diffuse, self.specular, self.shininess = 0.1, 0.9, 0.9, 200.0
        self.pattern = None
        self.reflective, self.transparency, self.refractive_index = 0.0, 0.0, 1.0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    def _with(self, k, v):
        setattr(self, k, v)
        return self

    def with_color(self, v): return self._with("color", v)
    def with_ambient(self, v): return self._with("ambient", v)
    def with_diffuse(self, v): return self._with("diffuse", v)
    def with_specular(self, v): return self._with("specular", v)
    def with_shininess(self, v): return self._with("shininess", v)
    def with_pattern(self, v): return self._with("pattern", v)
    def with_reflective(self, v): return self._with("reflective", v)
    def with_transparency(self, v): return self._with("transparency", v)
    def with_refractive_index(self, v): return self._with("refractive_index", v)

    def _c(self):
        m = CMaterial()
        m.color[:] = self.color
        m.ambient, m.diffuse, m.specular, m.shininess = self.ambient, self.diffuse, self.specular, self.shininess
        m.reflective, m.transparency, m.refractive_index = self.reflective, self.transparency, self.refractive_index
        m.pattern_kind = -1
        ident = Matrix()
        m.pattern_transform[:] = list(ident.m)
        if self.pattern is not None:
            m.pattern_kind = self.pattern.kind
            m.pattern_a[:] = self.pattern.a
            m.pattern_b[:] = self.pattern.b
            m.pattern_transform[:] = list(self.pattern.transform.m)
        return m

    @staticmethod
    def _from_c(m):
        out = Material(color=tuple(m.color), ambient=m.ambient, diffuse=m.diffuse, specular=m.specular,
                       shininess=m.shininess, reflective=m.reflective, transparency=m.transparency,
                       refractive_index=m.refractive_index)
        if m.pattern_kind >= 0:
            out.pattern = Pattern(m.pattern_kind, tuple(m.pattern_a), tuple(m.pattern_b))
            out.pattern.transform = Matrix(list(m.pattern_transform))
        return out


class Light:
    def __init__(self, position, intensity):
        self.position, self.intensity = tuple(position)[:3], tuple(intensity)

    @staticmethod
    def point_light(position, intensity):
        return Light(position, intensity)


# ---- shapes ------------------------------------------------------------------------------------------------
class Shape:
    """A handle on a C++ ray_tracer::Shape value.  Rust's `clone()` is .clone(); add_child stores a clone."""

    def __init__(self, handle):
        _chk(handle)
        self.h = handle

    def __del__(self):
        try:
            lib.rth_shape_free(self.h)
        except Exception:
            pass

    @staticmethod
    def sphere(): return Shape(lib.rth_shape_new(RT_SPHERE))
    @staticmethod
    def glass_sphere(): return Shape(lib.rth_glass_sphere())
    @staticmethod
    def plane(): return Shape(lib.rth_shape_new(RT_PLANE))
    @staticmethod
    def cube(): return Shape(lib.rth_shape_new(RT_CUBE))
    @staticmethod
    def cylinder(): return Shape(lib.rth_shape_new(RT_CYLINDER))
    @staticmethod
    def cone(): return Shape(lib.rth_shape_new(RT_CONE))
    @staticmethod
    def group(): return Shape(lib.rth_shape_new(RT_GROUP))

    @staticmethod
    def triangle(p1, p2, p3):
        return Shape(lib.rth_triangle_new(_dp(_vec3(p1)), _dp(_vec3(p2)), _dp(_vec3(p3))))

    @staticmethod
    def smooth_triangle(p1, p2, p3, n1, n2, n3):
        return Shape(lib.rth_smooth_triangle_new(*[_dp(_vec3(v)) for v in (p1, p2, p3, n1, n2, n3)]))

    @staticmethod
    def csg(operation, left, right):
        return Shape(lib.rth_csg_new(operation.encode(), left.h, right.h))

    @staticmethod
    def hexagon():
        return Shape(lib.rth_hexagon())

    def clone(self): return Shape(lib.rth_shape_clone(self.h))
    def kind(self): return _chk(lib.rth_shape_kind(self.h))
    def get_id(self): return _chk(lib.rth_shape_id(self.h))
    def get_parent(self):
        p = _chk(lib.rth_shape_parent(self.h))
        return p if p else None

    def set_transform(self, m):
        _chk(lib.rth_shape_set_transform(self.h, _dp(m.m)))
        return self

    def set_material(self, material):
        cm = material._c()
        _chk(lib.rth_shape_set_material(self.h, C.byref(cm)))
        return self

    def get_material(self):
        cm = CMaterial()
        _chk(lib.rth_shape_get_material(self.h, C.byref(cm)))
        return Material._from_c(cm)

    def set_casts_shadow(self, v):
        _chk(lib.rth_shape_set_casts_shadow(self.h, int(bool(v))))
        return self

    def set_minimum(self, v):
        _chk(lib.rth_shape_set_minimum(self.h, float(v)))
        return self

    def set_maximum(self, v):
        _chk(lib.rth_shape_set_maximum(self.h, float(v)))
        return self

    def set_closed(self, v):
        _chk(lib.rth_shape_set_closed(self.h, int(bool(v))))
        return self

    def add_child(self, child):
        _chk(lib.rth_group_add_child(self.h, child.h))
        return self

    def child_count(self): return _chk(lib.rth_shape_child_count(self.h))
    def child(self, i): return Shape(lib.rth_shape_child(self.h, i))
    def get_shapes(self): return [self.child(i) for i in range(self.child_count())]

    def divide(self, threshold):
        _chk(lib.rth_shape_divide(self.h, int(threshold)))
        return self

    def bounds_of(self):
        out = np.zeros(6)
        _chk(lib.rth_shape_bounds_of(self.h, _dp(out)))
        return out[:3].copy(), out[3:].copy()

    def parent_space_bounds_of(self):
        out = np.zeros(6)
        _chk(lib.rth_shape_parent_space_bounds_of(self.h, _dp(out)))
        return out[:3].copy(), out[3:].copy()

    def includes(self, other): return bool(_chk(lib.rth_shape_includes(self.h, other.h)))
    def __eq__(self, other): return isinstance(other, Shape) and bool(_chk(lib.rth_shape_eq(self.h, other.h)))
    __hash__ = None

    def triangle_data(self):
        out = np.zeros(27)
        _chk(lib.rth_triangle_data(self.h, _dp(out)))
        return out.reshape(9, 3)


def obj_to_group(text, packed=None):
    """parse_obj_file + obj_to_group (src/obj_file.rs:29-172).  Returns (main group, counts dict).
    packed (default: yes, unless RT_HOST_OBJ_SHAPES=1): the group keeps its triangles as packed records until something
    asks for them as shapes (rth_obj_file_to_group) — same ids, same flattened tables, a fraction of the time."""
    raw = text.encode() if isinstance(text, str) else text
    counts = np.zeros(5, dtype=np.int32)
    if packed is None:
        packed = os.environ.get("RT_HOST_OBJ_SHAPES", "0") != "1"
    fn = lib.rth_obj_file_to_group if packed else lib.rth_obj_to_group
    g = Shape(fn(raw, len(raw), counts.ctypes.data_as(c_ip)))
    keys = ("lines_ignored", "vertices", "normals", "groups", "triangles")
    return g, dict(zip(keys, (int(c) for c in counts)))


# ---- world / camera ---------------------------------------------------------------------------------------
class Image:
    def __init__(self, rgb, hit_id, flags, stats):
        self.rgb, self.hit_id, self.flags, self.stats = rgb, hit_id, flags, stats

    def pixel_at(self, x, y):
        return tuple(self.rgb[y, x])

    def quantised(self):
        """Canvas::scale_color per channel (src/canvas.rs:86-98)."""
        c = self.rgb
        q = np.where(np.isnan(c), 0.0, np.floor(np.abs(c * 255.0) + 0.5) * np.sign(c))
        q = np.where(c > 1.0, 255.0, q)
        q = np.where(c < 0.0, 0.0, q)
        return q.astype(np.int32)

    def to_ppm(self):
        v, h, _ = self.rgb.shape
        flat = np.ascontiguousarray(self.rgb.reshape(-1))
        n = lib.rth_canvas_to_ppm(_dp(flat), h, v, None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        lib.rth_canvas_to_ppm(_dp(flat), h, v, buf, n)
        return buf.raw[:n].decode()


class World:
    def __init__(self):
        self.objects, self.lights = [], []
        self._h = None  # built on first use; call invalidate() after changing objects / lights

    @staticmethod
    def default():
        """impl Default for World (src/world.rs:20-39)."""
        w = World()
        w.lights.append(Light.point_light((-10., 10., -10.), WHITE))
        s1 = Shape.sphere().set_material(Material(color=(0.8, 1., 0.6), diffuse=0.7, specular=0.2))
        s2 = Shape.sphere().set_transform(scale(0.5, 0.5, 0.5))
        w.objects += [s1, s2]
        return w

    def invalidate(self):
        if self._h is not None:
            lib.rth_world_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.invalidate()
        except Exception:
            pass

    def handle(self):
        if self._h is not None:
            return self._h
        hs = np.array([o.h for o in self.objects], dtype=np.int32)
        lp = np.array([l.position for l in self.lights], dtype=np.float64).reshape(-1)
        li = np.array([l.intensity for l in self.lights], dtype=np.float64).reshape(-1)
        self._h = _chk(lib.rth_world_create(hs.ctypes.data_as(c_ip), len(self.objects), _dp(lp) if len(lp) else None,
                                            _dp(li) if len(li) else None, len(self.lights)))
        return self._h

    def flatten(self):
        """POINTER(SceneDesc) owned by the world handle."""
        p = lib.rth_world_flatten(self.handle())
        if not p:
            raise RtError(RT_ERR_UNSUPPORTED, last_error())
        return p

    def scene(self, device=0):
        h = self.handle()
        s = lib.rth_world_scene(h, device)
        if not s:
            raise RtError(RT_ERR_HIP, last_error())
        return s

    def node_of(self, shape):
        """Index of the (first) flattened node carrying this shape's object id."""
        d = self.flatten().contents
        ids = np.ctypeslib.as_array(d.node_object_id, shape=(d.n_nodes,))
        hits = np.nonzero(ids == shape.get_id())[0]
        if len(hits) == 0:
            raise KeyError("shape is not in this world")
        return int(hits[0])

    # World::color_at (src/world.rs:81-90), batched
    def color_at_batch(self, origins, directions, remaining=DEFAULT_RECURSION, want_stats=False):
        o = np.ascontiguousarray(np.asarray(origins, dtype=np.float64).reshape(-1, 3))
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).reshape(-1, 3))
        n = len(o)
        rgb = np.zeros((n, 3))
        ids = np.zeros(n, dtype=np.int32)
        flags = np.zeros(n, dtype=np.uint8)
        st = Stats()
        _chk(lib.rt_color_at(self.scene(), n, _dp(o), _dp(d), remaining, _dp(rgb), ids.ctypes.data_as(c_ip),
                             flags.ctypes.data_as(c_bp), C.byref(st) if want_stats else None))
        return (rgb, ids, flags, st) if want_stats else (rgb, ids, flags)

    def color_at(self, ray, remaining=DEFAULT_RECURSION):
        rgb, _, _ = self.color_at_batch([ray[0]], [ray[1]], remaining)
        return tuple(rgb[0])

    # World::intersect_world (src/world.rs:46-55)
    def intersect_world_batch(self, origins, directions, max_hits=64):
        o = np.ascontiguousarray(np.asarray(origins, dtype=np.float64).reshape(-1, 3))
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).reshape(-1, 3))
        n = len(o)
        cnt = np.zeros(n, dtype=np.int32)
        t = np.zeros((n, max_hits))
        ids = np.zeros((n, max_hits), dtype=np.int32)
        _chk(lib.rt_intersect_world(self.scene(), n, _dp(o), _dp(d), max_hits, cnt.ctypes.data_as(c_ip), _dp(t),
                                    ids.ctypes.data_as(c_ip)))
        return cnt, t, ids

    def intersect_world(self, ray):
        cnt, t, ids = self.intersect_world_batch([ray[0]], [ray[1]])
        return [(float(t[0, k]), int(ids[0, k])) for k in range(int(cnt[0]))]

    # Intersection::prepare_computations (src/intersection.rs:38-95) for the hit of each ray
    def prepare_computations_batch(self, origins, directions):
        """Returns (hit object ids [n], comps [n, RT_COMP_COUNT], flags [n]); columns: COMP dict below."""
        o = np.ascontiguousarray(np.asarray(origins, dtype=np.float64).reshape(-1, 3))
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).reshape(-1, 3))
        n = len(o)
        ids = np.zeros(n, dtype=np.int32)
        comps = np.zeros((n, RT_COMP_COUNT))
        flags = np.zeros(n, dtype=np.uint8)
        _chk(lib.rt_prepare_computations(self.scene(), n, _dp(o), _dp(d), ids.ctypes.data_as(c_ip), _dp(comps), flags.ctypes.data_as(c_bp)))
        return ids, comps, flags

    # Shape::normal_at (src/shape.rs:92-108)
    def normal_at_batch(self, shape, points, uv=None):
        p = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        q = None if uv is None else np.ascontiguousarray(np.asarray(uv, dtype=np.float64).reshape(-1, 2))
        out = np.zeros_like(p)
        _chk(lib.rt_normal_at(self.scene(), self.node_of(shape), len(p), _dp(p), None if q is None else _dp(q), _dp(out)))
        return out

    # Material::lighting (src/material.rs:97-129) with `shape`'s material
    def lighting_batch(self, shape, light, points, eyes, normals, in_shadow):
        p = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        e = np.ascontiguousarray(np.asarray(eyes, dtype=np.float64).reshape(-1, 3))
        nn = np.ascontiguousarray(np.asarray(normals, dtype=np.float64).reshape(-1, 3))
        sh = np.ascontiguousarray(np.asarray(in_shadow, dtype=np.uint8).reshape(-1))
        out = np.zeros_like(p)
        _chk(lib.rt_lighting(self.scene(), self.node_of(shape), int(light), len(p), _dp(p), _dp(e), _dp(nn), sh.ctypes.data_as(c_bp), _dp(out)))
        return out

    # World::is_shadowed (src/world.rs:92-107)
    def is_shadowed_batch(self, points):
        p = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        out = np.zeros((len(p), max(len(self.lights), 1)), dtype=np.uint8)
        _chk(lib.rt_is_shadowed(self.scene(), len(p), _dp(p), out.ctypes.data_as(c_bp)))
        return out[:, :len(self.lights)].astype(bool)

    def is_shadowed(self, point):
        return list(self.is_shadowed_batch([point])[0])


class Camera:
    """struct Camera (src/camera.rs:10-35)."""

    def __init__(self, hsize, vsize, field_of_view):
        self.hsize, self.vsize, self.field_of_view = int(hsize), int(vsize), float(field_of_view)
        self.transform = Matrix()
        d = self.desc()
        self.half_width, self.half_height, self.pixel_size = d.half_width, d.half_height, d.pixel_size

    def desc(self):
        d = CameraDesc()
        _chk(lib.rth_camera_desc(self.hsize, self.vsize, self.field_of_view, _dp(self.transform.m), C.byref(d)))
        return d

    def ray_for_pixel(self, px, py):
        """Camera::ray_for_pixel (src/camera.rs:37-60) evaluated by the device ray-generation kernel."""
        o, d = self.rays_for_pixels([px], [py])
        return tuple(o[0]), tuple(d[0])

    def rays_for_pixels(self, px, py):
        px = np.ascontiguousarray(np.asarray(px, dtype=np.int32))
        py = np.ascontiguousarray(np.asarray(py, dtype=np.int32))
        o = np.zeros((len(px), 3))
        d = np.zeros((len(px), 3))
        cd = self.desc()
        _chk(lib.rt_ray_for_pixel(C.byref(cd), len(px), px.ctypes.data_as(c_ip), py.ctypes.data_as(c_ip), _dp(o), _dp(d)))
        return o, d

    def render(self, world, depth=DEFAULT_RECURSION, want_counters=False, rank=0, n_ranks=1, want_stats=True):
        """Camera::render (src/camera.rs:62-74) on the GPU.  want_stats=False passes no rt_stats, which lets the
        library enqueue the whole frame without a host round trip per recursion level."""
        cd = self.desc()
        opts = RenderOpts(max_depth=depth, rank=rank, n_ranks=n_ranks, want_counters=int(want_counters), stream=None)
        rgb = np.zeros((self.vsize, self.hsize, 3))
        ids = np.zeros((self.vsize, self.hsize), dtype=np.int32)
        flags = np.zeros((self.vsize, self.hsize), dtype=np.uint8)
        st = Stats()
        _chk(lib.rt_render(world.scene(), C.byref(cd), C.byref(opts), _dp(rgb), ids.ctypes.data_as(c_ip),
                           flags.ctypes.data_as(c_bp), C.byref(st) if want_stats else None))
        return Image(rgb, ids, flags, st)


class MultiScene:
    """A scene replicated on several GPUs of one node (rt_scene_create_multi); render() is Camera::render with the image
    tile-sharded over them and gathered to the first device (rt_render_multi)."""

    def __init__(self, world, devices, **scene_opts):
        devs = np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
        self.h = C.c_void_p()
        self.n = len(devs)
        if scene_opts:  # explicit switches (rt_scene_opts) instead of the defaults
            so = SceneOpts(**scene_opts)
            _chk(lib.rt_scene_create_opts(world.flatten(), len(devs), devs.ctypes.data_as(c_ip), C.byref(so), C.byref(self.h)))
        else:
            _chk(lib.rt_scene_create_multi(world.flatten(), len(devs), devs.ctypes.data_as(c_ip), C.byref(self.h)))

    def close(self):
        if self.h:
            lib.rt_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def gather_kind(self):
        return int(lib.rt_scene_gather_kind(self.h))

    def render(self, camera, depth=DEFAULT_RECURSION, want_rgb=True, want_quantised=False):
        cd = camera.desc()
        opts = RenderOpts(max_depth=depth, rank=0, n_ranks=1, want_counters=0, stream=None)
        rgb = np.zeros((camera.vsize, camera.hsize, 3)) if want_rgb else None
        q = np.zeros((camera.vsize, camera.hsize, 3), dtype=np.uint8) if want_quantised else None
        ids = np.zeros((camera.vsize, camera.hsize), dtype=np.int32)
        flags = np.zeros((camera.vsize, camera.hsize), dtype=np.uint8)
        stats = (Stats * self.n)()
        _chk(lib.rt_render_multi(self.h, C.byref(cd), C.byref(opts), _dp(rgb) if want_rgb else None, ids.ctypes.data_as(c_ip),
                                 flags.ctypes.data_as(c_bp), q.ctypes.data_as(c_bp) if want_quantised else None, stats))
        img = Image(rgb, ids, flags, list(stats))
        img.q = q
        return img


def canvas_to_ppm_device(scene, d_rgb_frame, hsize, vsize, d_quantised=None):
    """Canvas::canvas_to_ppm (src/canvas.rs:33-98) encoded on the GPU from a frame that lives in HBM.
    d_rgb_frame / d_quantised: device pointers (ints).  Returns the PPM text as bytes."""
    n = lib.rt_canvas_to_ppm(scene, d_rgb_frame, hsize, vsize, d_quantised, None, 0)
    if n < 0:
        raise RtError(int(n), last_error())
    buf = C.create_string_buffer(int(n))
    m = lib.rt_canvas_to_ppm(scene, d_rgb_frame, hsize, vsize, d_quantised, buf, n)
    if m < 0:
        raise RtError(int(m), last_error())
    return buf.raw[:int(m)]


def check_exports():
    """Every symbol the headers declare must be exported by the library."""
    missing = [n for n in ABI_SYMBOLS + HOST_SYMBOLS if not hasattr(lib, n)]
    if missing:
        raise ImportError("libraytracer_amd.so lacks: " + ", ".join(missing))


check_exports()


def shard_pixels(camera_desc, rank, n_ranks):
    """Host mirror of the device slot -> pixel map (slot_to_pixel in rt_kernels.hip.h): shard-local slot i covers
    pixel (px[i], py[i]) when valid[i].  8x8-pixel tiles of the (hsize-1)x(vsize-1) domain dealt round-robin."""
    w, h = camera_desc.hsize - 1, camera_desc.vsize - 1
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    n = int(lib.rt_shard_pixel_count(C.byref(camera_desc), rank, n_ranks))
    slot = np.arange(n, dtype=np.int64)
    tile = (slot >> 6) * n_ranks + rank
    lane = slot & 63
    ty, tx = tile // max(tiles_x, 1), tile % max(tiles_x, 1)
    px, py = tx * 8 + (lane & 7), ty * 8 + (lane >> 3)
    valid = (ty < tiles_y) & (px < w) & (py < h)
    return px.astype(np.int64), py.astype(np.int64), valid


def shard_pipeline(n_frames, depth, n_buffers):
    """The order of steps of a one-process-per-GPU driver that renders frame after frame with `depth` of its rank's shards in
    flight (rt_render_shard_begin / rt_render_end) and `n_buffers` shard buffers used in turn; yields (step, frame, buffer):
      "end"    rt_render_end: frame's shard is complete in its buffer
      "begin"  rt_render_shard_begin of frame's shard into buffer — after waiting for the gather that last read that buffer
               (frame - n_buffers, posted depth steps after its begin: n_buffers > depth makes sure it has been posted)
      "post"   enqueue the gather (and on the root the scatter) of frame's shard; after the begin that follows its end, so that
               the render lane that has just come free is busy again while the host enqueues the collective
    Every rank walks the same sequence, so the collectives of all ranks match (bench.py's N > 1 path; tests/test_dist_gloo.py)."""
    if depth < 1 or n_buffers <= depth:
        raise ValueError("need depth >= 1 and more shard buffers than shards in flight")
    for k in range(n_frames):
        if k >= depth:
            yield "end", k - depth, (k - depth) % n_buffers
        yield "begin", k, k % n_buffers
        if k >= depth:
            yield "post", k - depth, (k - depth) % n_buffers
    for k in range(max(n_frames - depth, 0), n_frames):
        yield "end", k, k % n_buffers
        yield "post", k, k % n_buffers


def assemble_host(camera_desc, n_ranks, shards):
    """Host mirror of rt_assemble for tests of the gather plumbing: shards[r] is rank r's [count_r, C] array."""
    ch = shards[0].shape[1]
    frame = np.zeros((camera_desc.vsize, camera_desc.hsize, ch), dtype=shards[0].dtype)
    for r in range(n_ranks):
        px, py, valid = shard_pixels(camera_desc, r, n_ranks)
        frame[py[valid], px[valid]] = shards[r][: len(px)][valid]
    return frame
