// rt_general.hip.h — the 4-component path: World::color_at for scenes and cameras the w-free kernels cannot represent.
//
// The kernels of rt_kernels.hip.h drop the w of every tuple, which is exact as long as every inverse matrix is affine
// (w stays 1 for points and 0 for vectors).  Two things a user of the reference can build break that:
//   * a camera whose transform has no inverse: Camera::ray_for_pixel then returns Ray::new(ORIGIN, ORIGIN)
//     (src/camera.rs:44-46) — a POINT as direction, w = 1, for every pixel;
//   * Matrix::new(4, values) with a last row other than 0,0,0,1 as a shape, pattern or camera transform: points leave
//     Matrix * Tuple (src/matrix.rs:171-200) with w != 1, and Tuple's dot / magnitude / normalize / sub / add
//     (src/tuple.rs:23-130) all carry the fourth component along.
// Nothing in the reference rejects either, so the library renders them too: the scene (or the one frame, for a camera)
// takes this path instead of an error.  It is the reference's algorithm with nothing taken away — four components,
// full 4x4 inverses, lists materialised and sorted, prepare_computations with its container walk for every hit, one
// recursion of shade_hit per lane — and with no acceleration structure: every ray meets every node, gated only by the
// reference's own Group boxes.  Orders of magnitude slower than the w-free path, and only ever taken by inputs that
// path has to refuse.  Primitive formulas that never read w (plane, cube, cylinder, cone, triangles, Bound) are the
// SAME device functions the fast kernels use; Sphere, the transforms, normals and lighting are restated with w.
//
// Memory: one slab of HBM per lane (lists, the explicit stacks that stand in for the reference's recursion); a lane
// that needs more than its slab holds raises RT_FLAG_LIST_OVERFLOW for its pixel.
#pragma once

#include "rt_kernels.hip.h"

namespace rt {

struct V4 {
    double x, y, z, w;
};
__device__ __forceinline__ V4 mk4(double x, double y, double z, double w) { V4 v = {x, y, z, w}; return v; }
__device__ __forceinline__ V3 xyz(V4 a) { return mk(a.x, a.y, a.z); }
__device__ __forceinline__ V4 vec4(V3 a) { return mk4(a.x, a.y, a.z, 0.); }
__device__ __forceinline__ V4 add4(V4 a, V4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }  // src/tuple.rs:70-79
__device__ __forceinline__ V4 sub4(V4 a, V4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }  // :81-90
__device__ __forceinline__ V4 neg4(V4 a) { return mk4(-a.x, -a.y, -a.z, -a.w); }                            // :92-98
__device__ __forceinline__ V4 mul4(V4 a, double s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }      // :100-109
__device__ __forceinline__ double dot4(V4 a, V4 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z) + (a.w * b.w); }  // :45-47
__device__ __forceinline__ double magnitude4(V4 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w); }    // :32-34
__device__ __forceinline__ V4 normalize4(V4 a) {                                                                      // :36-43
    double m = magnitude4(a);
    return mk4(a.x / m, a.y / m, a.z / m, a.w / m);
}
__device__ __forceinline__ V4 reflect4(V4 v, V4 n) { return sub4(v, mul4(mul4(n, 2.), dot4(v, n))); }  // :54-56

// Matrix * Tuple, src/matrix.rs:171-200: each component accumulated from 0.0, column by column
__device__ __forceinline__ V4 mat4(const double* __restrict__ m, V4 t) {
    double o[4];
    for (int r = 0; r < 4; ++r) {
        double p = 0.0;
        p += m[r * 4 + 0] * t.x;
        p += m[r * 4 + 1] * t.y;
        p += m[r * 4 + 2] * t.z;
        p += m[r * 4 + 3] * t.w;
        o[r] = p;
    }
    return mk4(o[0], o[1], o[2], o[3]);
}
// the same with the transpose of m (normal_to_world multiplies by inverse.transpose(), src/shape.rs:407)
__device__ __forceinline__ V4 mat4_transposed(const double* __restrict__ m, V4 t) {
    double o[4];
    for (int r = 0; r < 4; ++r) {
        double p = 0.0;
        p += m[0 * 4 + r] * t.x;
        p += m[1 * 4 + r] * t.y;
        p += m[2 * 4 + r] * t.z;
        p += m[3 * 4 + r] * t.w;
        o[r] = p;
    }
    return mk4(o[0], o[1], o[2], o[3]);
}

// What this path reads besides DScene: the full inverses and the descriptor's tree tables as they came in.
struct GScene {
    const double* xf16;             // [n_xforms][16] inverse of every transform
    const double* pat16;            // [n_patterns][16]
    const int32_t* node_xform;
    const int32_t* node_end;        // node_subtree_end
    const int32_t* node_ancestor;   // registry chain (src/shape.rs:394-420): first ancestor entry of a leaf, -1 none
    const int32_t* anc_parent;
    const int32_t* anc_xform;
    const int32_t* csg_op;
    char* slab;                     // lane_bytes per lane of the launch
    long long lane_bytes;
    int32_t list_cap, depth_cap, frame_cap;
    unsigned long long* counters;   // G_* tallies
};
enum GCounter : int { G_REFLECT = 0, G_REFRACT, G_SHADOW, G_SHADED, G_COUNT };

// one pending World::color_at call (src/world.rs:81-90) and what shade_hit (:57-79) still needs once its children return
struct GFrame {
    V4 o, d;
    V4 refr_o, refr_d;       // refracted_color's ray (:129-135), made when the hit is prepared
    double surface[3];       // sum over the lights of Material::lighting
    double reflected[3];     // reflected_color's result, already scaled
    double reflective, transparency, n1, n2, cos_i;
    int32_t remaining;
    int32_t stage;           // 0: to be traced; 1: waiting for the reflected child; 2: waiting for the refracted child
    int32_t refracts;        // 1: refracted_color spawns a ray
    int32_t pad;
};

__host__ __device__ constexpr long long g_lane_bytes(int list_cap, int depth_cap, int frame_cap) {
    return (long long)list_cap * (3 * 8 + 2 * 4) + (long long)depth_cap * (8 * 8 + 3 * 4) + (long long)frame_cap * (long long)sizeof(GFrame) + 64;
}

struct GLane {
    double *lt, *lu, *lv;   // the intersection list: t, and the u, v a smooth triangle keeps (src/intersection.rs:8-13)
    double* st_ray;         // [depth][8] the ray in the space OUTSIDE each open group / CSG node
    GFrame* frames;
    int32_t *lk, *cont;     // list keys (node * 4 + push index) and prepare_computations' `containers`
    int32_t *st_end, *st_mark, *st_csg;
    int32_t cap, dcap, fcap;
};
__device__ __forceinline__ GLane g_lane(const GScene& G, long long lane) {
    GLane M;
    char* p = G.slab + lane * G.lane_bytes;
    M.cap = G.list_cap; M.dcap = G.depth_cap; M.fcap = G.frame_cap;
    M.lt = reinterpret_cast<double*>(p); p += (size_t)M.cap * 8;
    M.lu = reinterpret_cast<double*>(p); p += (size_t)M.cap * 8;
    M.lv = reinterpret_cast<double*>(p); p += (size_t)M.cap * 8;
    M.st_ray = reinterpret_cast<double*>(p); p += (size_t)M.dcap * 64;
    M.frames = reinterpret_cast<GFrame*>(p); p += (size_t)M.fcap * sizeof(GFrame);
    M.lk = reinterpret_cast<int32_t*>(p); p += (size_t)M.cap * 4;
    M.cont = reinterpret_cast<int32_t*>(p); p += (size_t)M.cap * 4;
    M.st_end = reinterpret_cast<int32_t*>(p); p += (size_t)M.dcap * 4;
    M.st_mark = reinterpret_cast<int32_t*>(p); p += (size_t)M.dcap * 4;
    M.st_csg = reinterpret_cast<int32_t*>(p);
    return M;
}

// Vec::sort_by(|a, b| a.t.partial_cmp(&b.t).unwrap()) — stable, so equal t keep emission order = ascending key; a NaN t
// is where the reference panics (flag).  Entries [lo, lo + n).
__device__ inline void g_sort(const GLane& M, int lo, int n, unsigned& flags) {
    for (int i = lo; i < lo + n; ++i)
        if (M.lt[i] != M.lt[i]) flags |= RT_FLAG_NAN_T;
    auto less = [&](int a, int b) { return key_less(M.lt[a], M.lk[a], M.lt[b], M.lk[b]); };
    auto swap = [&](int a, int b) {
        double x = M.lt[a]; M.lt[a] = M.lt[b]; M.lt[b] = x;
        x = M.lu[a]; M.lu[a] = M.lu[b]; M.lu[b] = x;
        x = M.lv[a]; M.lv[a] = M.lv[b]; M.lv[b] = x;
        int k = M.lk[a]; M.lk[a] = M.lk[b]; M.lk[b] = k;
    };
    if (n > 48) { heap_sort(lo, n, less, swap); return; }
    for (int i = lo + 1; i < lo + n; ++i)
        for (int j = i; j > lo && less(j, j - 1); --j) swap(j, j - 1);
}

// Sphere::intersect with all four components (src/sphere.rs:43-63)
template <class E>
__device__ __forceinline__ void g_isect_sphere(V4 o, V4 d, E&& e) {
    V4 sphere_to_ray = sub4(o, mk4(0., 0., 0., 1.));
    double a = dot4(d, d);
    double b = 2. * dot4(d, sphere_to_ray);
    double c = dot4(sphere_to_ray, sphere_to_ray) - 1.;
    double discriminant = b * b - 4. * a * c;
    if (discriminant < 0.) return;
    double t1 = (-b - sqrt(discriminant)) / (2. * a);
    double t2 = (-b + sqrt(discriminant)) / (2. * a);
    e(t1, 0, 0., 0., 0);
    e(t2, 1, 0., 0., 0);
}

// CSG::intersect's tail (src/csg.rs:69-73): sort what both operands returned, then filter_intersections (:106-126);
// Shape::includes (src/shape.rs:422-435) is membership of the hit's object id among the left operand's leaves.
__device__ inline int g_csg_close(const DScene& S, const GScene& G, const GLane& M, int csg_node, int mark, int n, unsigned& flags) {
    g_sort(M, mark, n - mark, flags);
    const int left = csg_node + 1, left_end = G.node_end[left];
    const int op = G.csg_op[S.node_param[csg_node]];
    bool inside_left_hit = false, inside_right_hit = false;
    int out = mark;
    for (int i = mark; i < n; ++i) {
        const int id = S.node_object_id[M.lk[i] >> 2];
        bool left_hit = false;
        for (int j = left; j < left_end && !left_hit; ++j) {
            const int k = S.node_kind[j];
            left_hit = k != RT_GROUP && k != RT_CSG && S.node_object_id[j] == id;
        }
        if (csg_allowed(op, left_hit, inside_left_hit, inside_right_hit)) {
            M.lt[out] = M.lt[i]; M.lk[out] = M.lk[i]; M.lu[out] = M.lu[i]; M.lv[out] = M.lv[i];
            ++out;
        }
        if (left_hit) inside_left_hit = !inside_left_hit;
        else inside_right_hit = !inside_right_hit;
    }
    return out;
}

// World::intersect_world (src/world.rs:46-55) over Shape::intersect (src/shape.rs:74-90), Group::intersect
// (src/group.rs:52-65) and CSG::intersect (src/csg.rs:66-74).  Nodes are numbered in depth-first order, so the
// reference's recursion is a walk over ascending node indices with a stack of the open containers; a group's own sort
// is subsumed by the world's (all of them are stable sorts by t of lists in emission order), a CSG node's is not — its
// filter reads the sorted list.  Returns the length of the sorted list left in M.lt / M.lk / M.lu / M.lv.
__device__ inline int g_intersect_world(const DScene& S, const GScene& G, const GLane& M, V4 ro, V4 rd, unsigned& flags) {
    int n = 0, sp = 0, node = 0;
    V4 co = ro, cd = rd;  // the ray in the space the current node's transform maps from
    Work w = {};
    while (true) {
        while (sp > 0 && node >= M.st_end[sp - 1]) {
            --sp;
            if (M.st_csg[sp] >= 0) n = g_csg_close(S, G, M, M.st_csg[sp], M.st_mark[sp], n, flags);
            const double* r = M.st_ray + (size_t)sp * 8;
            co = mk4(r[0], r[1], r[2], r[3]);
            cd = mk4(r[4], r[5], r[6], r[7]);
        }
        if (node >= S.n_nodes) break;
        const double* inv = G.xf16 + (size_t)G.node_xform[node] * 16;
        const V4 lo = mat4(inv, co), ld = mat4(inv, cd);  // Ray::transform (src/ray.rs:25-30) by this node's inverse
        const int kind = S.node_kind[node], prm = S.node_param[node];
        if (kind == RT_GROUP || kind == RT_CSG) {
            bool enter = true;
            if (kind == RT_GROUP) {  // Bound::intersects (src/bound.rs:85-98) reads x, y, z only
                double tmin, tmax;
                enter = slabs(xyz(lo), xyz(ld), S.group_box + (size_t)prm * 6, tmin, tmax);
            }
            if (enter && sp >= M.dcap) { flags |= RT_FLAG_LIST_OVERFLOW; enter = false; }
            if (!enter) { node = G.node_end[node]; continue; }
            M.st_end[sp] = G.node_end[node];
            M.st_mark[sp] = n;
            M.st_csg[sp] = kind == RT_CSG ? node : -1;
            double* r = M.st_ray + (size_t)sp * 8;
            r[0] = co.x; r[1] = co.y; r[2] = co.z; r[3] = co.w; r[4] = cd.x; r[5] = cd.y; r[6] = cd.z; r[7] = cd.w;
            ++sp;
            co = lo; cd = ld;
            ++node;
            continue;
        }
        auto e = [&](double t, int push, double u, double v, int) {
            if (n < M.cap) { M.lt[n] = t; M.lk[n] = node * 4 + push; M.lu[n] = u; M.lv[n] = v; ++n; }
            else flags |= RT_FLAG_LIST_OVERFLOW;
        };
        switch (kind) {
            case RT_SPHERE: g_isect_sphere(lo, ld, e); break;
            case RT_PLANE: isect_plane(xyz(lo), xyz(ld), e); break;
            case RT_CUBE: isect_cube(xyz(lo), xyz(ld), e); break;
            case RT_CYLINDER: isect_cylinder(xyz(lo), xyz(ld), S.conic[2 * prm], S.conic[2 * prm + 1], S.conic_closed[prm] != 0, e); break;
            case RT_CONE: isect_cone(xyz(lo), xyz(ld), S.conic[2 * prm], S.conic[2 * prm + 1], S.conic_closed[prm] != 0, e); break;
            default: {  // Triangle / SmoothTriangle: cross products and dots with vectors (w = 0) only, src/triangle.rs:62-87
                const double* __restrict__ t = S.tri + (size_t)prm * 12;
                isect_triangle<false>(xyz(lo), xyz(ld), mk(t[0], t[1], t[2]), mk(t[3], t[4], t[5]), mk(t[6], t[7], t[8]),
                                      kind == RT_SMOOTH_TRIANGLE ? 1 : 0, w, e);
                break;
            }
        }
        ++node;
    }
    g_sort(M, 0, n, flags);
    return n;
}

__device__ inline int g_hit_index(const GLane& M, int n) {  // Intersection::hit, src/intersection.rs:30-36
    for (int i = 0; i < n; ++i)
        if (M.lt[i] > 0.) return i;
    return -1;
}

// Shape::world_to_object (src/shape.rs:394-404): the outermost ancestor's inverse first
__device__ inline V4 g_world_to_object(const GScene& G, int node, V4 p) {
    int len = 0;
    for (int a = G.node_ancestor[node]; a >= 0; a = G.anc_parent[a]) ++len;
    for (int k = len - 1; k >= 0; --k) {
        int a = G.node_ancestor[node];
        for (int j = 0; j < k; ++j) a = G.anc_parent[a];
        p = mat4(G.xf16 + (size_t)G.anc_xform[a] * 16, p);
    }
    return mat4(G.xf16 + (size_t)G.node_xform[node] * 16, p);
}
__device__ __forceinline__ V4 g_normal_step(const GScene& G, int xform, V4 n) {  // src/shape.rs:407-409
    V4 nn = mat4_transposed(G.xf16 + (size_t)xform * 16, n);
    nn.w = 0.;
    return normalize4(nn);
}
// Shape::normal_at (src/shape.rs:92-108); hu, hv: the u, v SmoothTriangle::normal_at is handed
__device__ inline V4 g_normal_at_uv(const DScene& S, const GScene& G, int node, V4 world_point, double hu, double hv) {
    const V4 p = g_world_to_object(G, node, world_point);
    const int kind = S.node_kind[node], prm = S.node_param[node];
    V4 n;
    if (kind == RT_SPHERE) {
        n = sub4(p, mk4(0., 0., 0., 1.));  // src/sphere.rs:65-67
    } else {
        V3 tri_n = mk(0., 0., 0.);
        double cmin = 0., cmax = 0.;
        if (kind == RT_TRIANGLE) {
            const double* __restrict__ t = S.tri + (size_t)prm * 12;
            tri_n = mk(t[9], t[10], t[11]);
        } else if (kind == RT_CYLINDER || kind == RT_CONE) {
            cmin = S.conic[2 * prm];
            cmax = S.conic[2 * prm + 1];
        }
        n = vec4(local_normal_at(S, kind, prm, xyz(p), hu, hv, tri_n, cmin, cmax));
    }
    n = g_normal_step(G, G.node_xform[node], n);  // normal_to_world, src/shape.rs:406-420: every level normalises
    for (int a = G.node_ancestor[node]; a >= 0; a = G.anc_parent[a]) n = g_normal_step(G, G.anc_xform[a], n);
    return n;
}

__device__ __forceinline__ bool g_shape_eq(const DScene& S, int a, int b) {
    return S.node_eq_class[a] >= 0 && S.node_eq_class[a] == S.node_eq_class[b];
}

// Intersection::prepare_computations (src/intersection.rs:38-95) for entry `h` of the lane's sorted list of n
struct GComps {
    V4 point, eye, normal, over, under, reflect;
    double t, n1, n2, u, v;
    int32_t node;
    bool inside;
};
__device__ inline void g_prepare(const DScene& S, const GScene& G, const GLane& M, V4 ro, V4 rd, int n, int h, unsigned& flags, GComps& c) {
    const double t = M.lt[h];
    const int node = M.lk[h] >> 2;
    c.t = t;
    c.node = node;
    c.point = add4(ro, mul4(rd, t));  // Ray::position, src/ray.rs:21-23
    c.u = M.lu[h];
    c.v = M.lv[h];
    if (S.node_kind[node] == RT_SMOOTH_TRIANGLE) {  // F5: the u, v of intersections[0] (:40)
        if (S.node_kind[M.lk[0] >> 2] == RT_SMOOTH_TRIANGLE) { c.u = M.lu[0]; c.v = M.lv[0]; }
        else flags |= RT_FLAG_REF_PANIC_UV;  // `hit.u.unwrap()` panics in the reference; the hit's own u, v stand in
    }
    c.normal = g_normal_at_uv(S, G, node, c.point, c.u, c.v);
    c.eye = neg4(rd);
    c.inside = dot4(c.normal, c.eye) < 0.;
    if (c.inside) c.normal = neg4(c.normal);
    c.over = add4(c.point, mul4(c.normal, EPS));
    c.under = sub4(c.point, mul4(c.normal, EPS));
    c.reflect = reflect4(rd, c.normal);
    double n1 = 0., n2 = 0.;
    int nc = 0;
    for (int i = 0; i < n; ++i) {  // every intersection, no break, the last match wins (:57-80; F6)
        const int xn = M.lk[i] >> 2;
        const bool is_self = near_eq(M.lt[i], t) && g_shape_eq(S, xn, node);  // Intersection::eq, :15-19
        if (is_self) n1 = nc == 0 ? 1. : S.mat[(size_t)S.node_material[M.cont[nc - 1]] * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
        int found = -1;
        for (int k = 0; k < nc && found < 0; ++k)
            if (g_shape_eq(S, M.cont[k], xn)) found = k;
        if (found >= 0) {
            for (int k = found; k + 1 < nc; ++k) M.cont[k] = M.cont[k + 1];
            --nc;
        } else {
            M.cont[nc++] = xn;
        }
        if (is_self) n2 = nc == 0 ? 1. : S.mat[(size_t)S.node_material[M.cont[nc - 1]] * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
    }
    c.n1 = n1;
    c.n2 = n2;
}

// Material::lighting (src/material.rs:97-129) with Pattern::pattern_at_shape (src/pattern.rs:23-35)
__device__ inline V3 g_lighting(const DScene& S, const GScene& G, int node, int light, V4 pt, V4 eye_vector, V4 normal_vector, bool in_shadow) {
    const int mi = S.node_material[node];
    const double* __restrict__ m = S.mat + (size_t)mi * RT_MAT_STRIDE;
    const int pat = S.mat_pattern[mi];
    V3 real_color = mk(m[RT_MAT_R], m[RT_MAT_G], m[RT_MAT_B]);
    if (pat >= 0) {
        V4 object_point = g_world_to_object(G, node, pt);
        V4 p = mat4(G.pat16 + (size_t)pat * 16, object_point);
        real_color = pattern_at(S, pat, xyz(p));
    }
    const double* __restrict__ l = S.light + (size_t)light * 6;
    V4 light_position = mk4(l[0], l[1], l[2], 1.);
    V3 intensity = mk(l[3], l[4], l[5]);
    V3 effective_color = vhad(real_color, intensity);
    V4 light_vector = normalize4(sub4(light_position, pt));
    V3 ambient = vmul(effective_color, m[RT_MAT_AMBIENT]);
    double light_dot_normal = dot4(light_vector, normal_vector);
    V3 diffuse, specular;
    if (light_dot_normal < 0.) {
        diffuse = mk(0., 0., 0.);
        specular = mk(0., 0., 0.);
    } else {
        diffuse = vmul(vmul(effective_color, m[RT_MAT_DIFFUSE]), light_dot_normal);
        V4 reflect_vector = reflect4(neg4(light_vector), normal_vector);
        double reflect_dot_eye = dot4(reflect_vector, eye_vector);
        if (near_eq(0., reflect_dot_eye) || reflect_dot_eye < 0.) {
            specular = mk(0., 0., 0.);
        } else {
            double factor = pow(reflect_dot_eye, m[RT_MAT_SHININESS]);
            specular = vmul(vmul(intensity, m[RT_MAT_SPECULAR]), factor);
        }
    }
    if (in_shadow) return ambient;
    return vadd(vadd(ambient, diffuse), specular);
}

// World::is_shadowed for one light (src/world.rs:92-107); uses the lane's list
__device__ inline bool g_is_shadowed(const DScene& S, const GScene& G, const GLane& M, V4 pt, int light, unsigned& flags, unsigned& n_shadow) {
    const double* __restrict__ l = S.light + (size_t)light * 6;
    V4 v = sub4(mk4(l[0], l[1], l[2], 1.), pt);
    double distance = magnitude4(v);
    V4 direction = normalize4(v);
    ++n_shadow;
    const int n = g_intersect_world(S, G, M, pt, direction, flags);
    const int h = g_hit_index(M, n);
    return h >= 0 && M.lt[h] < distance && S.node_casts_shadow[M.lk[h] >> 2] != 0;
}

struct GTally {
    unsigned reflect, refract, shadow, shaded;
};

// World::color_at (src/world.rs:81-90) with everything under it, for one ray.  The reference's recursion
// (shade_hit -> reflected_color / refracted_color -> color_at) runs on the lane's explicit frame stack.
__device__ inline void g_color_at(const DScene& S, const GScene& G, const GLane& M, V4 ro, V4 rd, int remaining0, double rgb[3], int& hit_id,
                                  unsigned& flags, GTally& tally) {
    int fp = 0;
    hit_id = 0;
    {
        GFrame& f = M.frames[0];
        f.o = ro; f.d = rd; f.remaining = remaining0; f.stage = 0;
    }
    double ret[3] = {0., 0., 0.};  // the value the innermost finished call returned
    while (fp >= 0) {
        GFrame& f = M.frames[fp];
        if (f.stage == 0) {
            const int n = g_intersect_world(S, G, M, f.o, f.d, flags);
            const int h = g_hit_index(M, n);
            if (h < 0) {  // BLACK
                ret[0] = 0.; ret[1] = 0.; ret[2] = 0.;
                --fp;
                continue;
            }
            ++tally.shaded;
            GComps c;
            g_prepare(S, G, M, f.o, f.d, n, h, flags, c);
            const int node = c.node;
            if (fp == 0) hit_id = S.node_object_id[node];
            const V4 normal_vector = c.normal, eye_vector = c.eye, over_point = c.over, under_point = c.under, reflect_vector = c.reflect;
            const double n1 = c.n1, n2 = c.n2;
            // ---- World::shade_hit (src/world.rs:57-79): the lights (the list is free again from here on)
            const double* __restrict__ m = S.mat + (size_t)S.node_material[node] * RT_MAT_STRIDE;
            V3 surface = mk(0., 0., 0.);
            for (int l = 0; l < S.n_lights; ++l) {
                const bool in_shadow = g_is_shadowed(S, G, M, over_point, l, flags, tally.shadow);
                surface = vadd(surface, g_lighting(S, G, node, l, over_point, eye_vector, normal_vector, in_shadow));
            }
            f.surface[0] = surface.x; f.surface[1] = surface.y; f.surface[2] = surface.z;
            f.reflective = m[RT_MAT_REFLECTIVE];
            f.transparency = m[RT_MAT_TRANSPARENCY];
            f.n1 = n1; f.n2 = n2;
            f.cos_i = dot4(eye_vector, normal_vector);
            // refracted_color's tests and ray (src/world.rs:120-137), decided now while the vectors are at hand
            f.refracts = 0;
            if (!(near_eq(f.transparency, 0.) || f.remaining <= 0)) {
                const double n_ratio = n1 / n2;
                const double cos_i = f.cos_i;
                const double sin2_t = (n_ratio * n_ratio) * (1. - cos_i * cos_i);
                if (!(sin2_t > 1.)) {
                    const double cos_t = sqrt(1. - sin2_t);
                    f.refr_o = under_point;
                    f.refr_d = sub4(mul4(normal_vector, (n_ratio * cos_i - cos_t)), mul4(eye_vector, n_ratio));
                    f.refracts = 1;
                }
            }
            f.reflected[0] = 0.; f.reflected[1] = 0.; f.reflected[2] = 0.;
            f.stage = 1;
            // reflected_color (src/world.rs:109-118)
            if (!(near_eq(f.reflective, 0.) || f.remaining <= 0)) {
                if (fp + 1 >= M.fcap) { flags |= RT_FLAG_LIST_OVERFLOW; }
                else {
                    ++tally.reflect;
                    GFrame& c = M.frames[fp + 1];
                    c.o = over_point; c.d = reflect_vector; c.remaining = f.remaining - 1; c.stage = 0;
                    ++fp;
                    continue;
                }
            }
            ret[0] = 0.; ret[1] = 0.; ret[2] = 0.;
            f.stage = 3;  // no reflected child: fall through to the refraction with BLACK
        }
        if (f.stage == 1 || f.stage == 3) {  // reflected_color returned (`ret`), or was BLACK
            if (f.stage == 1) { f.reflected[0] = ret[0] * f.reflective; f.reflected[1] = ret[1] * f.reflective; f.reflected[2] = ret[2] * f.reflective; }
            f.stage = 2;
            if (f.refracts) {
                if (fp + 1 >= M.fcap) { flags |= RT_FLAG_LIST_OVERFLOW; }
                else {
                    ++tally.refract;
                    GFrame& c = M.frames[fp + 1];
                    c.o = f.refr_o; c.d = f.refr_d; c.remaining = f.remaining - 1; c.stage = 0;
                    ++fp;
                    continue;
                }
            }
            f.refracts = 0;
        }
        // stage 2: refracted_color returned (or was BLACK): the blend of shade_hit, in the reference's operand order
        double refracted[3] = {0., 0., 0.};
        if (f.refracts) { refracted[0] = ret[0] * f.transparency; refracted[1] = ret[1] * f.transparency; refracted[2] = ret[2] * f.transparency; }
        if (f.reflective > 0. && f.transparency > 0.) {
            const double reflectance = schlick_from(f.cos_i, f.n1, f.n2);
            for (int k = 0; k < 3; ++k) ret[k] = (f.surface[k] + f.reflected[k] * reflectance) + refracted[k] * (1. - reflectance);
        } else {
            for (int k = 0; k < 3; ++k) ret[k] = (f.surface[k] + f.reflected[k]) + refracted[k];
        }
        --fp;
    }
    rgb[0] = ret[0]; rgb[1] = ret[1]; rgb[2] = ret[2];
}

// Camera::ray_for_pixel with every component (src/camera.rs:37-60)
struct GCamera {
    double inv[16];
    double half_width, half_height, pixel_size;
    int32_t invertible;
};
__device__ inline void g_ray_for_pixel(const GCamera& c, int px, int py, V4& origin, V4& direction) {
    if (!c.invertible) {  // Ray::new(ORIGIN, ORIGIN), src/camera.rs:44-46
        origin = mk4(0., 0., 0., 1.);
        direction = mk4(0., 0., 0., 1.);
        return;
    }
    double x_offset = ((double)px + 0.5) * c.pixel_size;
    double y_offset = ((double)py + 0.5) * c.pixel_size;
    double world_x = c.half_width - x_offset;
    double world_y = c.half_height - y_offset;
    V4 pixel = mat4(c.inv, mk4(world_x, world_y, -1., 1.));
    origin = mat4(c.inv, mk4(0., 0., 0., 1.));
    direction = normalize4(sub4(pixel, origin));
}

constexpr int GBLOCK = 64;
__device__ __forceinline__ void g_flush(const GScene& G, const GTally& t) {
    if (t.reflect) atomicAdd(G.counters + G_REFLECT, (unsigned long long)t.reflect);
    if (t.refract) atomicAdd(G.counters + G_REFRACT, (unsigned long long)t.refract);
    if (t.shadow) atomicAdd(G.counters + G_SHADOW, (unsigned long long)t.shadow);
    if (t.shaded) atomicAdd(G.counters + G_SHADED, (unsigned long long)t.shaded);
}

// Camera::render for the slots of one shard (slot order, or straight into the row-major frame: k_output's rules).
// `n_traced`: how many slots really get a ray of their own — with a camera that has no inverse every pixel's ray is the
// same, so one lane traces it (n_traced = 1, into `one`) and k_general_fill hands the result to every pixel.
__global__ __launch_bounds__(GBLOCK) void k_general_render(DScene S, GScene G, GCamera cam, CameraParams c, long long n_slots, int depth, double* rgb,
                                                           int32_t* hit_id, uint8_t* flags_out, int frame) {
    const long long lane = (long long)blockIdx.x * GBLOCK + threadIdx.x, lanes = (long long)gridDim.x * GBLOCK;
    const GLane M = g_lane(G, lane);
    GTally tally = {};
    for (long long slot = lane; slot < n_slots; slot += lanes) {
        int px, py;
        if (!slot_to_pixel(c, slot, px, py)) continue;
        V4 o, d;
        g_ray_for_pixel(cam, px, py, o, d);
        double col[3];
        int id;
        unsigned flags = 0;
        g_color_at(S, G, M, o, d, depth, col, id, flags, tally);
        const long long at = frame ? (long long)py * c.hsize + px : slot;
        rgb[3 * at] = col[0]; rgb[3 * at + 1] = col[1]; rgb[3 * at + 2] = col[2];
        if (hit_id) hit_id[at] = id;
        if (flags_out) flags_out[at] = (uint8_t)flags;
    }
    g_flush(G, tally);
}

// the one ray of a camera without an inverse: result (r, g, b, id, flags) into `one`
__global__ void k_general_one(DScene S, GScene G, GCamera cam, int depth, double* one) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const GLane M = g_lane(G, 0);
    GTally tally = {};
    V4 o, d;
    g_ray_for_pixel(cam, 0, 0, o, d);
    double col[3];
    int id;
    unsigned flags = 0;
    g_color_at(S, G, M, o, d, depth, col, id, flags, tally);
    one[0] = col[0]; one[1] = col[1]; one[2] = col[2];
    one[3] = (double)id;
    one[4] = (double)flags;
    g_flush(G, tally);
}
__global__ __launch_bounds__(BLOCK) void k_general_fill(CameraParams c, long long n_slots, const double* __restrict__ one, double* rgb, int32_t* hit_id,
                                                        uint8_t* flags_out, int frame) {
    const long long slot = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= n_slots) return;
    int px, py;
    if (!slot_to_pixel(c, slot, px, py)) return;
    const long long at = frame ? (long long)py * c.hsize + px : slot;
    rgb[3 * at] = one[0]; rgb[3 * at + 1] = one[1]; rgb[3 * at + 2] = one[2];
    if (hit_id) hit_id[at] = (int32_t)one[3];
    if (flags_out) flags_out[at] = (uint8_t)one[4];
}

// rt_color_at: rays given as 3-component origins (points, w = 1) and directions (vectors, w = 0)
__global__ __launch_bounds__(GBLOCK) void k_general_color_at(DScene S, GScene G, long long n, const double* __restrict__ origins,
                                                             const double* __restrict__ dirs, int remaining, double* rgb, int32_t* hit_id, uint8_t* flags_out) {
    const long long lane = (long long)blockIdx.x * GBLOCK + threadIdx.x, lanes = (long long)gridDim.x * GBLOCK;
    const GLane M = g_lane(G, lane);
    GTally tally = {};
    for (long long i = lane; i < n; i += lanes) {
        double col[3];
        int id;
        unsigned flags = 0;
        g_color_at(S, G, M, mk4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 1.), mk4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], 0.),
                   remaining, col, id, flags, tally);
        rgb[3 * i] = col[0]; rgb[3 * i + 1] = col[1]; rgb[3 * i + 2] = col[2];
        hit_id[i] = id;
        flags_out[i] = (uint8_t)flags;
    }
    g_flush(G, tally);
}

// rt_intersect_world: the sorted list of every ray (count = its full length, the first max_hits entries written)
__global__ __launch_bounds__(GBLOCK) void k_general_list(DScene S, GScene G, long long n, const double* __restrict__ origins,
                                                         const double* __restrict__ dirs, int max_hits, int32_t* count, double* t_out, int32_t* id_out,
                                                         uint8_t* flags_out) {
    const long long lane = (long long)blockIdx.x * GBLOCK + threadIdx.x, lanes = (long long)gridDim.x * GBLOCK;
    const GLane M = g_lane(G, lane);
    for (long long i = lane; i < n; i += lanes) {
        unsigned flags = 0;
        const int m = g_intersect_world(S, G, M, mk4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 1.),
                                        mk4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], 0.), flags);
        count[i] = m;
        flags_out[i] = (uint8_t)flags;
        for (int k = 0; k < m && k < max_hits; ++k) {
            t_out[i * max_hits + k] = M.lt[k];
            id_out[i * max_hits + k] = S.node_object_id[M.lk[k] >> 2];
        }
    }
}

// rt_is_shadowed: one answer per (point, light)
__global__ __launch_bounds__(GBLOCK) void k_general_shadow(DScene S, GScene G, long long n, const double* __restrict__ points, uint8_t* out,
                                                           uint8_t* flags_out) {
    const long long lane = (long long)blockIdx.x * GBLOCK + threadIdx.x, lanes = (long long)gridDim.x * GBLOCK;
    const GLane M = g_lane(G, lane);
    for (long long i = lane; i < n; i += lanes) {
        unsigned flags = 0, n_shadow = 0;
        const V4 p = mk4(points[3 * i], points[3 * i + 1], points[3 * i + 2], 1.);
        for (int l = 0; l < S.n_lights; ++l) out[i * S.n_lights + l] = g_is_shadowed(S, G, M, p, l, flags, n_shadow) ? 1 : 0;
        flags_out[i] = (uint8_t)flags;
    }
}

// rt_prepare_computations: every field of Computations for the hit of each ray (x, y, z of the tuples; columns RT_COMP_*)
__global__ __launch_bounds__(GBLOCK) void k_general_comps(DScene S, GScene G, long long n, const double* __restrict__ origins,
                                                          const double* __restrict__ dirs, int32_t* hit_id_out, double* out, uint8_t* flags_out) {
    const long long lane = (long long)blockIdx.x * GBLOCK + threadIdx.x, lanes = (long long)gridDim.x * GBLOCK;
    const GLane M = g_lane(G, lane);
    for (long long i = lane; i < n; i += lanes) {
        unsigned flags = 0;
        const V4 o = mk4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 1.), d = mk4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], 0.);
        const int m = g_intersect_world(S, G, M, o, d, flags);
        const int h = g_hit_index(M, m);
        double* r = out + i * RT_COMP_COUNT;
        for (int k = 0; k < RT_COMP_COUNT; ++k) r[k] = 0.;
        hit_id_out[i] = 0;
        if (h >= 0) {
            GComps c;
            g_prepare(S, G, M, o, d, m, h, flags, c);
            hit_id_out[i] = S.node_object_id[c.node];
            auto put = [&](int at, V4 v) { r[at] = v.x; r[at + 1] = v.y; r[at + 2] = v.z; };
            r[RT_COMP_T] = c.t;
            put(RT_COMP_POINT, c.point); put(RT_COMP_EYE, c.eye); put(RT_COMP_NORMAL, c.normal);
            r[RT_COMP_INSIDE] = c.inside ? 1. : 0.;
            put(RT_COMP_OVER, c.over); put(RT_COMP_UNDER, c.under); put(RT_COMP_REFLECT, c.reflect);
            r[RT_COMP_N1] = c.n1;
            r[RT_COMP_N2] = c.n2;
            r[RT_COMP_SCHLICK] = schlick_from(dot4(c.eye, c.normal), c.n1, c.n2);
            r[RT_COMP_U] = c.u;
            r[RT_COMP_V] = c.v;
        }
        flags_out[i] = (uint8_t)flags;
    }
}

// rt_normal_at: Shape::normal_at of one leaf for a batch of world points (w = 1)
__global__ __launch_bounds__(BLOCK) void k_general_normal_at(DScene S, GScene G, int node, long long n, const double* __restrict__ points,
                                                             const double* __restrict__ uv, double* out) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V4 nv = g_normal_at_uv(S, G, node, mk4(points[3 * i], points[3 * i + 1], points[3 * i + 2], 1.), uv ? uv[2 * i] : 0., uv ? uv[2 * i + 1] : 0.);
    out[3 * i] = nv.x; out[3 * i + 1] = nv.y; out[3 * i + 2] = nv.z;
}

// rt_lighting: Material::lighting with one leaf's material for a batch of (point, eye vector, normal vector, in_shadow)
__global__ __launch_bounds__(BLOCK) void k_general_lighting(DScene S, GScene G, int node, int light, long long n, const double* __restrict__ points,
                                                            const double* __restrict__ eyes, const double* __restrict__ normals,
                                                            const uint8_t* __restrict__ in_shadow, double* out) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const V3 c = g_lighting(S, G, node, light, mk4(points[3 * i], points[3 * i + 1], points[3 * i + 2], 1.),
                            mk4(eyes[3 * i], eyes[3 * i + 1], eyes[3 * i + 2], 0.), mk4(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2], 0.),
                            in_shadow[i] != 0);
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}

}  // namespace rt
