// rt_kernels.hip.h — hand-written gfx950 device code for the render path.  Included once by rt_api.hip.
//
// Arithmetic rules: IEEE f64 everywhere, built with -ffp-contract=off, every
// expression below that restates a reference formula keeps the reference's operand order, sums run left to right,
// normalisation divides four... three times by the magnitude (never multiplies by a reciprocal), f64::max/min
// become fmax/fmin (both ignore a NaN operand).  Tuples drop w on the device: rt_scene_create only accepts affine
// inverses, for which w stays exactly 1 (points) or 0 (vectors) and its terms add exactly 0.0.
//
// Execution model: one ray per lane.  A frame is a wavefront pipeline over recursion levels d = 0..D:
//   k_trace(d)  closest hits of level d + prepare_computations, AND the shadow rays of level d-1 in the same launch
//   k_queue(d)  stream compaction: shadow queue, n1/n2-walk queue, child slots in level d+1
//   k_walk(d)   n1/n2 container walks (transparent hits only)      k_spawn(d)  reflection / refraction rays of level d+1
// then one more k_trace for the shadow rays of the last level, and k_shade(d) from the deepest level up: lighting,
// schlick and shade_hit's blend in the reference's own order (src/world.rs:57-79), so colours match the recursive form
// bit for bit.  Traversal kernels (k_trace, k_walk, k_list, k_shadow_points) are persistent 1024-lane workgroups, one
// per CU, fed 64-ray chunks per wavefront; the rest are plain streaming kernels.  DESIGN.md section 4 has the why and
// the measurements.
#pragma once
#include <hip/hip_runtime.h>

#include "rt_device.h"

namespace rt {

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 v = {x, y, z}; return v; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vneg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 vmul(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 vhad(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
// Tuple::dot (src/tuple.rs:45-47); the w term is exactly 0
__device__ __forceinline__ double dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__device__ __forceinline__ double magnitude(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 normalize(V3 a) {  // src/tuple.rs:36-43
    double m = magnitude(a);
    return mk(a.x / m, a.y / m, a.z / m);
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {  // src/tuple.rs:49-52
    return mk((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return vsub(v, vmul(vmul(n, 2.), dot(v, n))); }  // :54-56
__device__ __forceinline__ bool near_eq(double a, double b) { return fabs(a - b) < EPS; }          // src/lib.rs:56

// Tables every lane of a wavefront reads at the SAME index (the traversal program, its transforms, group boxes)
// are read through the constant address space: the compiler then fetches them with scalar loads into SGPRs
// (one request per wavefront through the scalar cache, ~4x lower latency than a vector load, no VGPRs) instead of
// 64 identical vector loads.  Read-only use of that cache; nothing is ever written through it.
#define RT_CONSTANT __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const RT_CONSTANT T* as_constant(const T* p) { return (const RT_CONSTANT T*)p; }

// Matrix * Tuple (src/matrix.rs:171-200) for an affine matrix given as rows 0..2; point: w = 1, vector: w = 0.
template <class M>
__device__ __forceinline__ V3 xf_point(M m, V3 p) {
    return mk(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
              ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
template <class M>
__device__ __forceinline__ V3 xf_vector(M m, V3 v) {
    return mk((m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
              (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}
// inverse.transpose() * vector: row r of the transpose is column r of the inverse
template <class M>
__device__ __forceinline__ V3 xf_normal(M m, V3 v) {
    return mk((m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
              (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}

// ---------------------------------------------------------------------------------------------------------
// Work counters of the instrumented build (rt_render_opts.want_counters); plain registers otherwise unused.
struct Work {
    unsigned xforms, box, gate, tri, tri_det, tri_u, tri_v, prim[5], bvh_nodes;
};
// A queue slot without a ray (tile padding, a refraction slot whose ray was totally internally reflected) is marked by
// ONE particular NaN in its origin's x: a NaN with a payload no arithmetic produces.  An ordinary NaN there is a ray
// like any other — the reference traces it (a plane only looks at y: it can even be hit) and so does this code.
constexpr unsigned long long DEAD_RAY_BITS = 0x7ff8dead0000beefull;
__device__ __forceinline__ double dead_ray() { return __longlong_as_double((long long)DEAD_RAY_BITS); }
__device__ __forceinline__ bool is_dead_ray(double ox) { return (unsigned long long)__double_as_longlong(ox) == DEAD_RAY_BITS; }

template <bool COUNT>
__device__ __forceinline__ void bump(unsigned& c, unsigned n = 1) {
    if (COUNT) c += n;
}

// ---------------------------------------------------------------------------------------------------------
// Sinks: what a traversal does with each intersection the reference would have put in its list.
// key = node * 4 + push index inside the primitive = the reference's emission order.

// Intersection::hit over the sorted list == minimum (t, key) among t > 0 (src/intersection.rs:30-36), and
// intersections[0] == minimum (t, key) over everything (src/intersection.rs:40).
template <bool NEED_FIRST>
struct ClosestSink {
    double best_t, bu, bv;
    int best_key;
    double first_t, fu, fv;
    int first_key, first_uv;
    unsigned flags;
    double stop_below;  // the search may end as soon as a hit with t below this is known (-inf: never)
    __device__ __forceinline__ void init(double tmax) {
        best_t = tmax; best_key = -1; bu = bv = 0.;
        first_t = INFINITY; first_key = -1; fu = fv = 0.; first_uv = 0;
        flags = 0;
        stop_below = -INFINITY;
    }
    static constexpr bool FAR_FIRST = false;
    // A triangle repeated under several nodes (MeshTri::smooth bits 1.., rt_program.cpp: build_mesh) is listed by the
    // reference once per node, all with the same t: a minimum over (t, key) only ever keeps the lowest node, which is
    // the one the BVH leaf carries, so this sink has no use for the repeats.
    static constexpr bool WANTS_DUPS = false;
    __device__ __forceinline__ bool done() const { return best_t < stop_below; }
    // intersections[0] is the SMALLEST t of the whole line, i.e. the farthest hit behind the origin: once one is known
    // at first_t < 0, nothing with first_t < t < 0 can matter any more (it is neither the first entry nor a hit), so BVH
    // boxes that lie wholly inside that gap are skipped (bvh_traverse)
    static constexpr bool HAS_GAP = NEED_FIRST;
    __device__ __forceinline__ double gap_lo() const { return first_t; }
    __device__ __forceinline__ double lo() const { return NEED_FIRST ? -INFINITY : 0.; }
    __device__ __forceinline__ double lo_mesh() const { return lo(); }
    __device__ __forceinline__ void emit_mesh(double t, int key, double u, double v, int uv) { emit(t, key, u, v, uv); }
    __device__ __forceinline__ double hi() const { return best_t; }
    __device__ __forceinline__ void emit(double t, int key, double u, double v, int uv) {
        if (t != t) flags |= RT_FLAG_NAN_T;
        if (t > 0. && (t < best_t || (t == best_t && (best_key < 0 || key < best_key)))) {
            best_t = t; best_key = key; bu = u; bv = v;
        }
        if (NEED_FIRST && (t < first_t || (t == first_t && (first_key < 0 || key < first_key)))) {
            first_t = t; first_key = key; fu = u; fv = v; first_uv = uv;
        }
    }
};

// Everything with t <= limit or near_eq(t, limit), kept as an unsorted bounded list (sorted by the caller).
struct ListSink {
    static constexpr bool FAR_FIRST = false;
    static constexpr bool HAS_GAP = false;
    __device__ __forceinline__ double gap_lo() const { return 0.; }
    static constexpr bool WANTS_DUPS = true;
    __device__ __forceinline__ bool done() const { return false; }
    __device__ __forceinline__ double lo_mesh() const { return lo(); }
    __device__ __forceinline__ void emit_mesh(double tt, int k, double u, double v, int uv) { emit(tt, k, u, v, uv); }
    double* t;    // [LIST_CAP], private
    int* key;
    int n;        // true count (may exceed cap)
    int cap;
    double limit;
    unsigned flags;
    __device__ __forceinline__ double lo() const { return -INFINITY; }
    __device__ __forceinline__ double hi() const { return limit + 2. * EPS; }
    __device__ __forceinline__ void emit(double tt, int k, double, double, int) {
        if (tt != tt) flags |= RT_FLAG_NAN_T;
        if (tt <= limit || fabs(tt - limit) < EPS) {
            if (n < cap) { t[n] = tt; key[n] = k; } else flags |= RT_FLAG_LIST_OVERFLOW;
            ++n;
        }
    }
};

// The list of the n1/n2 walk (container_walk): ListSink, except that the triangles of a mesh BVH are "passive" when the
// scene allows it (DScene::passive_mesh_tris: every mesh triangle is equal to no other shape of the scene).  A passive
// entry is never the hit (any triangle but the hit node), so in the reference's loop (src/intersection.rs:54-80) it is
// pushed onto `containers`, never found there, never removed: all it can do is BE containers.last() — or make the
// vector non-empty — when an entry of the hit object within EPSILON of the hit's t is processed.  Those entries all
// have t > hit_t - EPSILON, so of the passive entries with t <= hit_t - 2 EPSILON only the LAST one in list order (the
// maximum (t, key)) can ever be observed; it is kept, the others are dropped, and passive entries inside the window
// above that are all kept.  The walk's result is unchanged and the mesh traversal turns from an enumeration of the whole
// line into a pruned search (boxes that end below the best entry so far are skipped, far children first).
struct WalkSink {
    static constexpr bool FAR_FIRST = true;
    static constexpr bool HAS_GAP = false;
    __device__ __forceinline__ double gap_lo() const { return 0.; }
    static constexpr bool WANTS_DUPS = true;
    __device__ __forceinline__ bool done() const { return false; }
    double* t;    // [LIST_CAP], private
    int* key;
    int n;        // true count (may exceed cap)
    int cap;
    double limit;
    unsigned flags;
    double below_t;   // the passive entry kept from below the window
    int below_key;
    int hit_node;
    bool passive;
    __device__ __forceinline__ double lo() const { return -INFINITY; }
    __device__ __forceinline__ double lo_mesh() const { return passive ? below_t : -INFINITY; }
    __device__ __forceinline__ double hi() const { return limit + 2. * EPS; }
    __device__ __forceinline__ void emit(double tt, int k, double, double, int) {
        if (tt != tt) flags |= RT_FLAG_NAN_T;
        if (tt <= limit || fabs(tt - limit) < EPS) {
            if (n < cap) { t[n] = tt; key[n] = k; } else flags |= RT_FLAG_LIST_OVERFLOW;
            ++n;
        }
    }
    __device__ __forceinline__ void emit_mesh(double tt, int k, double u, double v, int uv) {
        if (passive && (k >> 2) != hit_node && tt <= limit - 2. * EPS) {
            if (tt > below_t || (tt == below_t && k > below_key)) { below_t = tt; below_key = k; }
            return;
        }
        emit(tt, k, u, v, uv);
    }
    __device__ __forceinline__ void finish() {  // the kept entry joins the list (its place comes from the sort)
        if (below_key >= 0) {
            if (n < cap) { t[n] = below_t; key[n] = below_key; } else flags |= RT_FLAG_LIST_OVERFLOW;
            ++n;
        }
    }
};

__device__ __forceinline__ bool key_less(double ta, int ka, double tb, int kb) { return ta < tb || (ta == tb && ka < kb); }

// A list of at most SHORT_LIST entries held in REGISTERS (every index is a compile-time constant after unrolling), for
// the n1/n2 walk of the common short lists: see container_walk.
constexpr int SHORT_LIST = 8;
struct ShortList {
    double t[SHORT_LIST];
    int key[SHORT_LIST];
    __device__ __forceinline__ void exchange(int a, int b) {  // compile-time a < b: order the pair by (t, key)
        if (key_less(t[b], key[b], t[a], key[a])) {
            const double x = t[a]; t[a] = t[b]; t[b] = x;
            const int y = key[a]; key[a] = key[b]; key[b] = y;
        }
    }
    __device__ __forceinline__ void sort() {  // 19-comparator network for eight inputs; unused slots are (+inf, INT_MAX)
        exchange(0, 1); exchange(2, 3); exchange(4, 5); exchange(6, 7);
        exchange(0, 2); exchange(1, 3); exchange(4, 6); exchange(5, 7);
        exchange(1, 2); exchange(5, 6);
        exchange(0, 4); exchange(1, 5); exchange(2, 6); exchange(3, 7);
        exchange(2, 4); exchange(3, 5);
        exchange(1, 2); exchange(3, 4); exchange(5, 6);
    }
};

// In-place heapsort by (t, key) for the long lists of the BIG variants (insertion sort is quadratic; the order is total,
// so stability does not matter).  `swap(a, b)` exchanges entries, `less(a, b)` compares them.
template <class Less, class Swap>
__device__ inline void heap_sort(int lo, int n, Less&& less, Swap&& swap) {
    auto sift = [&](int root, int end) {
        while (true) {
            int child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && less(lo + child, lo + child + 1)) ++child;
            if (!less(lo + root, lo + child)) break;
            swap(lo + root, lo + child);
            root = child;
        }
    };
    for (int i = n / 2 - 1; i >= 0; --i) sift(i, n);
    for (int end = n - 1; end > 0; --end) {
        swap(lo, lo + end);
        sift(0, end);
    }
}

// stable sort by t == sort by (t, key): insertion sort on a short private list (heapsort beyond 64 entries)
__device__ inline void sort_list(double* t, int* key, int n) {
    if (n > 64) {
        heap_sort(0, n, [&](int a, int b) { return key_less(t[a], key[a], t[b], key[b]); },
                  [&](int a, int b) { const double x = t[a]; t[a] = t[b]; t[b] = x; const int y = key[a]; key[a] = key[b]; key[b] = y; });
        return;
    }
    for (int i = 1; i < n; ++i) {
        double ti = t[i];
        int ki = key[i];
        int j = i - 1;
        while (j >= 0 && key_less(ti, ki, t[j], key[j])) { t[j + 1] = t[j]; key[j + 1] = key[j]; --j; }
        t[j + 1] = ti;
        key[j + 1] = ki;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Primitive intersections, each a literal restatement; `E` receives (t, push index, u, v, has_uv).

template <class E>
__device__ __forceinline__ void isect_sphere(V3 o, V3 d, E&& e) {  // src/sphere.rs:43-63
    V3 sphere_to_ray = o;  // origin - ORIGIN
    double a = dot(d, d);
    double b = 2. * dot(d, sphere_to_ray);
    double c = dot(sphere_to_ray, sphere_to_ray) - 1.;
    double discriminant = b * b - 4. * a * c;
    if (discriminant < 0.) return;
    double t1 = (-b - sqrt(discriminant)) / (2. * a);
    double t2 = (-b + sqrt(discriminant)) / (2. * a);
    e(t1, 0, 0., 0., 0);
    e(t2, 1, 0., 0., 0);
}

template <class E>
__device__ __forceinline__ void isect_plane(V3 o, V3 d, E&& e) {  // src/plane.rs:44-51
    if (fabs(d.y) < EPS) return;
    e(-o.y / d.y, 0, 0., 0., 0);
}

// src/cube.rs:63-79 == src/bound.rs:100-116
__device__ __forceinline__ void check_axis(double origin, double direction, double minimum, double maximum, double& tmin,
                                           double& tmax) {
    double tmin_numerator = minimum - origin;
    double tmax_numerator = maximum - origin;
    double a, b;
    if (fabs(direction) >= EPS) {
        a = tmin_numerator / direction;
        b = tmax_numerator / direction;
    } else {
        a = tmin_numerator * INFINITY;
        b = tmax_numerator * INFINITY;
    }
    if (a > b) { tmin = b; tmax = a; } else { tmin = a; tmax = b; }
}
__device__ __forceinline__ double max3nan(double a, double b, double c) { return fmax(fmax(fmax(NAN, a), b), c); }
__device__ __forceinline__ double min3nan(double a, double b, double c) { return fmin(fmin(fmin(NAN, a), b), c); }

// the slab test shared by Cube::intersect (src/cube.rs:45-61) and Bound::intersects (src/bound.rs:85-98)
template <class B>
__device__ __forceinline__ bool slabs(V3 o, V3 d, B box, double& tmin, double& tmax) {
    double xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(o.x, d.x, box[0], box[3], xtmin, xtmax);
    check_axis(o.y, d.y, box[1], box[4], ytmin, ytmax);
    check_axis(o.z, d.z, box[2], box[5], ztmin, ztmax);
    tmin = max3nan(xtmin, ytmin, ztmin);
    tmax = min3nan(xtmax, ytmax, ztmax);
    return !(tmin > tmax);
}

template <class E>
__device__ __forceinline__ void isect_cube(V3 o, V3 d, E&& e) {
    const double unit[6] = {-1., -1., -1., 1., 1., 1.};
    double tmin, tmax;
    if (!slabs(o, d, unit, tmin, tmax)) return;
    e(tmin, 0, 0., 0., 0);
    e(tmax, 1, 0., 0., 0);
}

template <class E>
__device__ __forceinline__ void isect_cylinder(V3 o, V3 d, double minimum, double maximum, bool closed, E&& e) {
    // src/cylinder.rs:53-102; push order t0, t1, lower cap, upper cap
    double a = d.x * d.x + d.z * d.z;
    if (!near_eq(a, 0.)) {
        double b = 2. * o.x * d.x + 2. * o.z * d.z;
        double c = o.x * o.x + o.z * o.z - 1.;
        double disc = b * b - 4. * a * c;
        if (disc < 0.) return;  // `return vec![]`: no caps either
        double t0 = (-b - sqrt(disc)) / (2. * a);
        double t1 = (-b + sqrt(disc)) / (2. * a);
        if (t0 > t1) { double s = t0; t0 = t1; t1 = s; }
        double y0 = o.y + t0 * d.y;
        if (minimum < y0 && y0 < maximum) e(t0, 0, 0., 0., 0);
        double y1 = o.y + t1 * d.y;
        if (minimum < y1 && y1 < maximum) e(t1, 1, 0., 0., 0);
    }
    if (!closed || near_eq(d.y, 0.)) return;
    double t_lower = (minimum - o.y) / d.y;
    {
        double x = o.x + t_lower * d.x, z = o.z + t_lower * d.z;
        double result = x * x + z * z;
        if (result < 1. || near_eq(result, 1.)) e(t_lower, 2, 0., 0., 0);
    }
    double t_upper = (maximum - o.y) / d.y;
    {
        double x = o.x + t_upper * d.x, z = o.z + t_upper * d.z;
        double result = x * x + z * z;
        if (result < 1. || near_eq(result, 1.)) e(t_upper, 3, 0., 0., 0);
    }
}

template <class E>
__device__ __forceinline__ void isect_cone(V3 o, V3 d, double minimum, double maximum, bool closed, E&& e) {
    // src/cone.rs:53-108; push order: linear-case t | t0, t1, then lower cap, upper cap
    double a = d.x * d.x - d.y * d.y + d.z * d.z;
    double b = 2. * o.x * d.x - 2. * o.y * d.y + 2. * o.z * d.z;
    double c = o.x * o.x - o.y * o.y + o.z * o.z;
    if (near_eq(a, 0.) && !near_eq(b, 0.)) {
        double t = -c / (2. * b);
        e(t, 0, 0., 0., 0);
    }
    if (!near_eq(a, 0.)) {
        double disc = b * b - 4. * a * c;
        if (disc > 0. || near_eq(disc, 0.)) {
            double t0 = (-b - sqrt(disc)) / (2. * a);
            double t1 = (-b + sqrt(disc)) / (2. * a);
            if (t0 > t1) { double s = t0; t0 = t1; t1 = s; }
            double y0 = o.y + t0 * d.y;
            if (minimum < y0 && y0 < maximum) e(t0, 0, 0., 0., 0);
            double y1 = o.y + t1 * d.y;
            if (minimum < y1 && y1 < maximum) e(t1, 1, 0., 0., 0);
        }
    }
    if (!closed || near_eq(d.y, 0.)) return;
    double t_lower = (minimum - o.y) / d.y;
    {
        double x = o.x + t_lower * d.x, z = o.z + t_lower * d.z;
        double result = x * x + z * z;
        if (result < minimum * minimum || near_eq(result, minimum * minimum)) e(t_lower, 2, 0., 0., 0);
    }
    double t_upper = (maximum - o.y) / d.y;
    {
        double x = o.x + t_upper * d.x, z = o.z + t_upper * d.z;
        double result = x * x + z * z;
        if (result < maximum * maximum || near_eq(result, maximum * maximum)) e(t_upper, 3, 0., 0., 0);
    }
}

// Moller-Trumbore exactly as src/triangle.rs:62-87 / src/smooth_triangle.rs:71-96
template <bool COUNT, class E>
__device__ __forceinline__ void isect_triangle(V3 o, V3 d, V3 point1, V3 edge_vector1, V3 edge_vector2, int smooth,
                                               Work& w, E&& e) {
    bump<COUNT>(w.tri);
    V3 dir_cross_e2 = cross(d, edge_vector2);
    double determinant = dot(edge_vector1, dir_cross_e2);
    if (fabs(determinant) < EPS) { bump<COUNT>(w.tri_det); return; }
    double f = 1.0 / determinant;
    V3 p1_to_origin = vsub(o, point1);
    double u = f * dot(p1_to_origin, dir_cross_e2);
    if (u < 0. || u > 1.) { bump<COUNT>(w.tri_u); return; }
    V3 origin_cross_e1 = cross(p1_to_origin, edge_vector1);
    double v = f * dot(d, origin_cross_e1);
    if (v < 0. || (u + v) > 1.) { bump<COUNT>(w.tri_v); return; }
    double t = f * dot(edge_vector2, origin_cross_e1);
    e(t, 0, u, v, smooth);
}

// Two triangles at once, no early exits: the same operations in the same order as isect_triangle, evaluated for both
// triangles side by side and accepted by mask.  Measured on the device: a dependent VALU instruction issues ~20 cycles
// after its producer when the wavefront is alone on its SIMD (the tail of every launch), ~8 with four independent
// chains — so two interleaved, branch-free tests finish sooner than two short-circuited ones, and in a wavefront
// whose lanes disagree every exit path was executed anyway.  Values computed past a failed test are never used.
struct TriHit {
    bool ok;
    double t, u, v;
};
template <bool COUNT>
__device__ __forceinline__ TriHit tri_eval(V3 o, V3 d, V3 point1, V3 edge_vector1, V3 edge_vector2, bool enabled, Work& w) {
    V3 dir_cross_e2 = cross(d, edge_vector2);
    double determinant = dot(edge_vector1, dir_cross_e2);
    double f = 1.0 / determinant;
    V3 p1_to_origin = vsub(o, point1);
    double u = f * dot(p1_to_origin, dir_cross_e2);
    V3 origin_cross_e1 = cross(p1_to_origin, edge_vector1);
    double v = f * dot(d, origin_cross_e1);
    double t = f * dot(edge_vector2, origin_cross_e1);
    const bool det_ok = !(fabs(determinant) < EPS);
    const bool u_ok = !(u < 0. || u > 1.);
    const bool v_ok = !(v < 0. || (u + v) > 1.);
    if (COUNT && enabled) {
        ++w.tri;
        if (!det_ok) ++w.tri_det;
        else if (!u_ok) ++w.tri_u;
        else if (!v_ok) ++w.tri_v;
    }
    TriHit h = {enabled && det_ok && u_ok && v_ok, t, u, v};
    return h;
}

// ---------------------------------------------------------------------------------------------------------
// Acceleration BVH4 over the triangle meshes (non-semantic: only decides which triangles get tested).
// Boxes are padded on the host and stored as outward-rounded f32; the slab test below runs in f64 and is
// deliberately loose (NaN-ignoring min/max, relative slack).  The first `n_lds` nodes (the top levels of every
// mesh, rt_program.h) are read from the workgroup's LDS copy, the rest from L2/HBM.

#define RT_LDS __attribute__((address_space(3)))
struct TravMem {
    const RT_LDS BvhNode* lds_nodes;  // workgroup copy of S.bvh[0 .. n_lds)
    int n_lds;
    RT_LDS int* stack;                // [LDS_STACK][TBLOCK], one column per lane
    int* spill;                       // this lane's column of the HBM spill area, rows spill_lanes apart
    long long spill_lanes;
};
__device__ __forceinline__ void stack_push(const TravMem& M, int sp, int v) {
    if (sp < LDS_STACK) M.stack[sp * TBLOCK + threadIdx.x] = v;
    else M.spill[(long long)(sp - LDS_STACK) * M.spill_lanes] = v;
}
__device__ __forceinline__ int stack_pop(const TravMem& M, int sp) {
    return sp < LDS_STACK ? M.stack[sp * TBLOCK + threadIdx.x] : M.spill[(long long)(sp - LDS_STACK) * M.spill_lanes];
}

__device__ __forceinline__ bool bvh_box(double lox, double loy, double loz, double hix, double hiy, double hiz, V3 o, V3 inv,
                                        double tlo, double thi, double& tnear) {
    double ax = (lox - o.x) * inv.x, bx = (hix - o.x) * inv.x;
    double ay = (loy - o.y) * inv.y, by = (hiy - o.y) * inv.y;
    double az = (loz - o.z) * inv.z, bz = (hiz - o.z) * inv.z;
    double tmin = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
    double tmax = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
    tnear = tmin;
    double slack = 1e-9 * (fabs(tmin) + fabs(tmax));
    if (!(slack == slack) || slack > 1e300) slack = 0.;  // inf - inf bookkeeping: infinities compare fine on their own
    return !(tmin > tmax + slack) && !(tmax < tlo - slack) && !(tmin > thi + slack);
}

// 1/d for the slab tests only (never for a reference formula): v_rcp_f64 plus one Newton step, ~1e-16 relative,
// seven orders of magnitude inside the slab test's own slack; 1/0 stays +-inf and 1/inf stays 0.
__device__ __forceinline__ double slab_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double r2 = __builtin_fma(__builtin_fma(-d, r, 1.), r, r);
    return r2 == r2 ? r2 : r;
}

// f32 twin of the slab test for BVH nodes (never for the mesh box, never for a reference formula).  Measured: a
// wavefront issues an f64 VALU instruction every 8-10 cycles against 4 for f32, and a node visit was ~75 % box
// arithmetic.  Conservative by an explicit error budget, not by luck: with o_f = fl32(o), inv_f = fl32(1/d) and the
// stored f32 planes p (already rounded outward), a_f = fl32(fl32(p - o_f) * inv_f) differs from the exact
// (p - o) / d by at most 3 * 2^-24 |a| + 2^-24 |o / d|.  The test widens the [tmin, tmax] interval by
// 2^-20 (|tmin| + |tmax|) + 2^-21 max_axis |o_f * inv_f| — four times that budget on each side.  A ray whose f32
// origin, reciprocal or budget is not finite takes the f64 test instead.
// Round 3: one instruction per plane instead of two.  With q = fl32(o_f * inv_f) kept per axis, a plane distance is
// fma(p, inv_f, -q) — a single rounding of p * inv_f - q, so its error against the exact (p - o) / d is at most
// 2^-24 (|a| + |p / d| + |o / d|) <= 2 * 2^-24 |a| + 2 * 2^-24 |o / d| (rounding of the result, of inv_f, of q): inside the
// same budget (the widening below is 16 * 2^-24 |a| + 8 * 2^-24 |o / d|).  An explicit fmaf in a conservative f32
// pre-test changes no reference arithmetic (-ffp-contract=off still keeps every f64 formula unfused).
struct RayF {
    float qx, qy, qz, ix, iy, iz, budget;
    bool ok;
};
__device__ __forceinline__ RayF make_rayf(V3 o, V3 inv) {
    RayF r;
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
    r.ix = (float)inv.x; r.iy = (float)inv.y; r.iz = (float)inv.z;
    r.qx = ox * r.ix; r.qy = oy * r.iy; r.qz = oz * r.iz;
    const float e = fmaxf(fmaxf(fabsf(r.qx), fabsf(r.qy)), fabsf(r.qz));
    r.budget = e * 4.76837158203125e-07f;  // 2^-21
    r.ok = fabsf(ox) < 3e38f && fabsf(oy) < 3e38f && fabsf(oz) < 3e38f && fabsf(r.ix) < 3e38f && fabsf(r.iy) < 3e38f &&
           fabsf(r.iz) < 3e38f && r.budget < 3e38f;  // NaN fails every comparison (budget < 3e38 also says every q is finite)
    return r;
}
template <bool GAP = false>
__device__ __forceinline__ bool bvh_box_f32(float lox, float loy, float loz, float hix, float hiy, float hiz, const RayF& r, float tlo,
                                            float thi, float& tnear, float gap_lo = 0.f) {
    const float ax = __builtin_fmaf(lox, r.ix, -r.qx), bx = __builtin_fmaf(hix, r.ix, -r.qx);
    const float ay = __builtin_fmaf(loy, r.iy, -r.qy), by = __builtin_fmaf(hiy, r.iy, -r.qy);
    const float az = __builtin_fmaf(loz, r.iz, -r.qz), bz = __builtin_fmaf(hiz, r.iz, -r.qz);
    const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    float slack = 9.5367431640625e-07f * (fabsf(tmin) + fabsf(tmax)) + r.budget;  // 2^-20
    tnear = tmin - slack;
    // a product that overflowed f32 is not an exact infinity: no culling on it
    if (!(slack < 3e38f)) return true;
    // gap_lo < 0: the box is also skipped when it lies wholly inside (gap_lo, 0) — see ClosestSink::gap_lo
    if (GAP && tmin - slack > gap_lo && tmax + slack < 0.f) return false;
    return !(tmin > tmax + slack) && !(tmax + slack < tlo) && !(tmin - slack > thi);
}

struct TriData {
    V3 p1, e1, e2;
    int node, smooth;
};
// The compiler sinks loads into the branch that first uses them; for a triangle test that meant three dependent
// memory round trips per triangle (edges -> det test -> p1 -> u test ... -> node id), ~8000 cycles per 4-triangle
// leaf.  pin() makes a loaded value count as used where it stands, so a leaf's loads are all issued together.
__device__ __forceinline__ void pin(double& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(int& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(V3& a) { pin(a.x); pin(a.y); pin(a.z); }
__device__ __forceinline__ TriData load_tri(const MeshTri* __restrict__ t) {
    TriData r;
    r.p1 = mk(t->p1[0], t->p1[1], t->p1[2]);
    r.e1 = mk(t->e1[0], t->e1[1], t->e1[2]);
    r.e2 = mk(t->e2[0], t->e2[1], t->e2[2]);
    r.node = t->node;
    r.smooth = t->smooth;
    return r;
}

__device__ __forceinline__ void cswap(float& ta, int& ca, float& tb, int& cb) {
    if (tb < ta) { float t = ta; ta = tb; tb = t; int c = ca; ca = cb; cb = c; }
}

template <bool COUNT, class Sink, class E>
__device__ __forceinline__ void bvh_traverse(const DScene& S, const TravMem& M, int root, int node_base, V3 o, V3 d, V3 inv, const Sink& sink,
                                             bool full_line, Work& w, int sp0, E&& e) {
    const RayF rf = make_rayf(o, inv);
    // A launch lasts as long as its slowest wavefront, and the slow ones are in here: let them issue first.
    __builtin_amdgcn_s_setprio(3);
    int sp = sp0;  // entries below sp0 belong to the top-level tree's walk (tlas_next), suspended while this mesh is traversed
    int cur = root;
    while (true) {
        if (cur >= 0) {
            // ---- inner node: one 128-byte record decides four children
            float lo[3][4], hi[3][4];
            int ch[4];
            if (cur < M.n_lds) {
                // LDS copy is column-major (16-byte column k of node i at [k][i]): lanes reading the same column of
                // different nodes spread over the banks instead of all landing on the banks a 128-byte stride shares
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const RT_LDS f32x4* __restrict__ col = (const RT_LDS f32x4*)M.lds_nodes;
                const int N = M.n_lds;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const f32x4 l = col[a * N + cur], h = col[(3 + a) * N + cur];
                    lo[a][0] = l.x; lo[a][1] = l.y; lo[a][2] = l.z; lo[a][3] = l.w;
                    hi[a][0] = h.x; hi[a][1] = h.y; hi[a][2] = h.z; hi[a][3] = h.w;
                }
                const i32x4 c4 = ((const RT_LDS i32x4*)M.lds_nodes)[6 * N + cur];
                ch[0] = c4.x; ch[1] = c4.y; ch[2] = c4.z; ch[3] = c4.w;
            } else {
                const BvhNode* __restrict__ n = S.bvh + cur;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) { lo[a][c] = n->lo[a][c]; hi[a][c] = n->hi[a][c]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) ch[c] = n->child[c];
            }
            bump<COUNT>(w.bvh_nodes);
            bump<COUNT>(w.box, 4);
            const double tlo = full_line ? -INFINITY : sink.lo_mesh();
            const double thi = sink.hi();
            float tn[4];
            if (rf.ok) {
                const float tlo_f = __double2float_rd(tlo), thi_f = __double2float_ru(thi);
                // rounded up: the gap only ever shrinks (no gap when the sink has none, inside CSG, or nothing behind yet)
                const float gap_f = Sink::HAS_GAP && !full_line ? fminf(__double2float_ru(sink.gap_lo()), 0.f) : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {  // all four tests run unconditionally: no branch per child
                    float t;
                    bool hit = bvh_box_f32<Sink::HAS_GAP>(lo[0][c], lo[1][c], lo[2][c], hi[0][c], hi[1][c], hi[2][c], rf, tlo_f, thi_f, t, gap_f) &
                               (ch[c] != INT32_MIN);
                    tn[c] = hit ? (t == t ? t : -INFINITY) : INFINITY;
                    if (!hit) ch[c] = INT32_MIN;
                }
            } else {
                // (rays whose f32 image is not finite are rare: the f64 reciprocals are made again here instead of being kept
                // in six registers through every step of every traversal)
                const V3 inv64 = mk(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z));
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double t;
                    bool hit = bvh_box((double)lo[0][c], (double)lo[1][c], (double)lo[2][c], (double)hi[0][c], (double)hi[1][c],
                                       (double)hi[2][c], o, inv64, tlo, thi, t) & (ch[c] != INT32_MIN);
                    tn[c] = hit ? (t == t ? (float)t : -INFINITY) : INFINITY;
                    if (!hit) ch[c] = INT32_MIN;
                }
            }
            // nearest first (order only matters for how soon the closest-hit bound tightens); a search for the LAST entry
            // below a bound (WalkSink) wants the far children first
            if (Sink::FAR_FIRST) {
#pragma unroll
                for (int c = 0; c < 4; ++c) tn[c] = ch[c] != INT32_MIN ? -tn[c] : INFINITY;
            }
            cswap(tn[0], ch[0], tn[1], ch[1]);
            cswap(tn[2], ch[2], tn[3], ch[3]);
            cswap(tn[0], ch[0], tn[2], ch[2]);
            cswap(tn[1], ch[1], tn[3], ch[3]);
            cswap(tn[1], ch[1], tn[2], ch[2]);
            if (ch[3] != INT32_MIN) stack_push(M, sp++, ch[3]);
            if (ch[2] != INT32_MIN) stack_push(M, sp++, ch[2]);
            if (ch[1] != INT32_MIN) stack_push(M, sp++, ch[1]);
            if (ch[0] != INT32_MIN) { cur = ch[0]; continue; }
        } else {
            // ---- leaf: at most BVH_LEAF_MAX = 4 triangles
            const int code = ~cur;
            const int first = code >> 2, count = (code & 3) + 1;
            for (int k = 0; k < count; k += 2) {  // two triangles per batch of loads
                const bool two = k + 1 < count;
                TriData ta = load_tri(S.mesh_tris + first + k);
                TriData tb = load_tri(S.mesh_tris + first + k + (two ? 1 : 0));
                pin(ta.p1); pin(ta.e1); pin(ta.e2); pin(ta.node); pin(ta.smooth);
                pin(tb.p1); pin(tb.e1); pin(tb.e2); pin(tb.node); pin(tb.smooth);
                const TriHit ha = tri_eval<COUNT>(o, d, ta.p1, ta.e1, ta.e2, true, w);
                const TriHit hb = tri_eval<COUNT>(o, d, tb.p1, tb.e1, tb.e2, two, w);
                // leaf records number their nodes relative to the mesh's group node (identical meshes share one tree)
                if (ha.ok) e(ha.t, (node_base + ta.node) * 4, ha.u, ha.v, ta.smooth & 1);
                if (hb.ok) e(hb.t, (node_base + tb.node) * 4, hb.u, hb.v, tb.smooth & 1);
                // a hit on a repeated triangle counts once per node that carries it (rt_program.cpp: build_mesh) — for sinks
                // that keep lists, and inside CSG subtrees, whose filter toggles on every entry
                if ((Sink::WANTS_DUPS || full_line) && ((ha.ok && (ta.smooth >> 1)) || (hb.ok && (tb.smooth >> 1)))) {
                    if (ha.ok && (ta.smooth >> 1)) {
                        const int* __restrict__ dl = S.mesh_dups + (ta.smooth >> 1) - 1;
                        for (int j = 1, c = dl[0]; j <= c; ++j) e(ha.t, (node_base + dl[j]) * 4, ha.u, ha.v, ta.smooth & 1);
                    }
                    if (hb.ok && (tb.smooth >> 1)) {
                        const int* __restrict__ dl = S.mesh_dups + (tb.smooth >> 1) - 1;
                        for (int j = 1, c = dl[0]; j <= c; ++j) e(hb.t, (node_base + dl[j]) * 4, hb.u, hb.v, tb.smooth & 1);
                    }
                }
            }
        }
        if (sp == sp0 || sink.done()) break;
        cur = stack_pop(M, --sp);
    }
    __builtin_amdgcn_s_setprio(0);
}

// The walk of the top-level tree (rt_program.h: Program::tlas_root), one lane = one ray, resumable: returns the item index
// of the next bounded top-level object whose world box the ray meets within what its sink still accepts — nearest
// boxes first — or INT_MAX when the tree is exhausted.  The pending entries stay in the lane's stack column (`tsp` of
// them) while the wavefront works through the object.  Acceleration only: the boxes are the ITEM_CULL boxes (padded,
// rounded outward to f32 and tested with the node test's error budget), a ray that starts beyond the scene's reach
// (`trusted` == false) is handed every object, and the object's own ITEM_CULL tests its f64 box once more — against
// the bound the sink has by then — before anything is intersected.
template <bool COUNT, class Sink>
__device__ __forceinline__ int tlas_next(const DScene& S, const TravMem& M, V3 ro, V3 rd, const Sink& sink, float inflate, int& tsp, Work& w) {
    const V3 inv = mk(slab_rcp(rd.x), slab_rcp(rd.y), slab_rcp(rd.z));
    const RayF rf = make_rayf(ro, inv);
    // rays that start beyond the program's reach test every box GROWN by what the reference's arithmetic can leak from that
    // far away (traverse: `inflate`); a wavefront without such a ray skips the adjustment
    const bool any_inflate = __ballot(inflate > 0.f) != 0;
    while (tsp > 0) {
        const int cur = stack_pop(M, --tsp);
        if (cur < 0) return ~cur;
        float lo[3][4], hi[3][4];
        int ch[4];
        if (cur < M.n_lds) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const RT_LDS f32x4* __restrict__ col = (const RT_LDS f32x4*)M.lds_nodes;  // column-major, see bvh_traverse
            const int N = M.n_lds;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const f32x4 l = col[a * N + cur], h = col[(3 + a) * N + cur];
                lo[a][0] = l.x; lo[a][1] = l.y; lo[a][2] = l.z; lo[a][3] = l.w;
                hi[a][0] = h.x; hi[a][1] = h.y; hi[a][2] = h.z; hi[a][3] = h.w;
            }
            const i32x4 c4 = ((const RT_LDS i32x4*)M.lds_nodes)[6 * N + cur];
            ch[0] = c4.x; ch[1] = c4.y; ch[2] = c4.z; ch[3] = c4.w;
        } else {
            const BvhNode* __restrict__ n = S.bvh + cur;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) { lo[a][c] = n->lo[a][c]; hi[a][c] = n->hi[a][c]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) ch[c] = n->child[c];
        }
        bump<COUNT>(w.bvh_nodes);
        bump<COUNT>(w.box, 4);
        if (any_inflate) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) { lo[a][c] -= inflate; hi[a][c] += inflate; }
        }
        const double tlo = sink.lo(), thi = sink.hi();
        float tn[4];
        if (rf.ok) {
            const float tlo_f = __double2float_rd(tlo), thi_f = __double2float_ru(thi);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t;
                bool hit = bvh_box_f32<false>(lo[0][c], lo[1][c], lo[2][c], hi[0][c], hi[1][c], hi[2][c], rf, tlo_f, thi_f, t) &
                           (ch[c] != INT32_MIN);
                tn[c] = hit ? (t == t ? t : -INFINITY) : INFINITY;
                if (!hit) ch[c] = INT32_MIN;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double t;
                bool hit = bvh_box((double)lo[0][c], (double)lo[1][c], (double)lo[2][c], (double)hi[0][c], (double)hi[1][c], (double)hi[2][c],
                                   ro, inv, tlo, thi, t) & (ch[c] != INT32_MIN);
                tn[c] = hit ? (t == t ? (float)t : -INFINITY) : INFINITY;
                if (!hit) ch[c] = INT32_MIN;
            }
        }
        cswap(tn[0], ch[0], tn[1], ch[1]);
        cswap(tn[2], ch[2], tn[3], ch[3]);
        cswap(tn[0], ch[0], tn[2], ch[2]);
        cswap(tn[1], ch[1], tn[3], ch[3]);
        cswap(tn[1], ch[1], tn[2], ch[2]);
        if (ch[3] != INT32_MIN) stack_push(M, tsp++, ch[3]);
        if (ch[2] != INT32_MIN) stack_push(M, tsp++, ch[2]);
        if (ch[1] != INT32_MIN) stack_push(M, tsp++, ch[1]);
        if (ch[0] != INT32_MIN) stack_push(M, tsp++, ch[0]);  // popped next: the nearest
    }
    return 0x7fffffff;
}

// ---------------------------------------------------------------------------------------------------------
// Scene traversal == World::intersect_world (src/world.rs:46-55) without materialising the list: every
// intersection the reference would list is handed to `sink` with its (t, key).

// The pending intersections of the CSG subtrees a ray is inside of.  Ordinary kernels keep them in per-lane private
// arrays of CSG_CAP entries; a ray that needs more says so (RT_FLAG_LIST_OVERFLOW), and the host traces it again with
// the BIG kernel variants (scene class bit SC_BIG), whose lists live in an HBM arena with run-time capacities
// (DScene::arena*, one slab per lane of the — small — grid those variants are launched with).
template <bool BIG>
struct CsgScratch;
template <>
struct CsgScratch<false> {
    double t[CSG_CAP], u[CSG_CAP], v[CSG_CAP];
    int key[CSG_CAP];
    int seg[CSG_DEPTH];
    int n, depth;
    unsigned flags;
    __device__ __forceinline__ void bind(const DScene&) {}
    __device__ __forceinline__ int cap() const { return CSG_CAP; }
    __device__ __forceinline__ int max_depth() const { return CSG_DEPTH; }
};
template <>
struct CsgScratch<true> {
    double *t, *u, *v;
    int* key;
    int* seg;  // one mark per open CSG node, in the slab as well: CSG trees may nest as deep as the slab holds entries
    int n, depth;
    unsigned flags;
    int cap_;
    __device__ __forceinline__ void bind(const DScene& S) {
        char* lane = S.arena + ((long long)blockIdx.x * TBLOCK + threadIdx.x) * S.arena_lane_bytes;
        cap_ = S.arena_csg_cap;
        t = reinterpret_cast<double*>(lane);
        u = t + cap_;
        v = u + cap_;
        key = reinterpret_cast<int*>(v + cap_);
        seg = key + cap_;
    }
    __device__ __forceinline__ int cap() const { return cap_; }
    __device__ __forceinline__ int max_depth() const { return cap_; }
};
// the list part of a lane's arena slab (behind the CSG part): t[list_cap], key[list_cap], containers[list_cap]
__device__ __forceinline__ char* arena_list(const DScene& S) {
    return S.arena + ((long long)blockIdx.x * TBLOCK + threadIdx.x) * S.arena_lane_bytes + (long long)S.arena_csg_cap * 32;
}



__device__ __forceinline__ bool csg_allowed(int op, bool left_hit, bool inl, bool inr) {  // src/csg.rs:95-104
    switch (op) {
        case RT_CSG_UNION: return (left_hit && !inr) || (!left_hit && !inl);
        case RT_CSG_INTERSECTION: return (left_hit && inr) || (!left_hit && inl);
        case RT_CSG_DIFFERENCE: return (left_hit && !inr) || (!left_hit && inl);
        default: return false;
    }
}

// Minimum of a per-lane int over the wavefront (every lane must be executing).
__device__ __forceinline__ int wave_min_i32(int v) {
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

// The item index `i` is kept wavefront-uniform: a lane whose Group gate fails does not jump, it sleeps until the
// walk reaches the item after that group (`until`).  Items, transform chains, group boxes and loose triangles are
// therefore fetched once per wavefront with scalar loads, and the kind switch is a scalar branch.  When every lane
// sleeps, the walk jumps to the nearest wake-up item.  `alive` = this lane has a ray at all; callers run traverse
// with the whole wavefront executing.
// SC = scene class: bit 0 the scene has CSG nodes, bit 1 its program has ITEM_CULL ranges (code for either is only
// compiled into the kernels of scenes that need it).
constexpr int SC_CSG = 1, SC_CULL = 2, SC_BIG = 4, SC_TLAS = 8;  // SC_TLAS (with SC_CULL): the world has a top-level tree
// `i_begin`, `i_end` (kernels without a top-level tree only; -1 = the program's end): walk that part of the program — the
// n1/n2 walk visits the meshes in a call of their own (container_walk).  A part must not cut through a group, a cull range
// or a CSG subtree (rt_program.cpp: walk_bound_item).
template <int SC, bool COUNT, class Sink>
__device__ __forceinline__ void traverse(const DScene& S, const TravMem& M, V3 ro, V3 rd, Sink& sink, Work& w, bool alive, int i_begin = 0,
                                         int i_end = -1) {
    constexpr bool HAS_CSG = (SC & SC_CSG) != 0, HAS_CULL = (SC & SC_CULL) != 0, BIG = (SC & SC_BIG) != 0;
    constexpr bool HAS_TLAS = (SC & SC_TLAS) != 0 && HAS_CULL;
    CsgScratch<BIG>* cs = nullptr;
    CsgScratch<BIG> cs_storage;
    if (HAS_CSG) {
        cs = &cs_storage;
        cs->bind(S);
        cs->n = 0;
        cs->depth = 0;
        cs->flags = 0;
    }
    auto out = [&](double t, int key, double u, double v, int uv) {
        if (HAS_CSG && cs->depth > 0) {
            if (cs->n < cs->cap()) {
                int i = cs->n++;
                cs->t[i] = t; cs->key[i] = key; cs->u[i] = u; cs->v[i] = v;
                if (uv) cs->key[i] |= (1 << 30);
            } else {
                cs->flags |= RT_FLAG_LIST_OVERFLOW;
            }
        } else {
            sink.emit(t, key, u, v, uv);
        }
    };
    auto out_mesh = [&](double t, int key, double u, double v, int uv) {  // from a mesh BVH leaf
        if (HAS_CSG && cs->depth > 0) out(t, key, u, v, uv);
        else sink.emit_mesh(t, key, u, v, uv);
    };
    // From very far away the reference's quadratics report tangent hits outside the padded bounds of the ITEM_CULL
    // ranges (rt_program.cpp: WBox::reach): a ray that starts beyond the scene's reach is never culled by them.
    // (the direction's length only matters to scenes with a bounded cube: rt_program.cpp, world_box)
    // (the direction's length only matters to scenes with a bounded cube: rt_program.cpp, world_box)
    bool cull_trusted = HAS_CULL && fmax(fmax(fabs(ro.x), fabs(ro.y)), fabs(ro.z)) < S.cull_reach &&
                        (S.cull_min_dir2 == 0. || dot(rd, rd) >= S.cull_min_dir2);
    // Under a top-level tree a ray from beyond the reach is not simply handed everything: from D away (L-infinity, to anything
    // bounded) the reference's quadratics leak by at most far_a D^2 and everything else by far_b D (rt_program.cpp: the bounds
    // behind Program::far_a / far_b), so that ray tests every box grown by (far_a D + far_b) D — a fraction of a unit from
    // thousands of units away.  Only a ray whose direction is too short for a bounded cube, or that is not finite, is left
    // with no culling at all (`far_lane`, below).  (Kernels without a tree keep the plain rule, and their code.)
    double inflate = 0.;
    bool cull_inflated = false;
    float inflate_f = 0.f;
    if (HAS_TLAS && !cull_trusted && (S.cull_min_dir2 == 0. || dot(rd, rd) >= S.cull_min_dir2)) {
        const double dist = fmax(fmax(fabs(ro.x), fabs(ro.y)), fabs(ro.z)) + S.far_extent;
        inflate = (S.far_a * dist + S.far_b) * dist;
        cull_inflated = inflate > 0. && inflate < 1e30;  // (NaN: false)
        inflate_f = cull_inflated ? __double2float_ru(inflate) : 0.f;
    }
    const RT_CONSTANT Item* __restrict__ items = as_constant(S.items);
    const RT_CONSTANT double* __restrict__ xf = as_constant(S.xf);
    const RT_CONSTANT int32_t* __restrict__ pool = as_constant(S.chain_pool);
    const int n_items = !HAS_TLAS && i_end >= 0 ? i_end : S.n_items;
    int i = HAS_TLAS ? 0 : i_begin;
    int until = alive ? i : n_items;
    // Worlds with a top-level tree (SC_TLAS kernels): the first n_linear items are walked by every ray as ever; the bounded
    // objects behind them are ranges [ITEM_CULL, ..., skip) that a ray visits only when its own walk of the tree
    // (tlas_next) names them.  Each lane holds the next object ITS ray has to visit (`cand`).  While the lanes of the
    // wavefront agree, or the object is not a mesh instance, the wavefront goes to the lowest of them with the item
    // index wavefront-uniform as ever (scalar fetches of item and matrices): the lanes that asked for it are awake there,
    // the others sleep through it and keep their candidate for a later round.  When lanes hold DIFFERENT mesh instances
    // (an object that is one ITEM_MESH under its transforms: the teapots of a grid seen by incoherent rays), every lane
    // walks its own instance at the same time, with its own transformed ray, instead of the wavefront taking them in turn.
    const bool tlas = HAS_TLAS && S.tlas_root >= 0;
    int range_end = HAS_TLAS ? S.n_linear : n_items;
    int tsp = 0, cand = -1;
    bool mine = false;  // this lane asked for the range the wavefront is in
    bool swept = false;
    const bool far_lane = tlas && alive && !cull_trusted && !cull_inflated;
    if (tlas && alive) {
        stack_push(M, 0, S.tlas_root);
        tsp = 1;
    }
    while (true) {
        // SC_TLAS kernels have ONE call of the mesh traversal, at the end of the loop body, for whoever set up a job in this
        // round: the lanes a wavefront-uniform ITEM_MESH is meant for, or lanes that each fetched an instance of their own
        // (a second inlined copy of bvh_traverse took these kernels from 43 to 192 spilled registers)
        bool job = false, job_full = false, items_round = true;
        V3 job_o = ro, job_d = rd, job_inv = rd;
        int job_root = 0, job_base = 0;
        if (HAS_TLAS && i >= range_end && tlas && !swept) {
            // Rays that may not trust the boxes (they start beyond the program's reach, or their direction is too short for
            // a bounded cube: cull_trusted) get ONE wavefront-uniform pass over every bounded object, boxes ignored, before the
            // tree is asked anything — handing such a ray its objects one dispatcher round each cost a world of two thousand
            // spheres 2.5 ms per 64 shadow rays that start near the horizon of its floor.
            swept = true;
            if (__ballot(far_lane) != 0) {
                mine = false;
                i = S.n_linear;
                range_end = n_items;
                until = far_lane ? i : n_items;
            }
        }
        if (HAS_TLAS && i >= range_end) {
            if (!tlas) break;
            if (mine) cand = -1;  // that range is done
            mine = false;
            if (!alive || far_lane) cand = 0x7fffffff;
            if (cand != 0x7fffffff && sink.done()) cand = 0x7fffffff;
            if (cand < 0) cand = tlas_next<COUNT>(S, M, ro, rd, sink, inflate_f, tsp, w);
            const bool instance = cand != 0x7fffffff && (cand & TLAS_INSTANCE);
            const int header = instance ? (cand & ~TLAS_INSTANCE) - 1 : cand;  // an instance's ITEM_CULL sits in front of its ITEM_MESH
            const int lowest = wave_min_i32(header);
            if (lowest == 0x7fffffff) break;
            if (__ballot(instance && header != lowest) == 0) {  // the wavefront agrees (or the lowest is no instance): uniform walk
                mine = header == lowest;
                range_end = items[lowest].skip;
                // an instance's ITEM_CULL is skipped: its ITEM_MESH tests the mesh's own, tighter box against the same bounds
                i = __ballot(mine && instance) ? lowest + 1 : lowest;
                until = mine ? i : (cand == 0x7fffffff ? n_items : range_end);
            } else {  // lanes hold different instances: each sets up its own, all are traversed together below
                items_round = false;
                if (instance) {
                    const Item* __restrict__ q = S.items + (cand & ~TLAS_INSTANCE);  // per-lane fetch
                    const int m_type = q->type, m_node = q->node, m_root = q->a, m_box = q->b, m_len = q->chain_len, m_x0 = q->x0, m_x1 = q->x1,
                              m_chain = q->chain;
                    V3 o = ro, d = rd;
                    for (int k = 0; k < m_len; ++k) {  // Ray::transform per level, outermost first (src/shape.rs:75-76)
                        const int x = k == 0 ? m_x0 : k == 1 ? m_x1 : S.chain_pool[m_chain + 1 + k];
                        const double* __restrict__ m = S.xf + (size_t)x * 12;
                        o = xf_point(m, o);
                        d = xf_vector(m, d);
                    }
                    if (COUNT) w.xforms += m_len;
                    if (m_type == ITEM_MESH) {
                        bump<COUNT>(w.box);
                        const double* __restrict__ mb = S.mesh_box + (size_t)m_box * 6;
                        const V3 inv = mk(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z));
                        double tnear;
                        job = bvh_box(mb[0], mb[1], mb[2], mb[3], mb[4], mb[5], o, inv, sink.lo_mesh(), sink.hi(), tnear);
                        job_o = o; job_d = d; job_inv = inv; job_root = m_root; job_base = m_node;
                    } else {
                        // a bounded primitive that is a top-level object of its own (sphere, cube, truncated cylinder): every lane
                        // intersects the one ITS ray reached, here and now — a world of two thousand spheres otherwise has the
                        // wavefront visit them one by one for a lane or two each
                        const int m_kind = q->kind, m_closed = q->op;
                        const double m_p0 = q->p0, m_p1 = q->p1;
                        auto lane_out = [&](double t, int k, double u, double v, int uv) { out(t, m_node * 4 + k, u, v, uv); };
                        if (m_kind == RT_SPHERE) { bump<COUNT>(w.prim[0]); isect_sphere(o, d, lane_out); }
                        else if (m_kind == RT_CUBE) { bump<COUNT>(w.prim[2]); isect_cube(o, d, lane_out); }
                        else { bump<COUNT>(w.prim[3]); isect_cylinder(o, d, m_p0, m_p1, m_closed != 0, lane_out); }
                    }
                    cand = -1;
                }
            }
        }
        if (!HAS_TLAS && i >= n_items) break;
        if (items_round) {
        Item it;  // one 64-byte scalar fetch (member-wise: aggregates do not copy across address spaces)
        {
            const RT_CONSTANT Item* __restrict__ q = items + i;
            it.type = q->type; it.node = q->node; it.skip = q->skip; it.kind = q->kind;
            it.a = q->a; it.b = q->b; it.op = q->op; it.chain_len = q->chain_len;
            it.x0 = q->x0; it.x1 = q->x1; it.chain = q->chain; it.p0 = q->p0; it.p1 = q->p1;
        }
        if (sink.done()) until = n_items;  // an any-hit search that has its answer sleeps through the rest
        const bool on = i >= until;
        ++i;
        if (it.type == ITEM_CSG_BEGIN) {
            if (HAS_CSG && on) {
                if (cs->depth < cs->max_depth()) cs->seg[cs->depth] = cs->n; else cs->flags |= RT_FLAG_LIST_OVERFLOW;
                cs->depth++;
            }
            continue;
        }
        if (it.type == ITEM_CSG_END) {
            if (HAS_CSG && on) {
                cs->depth--;
                int s0 = cs->depth < cs->max_depth() ? cs->seg[cs->depth] : 0;
                // sort the operand lists together (stable by t == by (t, key)), src/csg.rs:70-71
                for (int a = s0; a < cs->n; ++a)
                    if (cs->t[a] != cs->t[a]) cs->flags |= RT_FLAG_NAN_T;
                if (BIG && cs->n - s0 > 48) {
                    heap_sort(s0, cs->n - s0,
                              [&](int a, int b) { return key_less(cs->t[a], cs->key[a] & ~(1 << 30), cs->t[b], cs->key[b] & ~(1 << 30)); },
                              [&](int a, int b) {
                                  const double ta = cs->t[a], ua = cs->u[a], va = cs->v[a];
                                  const int ka = cs->key[a];
                                  cs->t[a] = cs->t[b]; cs->u[a] = cs->u[b]; cs->v[a] = cs->v[b]; cs->key[a] = cs->key[b];
                                  cs->t[b] = ta; cs->u[b] = ua; cs->v[b] = va; cs->key[b] = ka;
                              });
                } else {
                    for (int a = s0 + 1; a < cs->n; ++a) {
                        double ta = cs->t[a], ua = cs->u[a], va = cs->v[a];
                        int ka = cs->key[a];
                        int b = a - 1;
                        while (b >= s0 && key_less(ta, ka & ~(1 << 30), cs->t[b], cs->key[b] & ~(1 << 30))) {
                            cs->t[b + 1] = cs->t[b]; cs->u[b + 1] = cs->u[b]; cs->v[b + 1] = cs->v[b]; cs->key[b + 1] = cs->key[b];
                            --b;
                        }
                        cs->t[b + 1] = ta; cs->u[b + 1] = ua; cs->v[b + 1] = va; cs->key[b + 1] = ka;
                    }
                }
                // filter_intersections, src/csg.rs:106-126
                bool inl = false, inr = false;
                int wr = s0;
                for (int a = s0; a < cs->n; ++a) {
                    int node = (cs->key[a] & ~(1 << 30)) >> 2;
                    // Shape::includes (src/shape.rs:422-435) is membership of the hit's object id among the left operand's
                    // leaves: the node range of the left operand, plus the (normally empty) list of right-operand nodes
                    // that carry an id of the left operand (a shape cloned into both operands keeps its id)
                    bool left_hit = node >= it.a && node < it.b;
                    if (it.chain_len > 0 && !left_hit) {
                        const int* __restrict__ sh = S.chain_pool + it.chain;
                        int lo_i = 0, hi_i = it.chain_len;  // binary search: the list ascends
                        while (lo_i < hi_i) {
                            const int mid = (lo_i + hi_i) >> 1;
                            if (sh[mid] < node) lo_i = mid + 1; else hi_i = mid;
                        }
                        left_hit = lo_i < it.chain_len && sh[lo_i] == node;
                    }
                    if (csg_allowed(it.op, left_hit, inl, inr)) {
                        cs->t[wr] = cs->t[a]; cs->u[wr] = cs->u[a]; cs->v[wr] = cs->v[a]; cs->key[wr] = cs->key[a];
                        ++wr;
                    }
                    if (left_hit) inl = !inl; else inr = !inr;
                }
                cs->n = wr;
                if (cs->depth == 0) {  // outermost CSG done: its survivors join the world list
                    for (int a = 0; a < cs->n; ++a)
                        sink.emit(cs->t[a], cs->key[a] & ~(1 << 30), cs->u[a], cs->v[a], (cs->key[a] >> 30) & 1);
                    cs->n = 0;
                }
            }
            continue;
        }
        if (HAS_CULL && it.type == ITEM_CULL) {  // acceleration: world-space bounds of the top-level objects up to it.skip
            if (on) {
                const RT_CONSTANT double* __restrict__ cb = reinterpret_cast<const RT_CONSTANT double*>(items + (i - 1)) + 2;
                // 1/d is recomputed per range rather than kept: worlds without ranges (the headline scene) must not pay
                // registers for it, and a range that culls saves far more than three reciprocals
                const V3 inv_world = mk(slab_rcp(rd.x), slab_rcp(rd.y), slab_rcp(rd.z));
                double tnear;
                bump<COUNT>(w.box);
                if (HAS_TLAS) {
                    const double g = cull_inflated ? inflate : 0.;
                    if ((cull_trusted || cull_inflated) &&
                        !bvh_box(cb[0] - g, cb[1] - g, cb[2] - g, cb[3] + g, cb[4] + g, cb[5] + g, ro, inv_world, sink.lo(), sink.hi(), tnear))
                        until = it.skip;
                } else {
                    if (cull_trusted && !bvh_box(cb[0], cb[1], cb[2], cb[3], cb[4], cb[5], ro, inv_world, sink.lo(), sink.hi(), tnear)) until = it.skip;
                }
            }
            if (__ballot(i >= until) == 0) i = wave_min_i32(until);  // nobody needs the range: jump, uniformly
            continue;
        }
        // Ray::transform per level, outermost first (src/shape.rs:75-76, src/ray.rs:25-30)
        V3 o = ro, d = rd;
        for (int k = 0; k < it.chain_len; ++k) {
            const int x = k == 0 ? it.x0 : k == 1 ? it.x1 : pool[it.chain + 1 + k];
            const RT_CONSTANT double* __restrict__ m = xf + (size_t)x * 12;
            o = xf_point(m, o);
            d = xf_vector(m, d);
        }
        if (COUNT && on) w.xforms += it.chain_len;
        if (it.type == ITEM_GATE) {  // Group::intersect gate, src/group.rs:52-53
            double tmin, tmax;
            if (on) {
                bump<COUNT>(w.gate);
                if (!slabs(o, d, as_constant(S.group_box) + (size_t)it.a * 6, tmin, tmax)) until = it.skip;
            }
            if (__ballot(i >= until) == 0) i = wave_min_i32(until);  // nobody enters the group: jump, uniformly
            continue;
        }
        if (!on) continue;
        if (it.type == ITEM_MESH) {
            bool full_line = HAS_CSG && cs->depth > 0;  // a CSG filter needs its operands' hits behind the origin too
            // the mesh's own (padded f64) box first: most rays of a frame never come near the mesh
            const RT_CONSTANT double* __restrict__ mb = as_constant(S.mesh_box) + (size_t)it.b * 6;
            const V3 inv = mk(slab_rcp(d.x), slab_rcp(d.y), slab_rcp(d.z));
            double tnear;
            bump<COUNT>(w.box);
            if (bvh_box(mb[0], mb[1], mb[2], mb[3], mb[4], mb[5], o, inv, full_line ? -INFINITY : sink.lo_mesh(), sink.hi(), tnear))
            {
                if (HAS_TLAS) { job = true; job_full = full_line; job_o = o; job_d = d; job_inv = inv; job_root = it.a; job_base = it.node; }
                else bvh_traverse<COUNT>(S, M, it.a, it.node, o, d, inv, sink, full_line, w, tsp, out_mesh);
            }
            if (!HAS_TLAS) continue;
        } else {
        // ITEM_LEAF
        const int node = it.node;
        auto leaf_out = [&](double t, int k, double u, double v, int uv) { out(t, node * 4 + k, u, v, uv); };
        switch (it.kind) {
            case RT_SPHERE: bump<COUNT>(w.prim[0]); isect_sphere(o, d, leaf_out); break;
            case RT_PLANE: bump<COUNT>(w.prim[1]); isect_plane(o, d, leaf_out); break;
            case RT_CUBE: bump<COUNT>(w.prim[2]); isect_cube(o, d, leaf_out); break;
            case RT_CYLINDER: bump<COUNT>(w.prim[3]); isect_cylinder(o, d, it.p0, it.p1, it.op != 0, leaf_out); break;
            case RT_CONE: bump<COUNT>(w.prim[4]); isect_cone(o, d, it.p0, it.p1, it.op != 0, leaf_out); break;
            default: {  // RT_TRIANGLE / RT_SMOOTH_TRIANGLE outside a mesh
                const RT_CONSTANT double* __restrict__ t = as_constant(S.tri) + (size_t)it.a * 12;
                isect_triangle<COUNT>(o, d, mk(t[0], t[1], t[2]), mk(t[3], t[4], t[5]), mk(t[6], t[7], t[8]),
                                      it.kind == RT_SMOOTH_TRIANGLE, w, leaf_out);
                break;
            }
        }
        }
        }  // items_round
        if (HAS_TLAS && job) bvh_traverse<COUNT>(S, M, job_root, job_base, job_o, job_d, job_inv, sink, job_full, w, tsp, out_mesh);
    }
    if (HAS_CSG) sink.flags |= cs->flags;
}

// Chunk feed.  With counters (`cnt` != null) the feed is dynamic and two-level.  The chunk range is cut into
// FEED_PARTS equal partitions, each with its own counter on its own cache line (one hot word costs ~70 ns per
// returning atomic chip-wide; eight lines take that off the critical path).  A WORKGROUP reserves a batch of
// consecutive chunks from one partition with one returning atomic (guided self-scheduling: batches shrink from 64 to
// 8 chunks as the partition drains; a workgroup starts on partition blockIdx % FEED_PARTS and moves on when that one
// is dry), and its wavefronts take chunks out of the batch with an LDS atomic on one packed word,
// start << 24 | size << 16 | taken.  The wavefront that takes the slot just past the batch refills it; the others
// poll the word (s_sleep between polls) until the batch changes.  No barrier: a wavefront stuck on a heavy chunk
// never holds anyone up.  A batch that starts at n_chunks is the end mark, and every wavefront reaches it.
// Without counters: chunks are dealt round-robin over the wavefronts of the grid.  Values are wavefront-uniform.
struct FeedShared {
    unsigned long long word;  // start << 24 | size << 16 | taken
    int part, dry;            // touched only by the wavefront that refills
};
struct Feed {
    unsigned long long* cnt;
    RT_LDS FeedShared* sh;
    long long n_chunks;               // positions handed out (n_real rounded up to a multiple of FEED_ROWS)
    long long n_real, cols;           // chunks that exist; positions per row of the scatter below
    long long next_static, stride_static;

    // one lane: next batch as a packed word, or the end mark
    __device__ __forceinline__ unsigned long long reserve() {
        const long long span = (n_chunks + FEED_PARTS - 1) / FEED_PARTS;
        int part = sh->part, dry = sh->dry;
        while (dry < FEED_PARTS) {
            const long long lo = span * part, hi = lo + span < n_chunks ? lo + span : n_chunks;
            unsigned long long* c = cnt + part * FEED_STRIDE;
            const long long left = (hi - lo) - (long long)__atomic_load_n(c, __ATOMIC_RELAXED);
            if (left > 0) {
                long long size = left * FEED_PARTS / (2ll * gridDim.x);
                size = size < 8 ? 8 : size > 64 ? 64 : size;
                const long long start = lo + (long long)atomicAdd(c, (unsigned long long)size);
                if (start < hi) {
                    if (start + size > hi) size = hi - start;
                    sh->part = part; sh->dry = dry;
                    return ((unsigned long long)start << 24) | ((unsigned long long)size << 16);
                }
            }
            part = (part + 1) % FEED_PARTS;
            ++dry;
        }
        sh->part = part; sh->dry = dry;
        return (unsigned long long)n_chunks << 24;
    }
    // called by every thread of the workgroup before the setup barrier
    __device__ __forceinline__ void init(unsigned long long* counters, long long chunks, RT_LDS FeedShared* shared) {
        cnt = counters;
        sh = shared;
        n_real = chunks;
        cols = (chunks + FEED_ROWS - 1) / FEED_ROWS;
        n_chunks = cols * FEED_ROWS;
        const long long wave = (long long)blockIdx.x * (TBLOCK / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        next_static = wave;
        stride_static = (long long)gridDim.x * (TBLOCK / 64);
        // a launch with at most two chunks per wavefront has nothing to balance: deal statically, no atomics at all
        if (n_real <= 2 * stride_static) cnt = nullptr;
        if (cnt && threadIdx.x == 0) {
            sh->part = int(blockIdx.x % FEED_PARTS);
            sh->dry = 0;
            sh->word = reserve();
        }
    }
    __device__ __forceinline__ static unsigned long long uniform(unsigned long long v) {
        unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    }
    // Expensive rays come in clumps (the tiles that see a mesh, the shadow they cast), and so do the positions a
    // workgroup's batch or a wavefront's static share covers.  Position p therefore stands for chunk
    // (p % FEED_ROWS) * cols + p / FEED_ROWS: consecutive positions are a FEED_ROWS-th of the level apart, every batch
    // samples the whole level, and a clump of heavy chunks is spread over all compute units instead of queueing up
    // on the SIMDs of a few.  (Measured before: wavefront lifetimes within one launch ranged from 130k to over 1M
    // cycles.)  Positions past the last chunk are skipped.
    __device__ __forceinline__ long long next() {  // -1: nothing left
        while (true) {
            const long long p = next_position();
            if (p < 0) return -1;
            const long long c = (p % FEED_ROWS) * cols + p / FEED_ROWS;
            if (c < n_real) return (long long)uniform((unsigned long long)c);  // in scalar registers from here on
        }
    }
    __device__ __forceinline__ long long next_position() {
        if (!cnt) {
            long long c = next_static;
            next_static += stride_static;
            return c < n_chunks ? c : -1;
        }
        const bool leader = (threadIdx.x & 63) == 0;
        while (true) {
            unsigned long long v = 0;
            if (leader) v = __hip_atomic_fetch_add(&sh->word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            v = uniform(v);
            const long long start = (long long)(v >> 24);
            const int size = int((v >> 16) & 0xff), taken = int(v & 0xffff);
            if (start >= n_chunks) return -1;
            if (taken < size) return start + taken;
            if (taken == size) {  // refill
                if (leader) {
                    unsigned long long nv = reserve();
                    __hip_atomic_store(&sh->word, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                continue;
            }
            // somebody else is refilling
            unsigned long long cur;
            do {
                __builtin_amdgcn_s_sleep(2);
                cur = uniform(__hip_atomic_load(&sh->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            } while ((long long)(cur >> 24) == start);
        }
    }
};

// Traversal kernels are persistent: a grid of as many 1024-lane workgroups as fit the chip, each staging the BVH
// top levels into LDS once.  After that every WAVEFRONT works on its own: it pulls chunks of 64 rays from a feed
// until the feed is dry, so a wavefront stuck on a heavy chunk never holds anyone else up.
// Dynamic LDS layout: [n_lds BvhNode][LDS_STACK x TBLOCK int].
__device__ __forceinline__ TravMem trav_setup(const DScene& S, Feed& feed, unsigned long long* feed_counter, long long n_chunks) {
    extern __shared__ __align__(16) unsigned char trav_lds[];
    __shared__ FeedShared feed_shared;
    feed.init(feed_counter, n_chunks, (RT_LDS FeedShared*)&feed_shared);
    TravMem M;
    M.n_lds = S.n_lds;
    M.lds_nodes = (const RT_LDS BvhNode*)trav_lds;
    M.stack = (RT_LDS int*)(trav_lds + (size_t)S.n_lds * sizeof(BvhNode));
    M.spill_lanes = S.spill_lanes;
    M.spill = S.spill + ((long long)blockIdx.x * TBLOCK + threadIdx.x) % S.spill_lanes;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(S.bvh);
    RT_LDS u32x4* dst = (RT_LDS u32x4*)trav_lds;
    const int n16 = S.n_lds * int(sizeof(BvhNode) / 16);
    for (int k = threadIdx.x; k < n16; k += TBLOCK) dst[(k & 7) * S.n_lds + (k >> 3)] = src[k];  // column-major, see bvh_traverse
    __syncthreads();
    return M;
}

// ---------------------------------------------------------------------------------------------------------
// Wave-level helpers.

// `counters`: the first of the RT_WORK_COUNT words of one work set (CNT_W_XFORMS, CNT_WS_XFORMS or CNT_WC_XFORMS)
__device__ __forceinline__ void flush_work(const Work& w, unsigned long long* __restrict__ counters) {
    const unsigned* p = reinterpret_cast<const unsigned*>(&w);
    constexpr int N = sizeof(Work) / sizeof(unsigned);
    for (int k = 0; k < N; ++k) {
        unsigned v = p[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(counters + k, (unsigned long long)v);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Kernel: ray generation == Camera::ray_for_pixel (src/camera.rs:37-60) for this shard's 8x8 tiles.

__device__ __forceinline__ bool slot_to_pixel(const CameraParams& c, long long slot, int& px, int& py) {
    long long local_tile = slot >> 6;
    int lane = int(slot & 63);
    long long tile = local_tile * c.n_ranks + c.rank;
    int ty = int(tile / c.tiles_x), tx = int(tile - (long long)ty * c.tiles_x);
    px = tx * 8 + (lane & 7);
    py = ty * 8 + (lane >> 3);
    return ty < c.tiles_y && px < c.hsize - 1 && py < c.vsize - 1;  // F1: last row / column never rendered
}

__device__ __forceinline__ void ray_for_pixel(const CameraParams& c, int px, int py, V3& origin, V3& direction) {
    double x_offset = ((double)px + 0.5) * c.pixel_size;
    double y_offset = ((double)py + 0.5) * c.pixel_size;
    double world_x = c.half_width - x_offset;
    double world_y = c.half_height - y_offset;
    V3 pixel = xf_point(c.inv, mk(world_x, world_y, -1.));
    origin = xf_point(c.inv, mk(0., 0., 0.));
    direction = normalize(vsub(pixel, origin));
}

__global__ __launch_bounds__(BLOCK) void k_raygen(CameraParams c, Level L, long long n_slots) {
    long long slot = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= n_slots) return;
    int px, py;
    if (!slot_to_pixel(c, slot, px, py)) {
        L.ox[slot] = dead_ray();  // padding slot: no ray
        return;
    }
    V3 origin, direction;
    ray_for_pixel(c, px, py, origin, direction);
    L.ox[slot] = origin.x; L.oy[slot] = origin.y; L.oz[slot] = origin.z;
    L.dx[slot] = direction.x; L.dy[slot] = direction.y; L.dz[slot] = direction.z;
}

// the camera of the frame being enqueued, for the level-0 k_trace that makes its rays from it (by value here, so that a
// captured frame graph holds it)
__global__ void k_set_camera(CameraParams c, CameraParams* dst) {
    if (threadIdx.x == 0) *dst = c;
}

__global__ __launch_bounds__(BLOCK) void k_rays_for_pixels(CameraParams c, long long n, const int32_t* __restrict__ px,
                                                           const int32_t* __restrict__ py, double* o_out, double* d_out) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    V3 origin, direction;
    ray_for_pixel(c, px[i], py[i], origin, direction);
    o_out[3 * i] = origin.x; o_out[3 * i + 1] = origin.y; o_out[3 * i + 2] = origin.z;
    d_out[3 * i] = direction.x; d_out[3 * i + 1] = direction.y; d_out[3 * i + 2] = direction.z;
}

// ---------------------------------------------------------------------------------------------------------
// Shading: prepare_computations, is_shadowed, lighting, ray spawn.

// Shape::normal_at (src/shape.rs:92-108): world_to_object down the registry chain, local normal, normal_to_world up
// (normal_at_wave below; world_to_object alone also serves Pattern::pattern_at_shape).
__device__ __forceinline__ V3 world_to_object(const DScene& S, int node, V3 p) {  // src/shape.rs:394-404
    const int* __restrict__ c = S.chain_pool + S.node_nchain[node];
    for (int k = c[0]; k >= 1; --k)  // outermost ancestor first
        if (c[k] != 0) p = xf_point(S.xf + (size_t)c[k] * 12, p);
    return p;
}
// Local normals (each kind's normal_at).  What they need from memory besides the point arrives as arguments, so that
// the caller can request it ahead of time: `tri_n` = the flat triangle's stored normal, `cmin`/`cmax` = conic limits.
__device__ __forceinline__ V3 local_normal_at(const DScene& S, int kind, int param, V3 p, double hu, double hv, V3 tri_n, double cmin,
                                              double cmax) {
    switch (kind) {
        case RT_SPHERE: return p;               // point - ORIGIN, src/sphere.rs:65-67
        case RT_PLANE: return mk(0., 1., 0.);   // src/plane.rs:53-56
        case RT_CUBE: {                         // src/cube.rs:81-92
            double maxc = max3nan(fabs(p.x), fabs(p.y), fabs(p.z));
            if (near_eq(maxc, fabs(p.x))) return mk(p.x, 0., 0.);
            else if (near_eq(maxc, fabs(p.y))) return mk(0., p.y, 0.);
            else return mk(0., 0., p.z);
        }
        case RT_CYLINDER: {  // src/cylinder.rs:113-127
            double minimum = cmin, maximum = cmax;
            double distance = p.x * p.x + p.z * p.z;
            if (distance < 1. && (p.y > maximum - EPS || near_eq(p.y, maximum - EPS))) return mk(0., 1., 0.);
            else if (distance < 1. && (p.y < minimum + EPS || near_eq(p.y, minimum + EPS))) return mk(0., -1., 0.);
            else return mk(p.x, 0., p.z);
        }
        case RT_CONE: {  // src/cone.rs:119-135
            double minimum = cmin, maximum = cmax;
            double distance = p.x * p.x + p.z * p.z;
            if (distance < maximum * maximum && (p.y > maximum - EPS || near_eq(p.y, maximum - EPS))) return mk(0., 1., 0.);
            else if (distance < minimum * minimum && (p.y < minimum + EPS || near_eq(p.y, minimum + EPS))) return mk(0., -1., 0.);
            else {
                double y = sqrt(distance);
                return mk(p.x, p.y > 0. ? -y : y, p.z);
            }
        }
        case RT_TRIANGLE: return tri_n;  // src/triangle.rs:89-91
        default: {  // RT_SMOOTH_TRIANGLE, src/smooth_triangle.rs:98-104
            const double* __restrict__ s = S.smooth + (size_t)S.tri_smooth[param] * 9;
            V3 n1 = mk(s[0], s[1], s[2]), n2 = mk(s[3], s[4], s[5]), n3 = mk(s[6], s[7], s[8]);
            return vadd(vadd(vmul(n2, hu), vmul(n3, hv)), vmul(n1, (1. - hu - hv)));
        }
    }
}

// Shape::normal_at for the hits of one wavefront (k_trace's epilogue).  Lanes of a wavefront hit a handful of distinct
// transform chains (all triangles of a mesh share one), so the chains are walked one at a time with the chain and its
// matrices fetched by scalar loads: a lane-private walk costs a dependent vector-load round trip per chain entry and
// per matrix (node -> chain offset -> length -> entry -> 12 matrix elements, twice), which made prepare_computations
// the longest part of a chunk.  Arithmetic and its order are those of world_to_object / normal_to_world above.
__device__ __forceinline__ V3 normal_at_wave(const DScene& S, bool hit, const NodeRec& rec, V3 world_point, double hu, double hv) {
    // what the local normal needs from memory, requested before the chain walk so that it arrives meanwhile
    V3 tri_n = mk(0., 0., 0.);
    double cmin = 0., cmax = 0.;
    if (hit && rec.kind == RT_TRIANGLE) {
        const double* __restrict__ t = S.tri + (size_t)rec.param * 12;
        tri_n = mk(t[9], t[10], t[11]);
    } else if (hit && (rec.kind == RT_CYLINDER || rec.kind == RT_CONE)) {
        cmin = S.conic[2 * rec.param];
        cmax = S.conic[2 * rec.param + 1];
    }
    const RT_CONSTANT int32_t* __restrict__ pool = as_constant(S.chain_pool);
    const RT_CONSTANT double* __restrict__ xf = as_constant(S.xf);
    V3 result = mk(0., 0., 0.);
    bool pending = hit;
    while (true) {
        const unsigned long long m = __ballot(pending);
        if (!m) break;
        const int cid = __builtin_amdgcn_readlane(rec.nchain, __ffsll((long long)m) - 1);
        // The compiler must not learn that the address below equals the lane's own rec.nchain inside the branch (it
        // would substitute the per-lane value and fall back to vector loads): the address uses an opaque scalar copy.
        int cid_scalar = cid;
        asm volatile("" : "+s"(cid_scalar));
        if (pending && rec.nchain == cid) {
            const RT_CONSTANT int32_t* __restrict__ c = pool + cid_scalar;  // [len, x_leaf, x_parent, ..., x_root]
            const int len = c[0];
            V3 p = world_point;
            for (int k = len; k >= 1; --k) {  // world_to_object, outermost ancestor first (src/shape.rs:394-404)
                const int x = c[k];
                if (x != 0) p = xf_point(xf + (size_t)x * 12, p);
            }
            V3 n = local_normal_at(S, rec.kind, rec.param, p, hu, hv, tri_n, cmin, cmax);
            for (int k = 1; k <= len; ++k) {  // normal_to_world (src/shape.rs:406-420): every level normalises
                const int x = c[k];
                if (x != 0) n = xf_normal(xf + (size_t)x * 12, n);
                n = normalize(n);
            }
            result = n;
            pending = false;
        }
    }
    return result;
}

__device__ __forceinline__ int as_i32(double x) {  // Rust `as i32`: saturating, NaN -> 0
    if (x != x) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return (-2147483647 - 1);
    return (int)x;
}

// Pattern::pattern_at_shape (src/pattern.rs:23-35) and the six pattern_at bodies
__device__ __forceinline__ V3 pattern_at(const DScene& S, int pat, V3 p);
__device__ __forceinline__ V3 pattern_at_shape(const DScene& S, int pat, int node, V3 world_point) {
    V3 object_point = world_to_object(S, node, world_point);
    V3 p = xf_point(S.pat_inv + (size_t)pat * 12, object_point);
    return pattern_at(S, pat, p);
}
// the pattern_at bodies (src/pattern.rs:72-177): x, y, z of the pattern-space point only
__device__ __forceinline__ V3 pattern_at(const DScene& S, int pat, V3 p) {
    const double* __restrict__ ab = S.pat_ab + (size_t)pat * 6;
    V3 a = mk(ab[0], ab[1], ab[2]), b = mk(ab[3], ab[4], ab[5]);
    switch (S.pat_kind[pat]) {
        case RT_PAT_STRIPED: return as_i32(floor(p.x)) % 2 == 0 ? a : b;
        case RT_PAT_GRADIENT: {
            V3 distance = vsub(b, a);
            double fraction = p.x - floor(p.x);
            return vadd(a, vmul(distance, fraction));
        }
        case RT_PAT_RING: return as_i32(floor(sqrt(p.x * p.x + p.z * p.z))) % 2 == 0 ? a : b;
        case RT_PAT_CHECKERED: return as_i32(floor(p.x) + floor(p.y) + floor(p.z)) % 2 == 0 ? a : b;
        case RT_PAT_RING_GRADIENT: {
            V3 distance = vsub(b, a);
            double c = sqrt(p.x * p.x + p.z * p.z);
            double fraction = c - floor(c);
            return vadd(a, vmul(distance, fraction));
        }
        default: return p;  // TestPattern
    }
}

// Material::lighting (src/material.rs:97-129); colours ride in V3
__device__ __forceinline__ V3 lighting(const DScene& S, const double* __restrict__ m, int pat, int node, int light, V3 point,
                                       V3 eye_vector, V3 normal_vector, bool in_shadow) {
    V3 real_color = pat >= 0 ? pattern_at_shape(S, pat, node, point) : mk(m[RT_MAT_R], m[RT_MAT_G], m[RT_MAT_B]);
    const double* __restrict__ l = S.light + (size_t)light * 6;
    V3 light_position = mk(l[0], l[1], l[2]), intensity = mk(l[3], l[4], l[5]);
    V3 effective_color = vhad(real_color, intensity);
    V3 light_vector = normalize(vsub(light_position, point));
    V3 ambient = vmul(effective_color, m[RT_MAT_AMBIENT]);
    double light_dot_normal = dot(light_vector, normal_vector);
    V3 diffuse, specular;
    if (light_dot_normal < 0.) {
        diffuse = mk(0., 0., 0.);
        specular = mk(0., 0., 0.);
    } else {
        diffuse = vmul(vmul(effective_color, m[RT_MAT_DIFFUSE]), light_dot_normal);
        V3 reflect_vector = reflect(vneg(light_vector), normal_vector);
        double reflect_dot_eye = dot(reflect_vector, eye_vector);
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

// World::is_shadowed for one light (src/world.rs:92-107): NEAREST hit only, then `t < distance && casts_shadow` (F8)
template <int SC, bool COUNT>
__device__ __forceinline__ bool is_shadowed(const DScene& S, const TravMem& M, V3 point, int light, Work& w, unsigned& flags,
                                            bool alive) {
    const double* __restrict__ l = S.light + (size_t)light * 6;
    V3 vector = vsub(mk(l[0], l[1], l[2]), point);
    double distance = magnitude(vector);
    V3 direction = normalize(vector);
    ClosestSink<false> sink;
    // hits at t >= distance cannot shadow and cannot displace a nearer hit: bound the search there
    sink.init(distance);
    // F8: the answer is `nearest.t < distance && nearest.casts_shadow`.  When EVERY leaf of the scene casts shadows that
    // is "some hit has 0 < t < distance", and the first such hit ends the search; otherwise the nearest must be found.
    if (S.all_cast_shadow) sink.stop_below = distance;
    traverse<SC, COUNT>(S, M, point, direction, sink, w, alive);
    flags |= sink.flags;
    if (sink.best_key < 0) return false;
    return sink.best_t < distance && S.node_casts_shadow[sink.best_key >> 2] != 0;
}

// The n1/n2 walk of prepare_computations (src/intersection.rs:54-80) over the sorted list up to the hit.
// The list is collected into per-lane scratch memory (ListSink).  Walking it there is a chain of ~50 dependent memory
// round trips per ray — insertion sort, then equality class and material of every entry and of every container —
// and a level's container-walk launch took up to 150 us for 4000 chunks, longer than the level's shadow rays.  So a
// list of at most SHORT_LIST entries (nearly all of them) is fetched into registers in one go, sorted with a network,
// what its entries need from memory is requested for all of them at once, and the reference's loop then runs literally
// on register arrays (`containers` = (class, refractive index) pairs, searched / shifted / appended to with
// compile-time indices).  Longer lists take the general loop below on the scratch arrays.
template <int SC, bool COUNT>
__device__ __forceinline__ void container_walk(const DScene& S, const TravMem& M, V3 o, V3 d, double hit_t, int hit_node, Work& w,
                                               double& n1_out, double& n2_out, unsigned& flags, bool alive) {
    constexpr bool BIG = (SC & SC_BIG) != 0;
    double lt_private[BIG ? 1 : LIST_CAP];
    int lk_private[BIG ? 1 : LIST_CAP];
    double* lt = lt_private;
    int* lk = lk_private;
    int list_cap = LIST_CAP;
    if (BIG) {
        list_cap = S.arena_list_cap;
        lt = reinterpret_cast<double*>(arena_list(S));
        lk = reinterpret_cast<int*>(lt + list_cap);
    }
    WalkSink sink;
    sink.t = lt; sink.key = lk; sink.n = 0; sink.cap = list_cap; sink.limit = hit_t; sink.flags = 0;
    sink.below_t = -INFINITY; sink.below_key = -1; sink.hit_node = hit_node; sink.passive = S.passive_mesh_tris != 0;
    const int hit_cls = alive ? S.node_eq_class[hit_node] : -1;
    double n1 = 0., n2 = 0.;
    // Round 3 — how far back the mesh search has to look.  A passive entry T (WalkSink) is only ever observed as
    // containers.last() at an entry that equals the hit (`is_self` below: src/intersection.rs:56-62 before the toggle,
    // :72-78 after it), and it is the last container there only if nothing pushed after it is still in the vector.  In a
    // program whose meshes all come behind everything else (DScene::walk_bound_item: the mesh-free objects are emitted
    // first, rt_program.cpp) the walk runs in two parts: the items in front of the first mesh, then the reference's loop
    // over what they listed, then the meshes.  Passive entries can neither remove a container nor equal the hit, so the
    // entries of the first part are toggled in that replay exactly as they will be in the full list, and every
    // observation names the container on top, pushed at t_top: a passive entry below t_top is buried under it there.  An
    // observation whose top has the refractive index that every mesh triangle of the scene has (walk_passive_ior; the
    // empty vector reads as 1.0) reads the same value with or without a passive entry above it and asks for nothing.
    // The minimum of t_top over the observations that do ask bounds the search (WalkSink::lo_mesh): a ray inside a glass
    // sphere no longer searches a teapot that its line meets somewhere behind the sphere.  When the meshes add nothing to
    // the list — nearly always — the first replay's n1 / n2 are the result.  No observation in the first part (the hit is
    // a mesh triangle), a NaN in the list, a list that is not short: no bound, and the walk is what it was.
    // rt_scene_opts.walk_bound = 0 switches it off (one part, bit-identical: tested).
    const int split = S.walk_bound_item >= 0 && sink.passive ? S.walk_bound_item : S.n_items;
    const long long passive_bits = __double_as_longlong(S.walk_passive_ior);
    int n_replayed = -1;
#pragma nounroll
    for (int part = 0; part < 2; ++part) {
    const int part_begin = part == 0 ? 0 : split, part_end = part == 0 ? split : S.n_items;
    if (part_begin >= part_end && part == 1) break;
    traverse<SC, COUNT>(S, M, o, d, sink, w, alive, part_begin, part_end);
    const bool last_part = part == 1 || split >= S.n_items;
    if (last_part) {
        sink.finish();
        flags |= sink.flags;
    }
    if (last_part && sink.n == n_replayed) break;  // (finish() included: the meshes added nothing)
    n_replayed = sink.n;
    const int n = sink.n < list_cap ? sink.n : list_cap;
    n1 = 0.; n2 = 0.;
    double bound = INFINITY;  // least push time of a container on top that a passive entry above it would change
    bool observed = false;
    const bool is_short = n <= SHORT_LIST;
    if (is_short) {
        ShortList L;
        int cls[SHORT_LIST];
        double ior[SHORT_LIST];
#pragma unroll
        for (int i = 0; i < SHORT_LIST; ++i) {
            L.t[i] = INFINITY;
            L.key[i] = 0x7fffffff;
            if (i < n) { L.t[i] = lt[i]; L.key[i] = lk[i]; }
        }
        L.sort();
#pragma unroll
        for (int i = 0; i < SHORT_LIST; ++i) {
            cls[i] = -1;
            ior[i] = 1.;
            if (i < n) {
                const int node = L.key[i] >> 2;
                cls[i] = S.node_eq_class[node];
                ior[i] = S.mat[(size_t)S.node_material[node] * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
            }
        }
        int ccls[SHORT_LIST];     // containers: equality class ...
        double cior[SHORT_LIST];  // ... and refractive index of each
        float cpush[SHORT_LIST];  // ... and the t of the entry that pushed it, rounded DOWN to f32 (it only ever bounds a search from below)
#pragma unroll
        for (int k = 0; k < SHORT_LIST; ++k) { ccls[k] = -1; cior[k] = 1.; cpush[k] = -INFINITY; }
        int nc = 0;
#pragma unroll
        for (int i = 0; i < SHORT_LIST; ++i) {
            if (i < n) {
                if (!(L.t[i] == L.t[i])) bound = -INFINITY;  // a NaN in the list: no bound
                const bool is_self = near_eq(L.t[i], hit_t) && cls[i] >= 0 && cls[i] == hit_cls;  // Intersection::eq (F6, F14)
                double last = 1.;  // containers.last()'s refractive index, 1.0 when empty (src/intersection.rs:57-62)
                float last_push = -INFINITY;
#pragma unroll
                for (int k = 0; k < SHORT_LIST; ++k)
                    if (k == nc - 1) { last = cior[k]; last_push = cpush[k]; }
                if (is_self) {
                    n1 = last;
                    observed = true;
                    if (__double_as_longlong(last) != passive_bits) bound = fmin(bound, (double)last_push);
                }
                int found = -1;  // position() of the first equal container (src/intersection.rs:65)
#pragma unroll
                for (int k = SHORT_LIST - 1; k >= 0; --k)
                    if (k < nc && ccls[k] >= 0 && ccls[k] == cls[i]) found = k;
                if (found >= 0) {
#pragma unroll
                    for (int k = 0; k + 1 < SHORT_LIST; ++k)
                        if (k >= found) { ccls[k] = ccls[k + 1]; cior[k] = cior[k + 1]; cpush[k] = cpush[k + 1]; }
                    --nc;
                } else {
#pragma unroll
                    for (int k = 0; k < SHORT_LIST; ++k)
                        if (k == nc) { ccls[k] = cls[i]; cior[k] = ior[i]; cpush[k] = __double2float_rd(L.t[i]); }
                    ++nc;
                }
                if (is_self) {
                    double last2 = 1.;
                    float last2_push = -INFINITY;
#pragma unroll
                    for (int k = 0; k < SHORT_LIST; ++k)
                        if (k == nc - 1) { last2 = cior[k]; last2_push = cpush[k]; }
                    n2 = last2;
                    if (__double_as_longlong(last2) != passive_bits) bound = fmin(bound, (double)last2_push);
                }
            }
        }
    } else {
        bound = -INFINITY;
        sort_list(lt, lk, n);
        int containers_private[BIG ? 1 : CONTAINER_CAP];
        int* containers = BIG ? lk + list_cap : containers_private;
        const int container_cap = BIG ? list_cap : CONTAINER_CAP;
        int nc = 0;
        for (int i = 0; i < n; ++i) {
            int node = lk[i] >> 2;
            int cls = S.node_eq_class[node];
            bool is_self = near_eq(lt[i], hit_t) && cls >= 0 && cls == hit_cls;  // Intersection::eq (F6, F14)
            if (is_self) n1 = nc == 0 ? 1. : S.mat[(size_t)S.node_material[containers[nc - 1]] * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
            int found = -1;
            for (int k = 0; k < nc; ++k) {
                int ck = S.node_eq_class[containers[k]];
                if (ck >= 0 && ck == cls) { found = k; break; }
            }
            if (found >= 0) {
                for (int k = found; k + 1 < nc; ++k) containers[k] = containers[k + 1];
                --nc;
            } else if (nc < container_cap) {
                containers[nc++] = node;
            } else {
                flags |= RT_FLAG_LIST_OVERFLOW;
            }
            if (is_self) n2 = nc == 0 ? 1. : S.mat[(size_t)S.node_material[containers[nc - 1]] * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
        }
    }
    // (entries inside the window of the hit, t > hit_t - 2 EPS, are listed in full whatever the bound)
    if (!last_part && observed && hit_t == hit_t) sink.below_t = fmin(bound, hit_t - 2. * EPS);
    }  // part
    n1_out = n1;
    n2_out = n2;
}

// Intersection::schlick (src/intersection.rs:97-115) from cos = eye_vector . normal_vector
__device__ __forceinline__ double schlick_from(double cosv, double n1, double n2) {
    if (n1 > n2) {
        double n = n1 / n2;
        double sin2_t = (n * n) * (1. - cosv * cosv);
        if (sin2_t > 1.) return 1.;
        double cos_t = sqrt(1. - sin2_t);
        cosv = cos_t;
    }
    double q = (n1 - n2) / (n1 + n2);
    double r0 = q * q;
    double x = 1. - cosv;
    return r0 + (1. - r0) * (x * ((x * x) * (x * x)));
}
__device__ __forceinline__ double schlick(V3 eye_vector, V3 normal_vector, double n1, double n2) {
    return schlick_from(dot(eye_vector, normal_vector), n1, n2);
}

// ---------------------------------------------------------------------------------------------------------
// Stage bits (Level::mode), written by k_trace for every ray of a level and read by k_queue, k_walk, k_spawn, k_shade.
// Between k_trace and k_shade the world-space point and normal of the hit live in the px..nz columns, and — once
// prepare_computations has consumed the hit's u,v — n1/n2 in the hit-u/v columns.
constexpr int ST_HIT = 1;        // the ray hit something: shade it
constexpr int ST_WALK = 2;       // prepare_computations' n1/n2 are observable for this hit: run the container walk
constexpr int ST_REFLECT = 4;    // reflected_color spawns a ray (src/world.rs:110)
constexpr int ST_REFRACT = 8;    // refracted_color gets past its first test (src/world.rs:121); a slot is reserved
constexpr int ST_TIR = 16;       // ... but total internal reflection (src/world.rs:128): the slot stays a dead ray

__device__ __forceinline__ unsigned long long lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }

__device__ __forceinline__ bool needs_walk(const DScene& S, double reflective, double transparency, int remaining) {
    // n1/n2 are read only by refracted_color and schlick, both behind a transparency test.  At remaining <= 0
    // refracted_color and reflected_color return BLACK before using them (src/world.rs:110,121) and schlick's result
    // only scales those two BLACKs: any FINITE reflectance gives surface + 0 + 0.  It is finite whenever every
    // refractive index in the scene is finite and positive and every leaf is equal to itself (F14: one that is not
    // leaves n1 = n2 = 0 and schlick = NaN) — S.walk_at_zero == 0 — so the walk is skipped there and k_shade uses
    // reflectance 0.
    if (remaining <= 0 && !S.walk_at_zero) return false;
    return !near_eq(transparency, 0.) || (reflective > 0. && transparency > 0.);
}
// the part of needs_walk that depends on the recursion level; the material's part is NODE_WALKS
__device__ __forceinline__ bool needs_walk_at(const DScene& S, int remaining) { return !(remaining <= 0 && !S.walk_at_zero); }

// Ray count of a level: known to the host (`n_dev` == null), or — when a whole frame is enqueued without a host
// round trip per level — read from the previous level's CNT_NEXT, clamped to the capacity `n` the host passes instead.
__device__ __forceinline__ long long level_count(long long n, const unsigned long long* __restrict__ n_dev) {
    if (!n_dev) return n;
    const long long v = (long long)*n_dev;
    return v < n ? v : n;
}

// over_point / under_point exactly as prepare_computations forms them (src/intersection.rs:50-51)
__device__ __forceinline__ V3 over_point_of(V3 point, V3 normal) { return vadd(point, vmul(normal, EPS)); }
__device__ __forceinline__ V3 under_point_of(V3 point, V3 normal) { return vsub(point, vmul(normal, EPS)); }

// reflected_color / refracted_color ray spawn (src/world.rs:109-137) for the slot k_queue reserved in the next level
// and tagged with its parent (`tag` = 2 * parent ray + 1 for the refraction ray): the ray that color_at is called with.
// Needs the parent's prepare_computations (k_trace) and, for a refraction ray, its n1/n2 walk (k_walk).  A refraction
// slot whose parent ran into total internal reflection stays a dead ray (refracted_color returns BLACK, src/world.rs:
// 128-130) and is unhooked from its parent.  Called by k_trace when it fetches the rays of a level (the spawn then
// costs no launch and no pass over the level: its loads overlap traversal work), or by k_spawn.
__device__ __forceinline__ void spawn_ray(const Level& P, int tag, V3& o, V3& d) {
    const long long i = tag >> 1;
    const V3 pd = mk(P.dx[i], P.dy[i], P.dz[i]);
    const V3 point = mk(P.px[i], P.py[i], P.pz[i]), normal_vector = mk(P.nx[i], P.ny[i], P.nz[i]);
    if (!(tag & 1)) {  // reflected_color, src/world.rs:109-118
        o = over_point_of(point, normal_vector);
        d = reflect(pd, normal_vector);  // reflect_vector, src/intersection.rs:52
        return;
    }
    const int stage = P.mode[i];
    if (stage & ST_TIR) {
        P.child_refr[i] = -1;
        return;  // o, d stay the caller's dead ray
    }
    // refracted_color, src/world.rs:120-137
    const bool walked = stage & ST_WALK;
    const double n1 = walked ? P.hu[i] : 0., n2 = walked ? P.hv[i] : 0.;
    const V3 eye_vector = vneg(pd);
    double n_ratio = n1 / n2;
    double cos_i = dot(eye_vector, normal_vector);
    double sin2_t = (n_ratio * n_ratio) * (1. - cos_i * cos_i);
    double cos_t = sqrt(1. - sin2_t);
    d = vsub(vmul(normal_vector, (n_ratio * cos_i - cos_t)), vmul(eye_vector, n_ratio));
    o = under_point_of(point, normal_vector);
}

// One launch = the closest-hit pass of level d (rays of L, `remaining` is the value color_at was called with at this
// level) AND the shadow rays of level d-1 (hits of P, queued by k_queue(d-1) in `prev_counters`' block).  is_shadowed
// feeds only Material::lighting (src/world.rs:57-66), which nothing downstream of the ray tree waits for: the rays of
// level d need prepare_computations and the n1/n2 walk of level d-1, never its shadows.  So the shadow rays ride along
// with the next level's closest-hit pass instead of having a launch of their own: a traversal launch lasts as long as
// its slowest 64-ray chunk (60-90 us whatever the ray count), and this way a frame pays that once per level, not
// twice.  Either part may be empty (n_cap == 0: shadow rays only — the launch after the last level; prev_counters ==
// null: level 0).  Every wavefront alternates between the two chunk feeds until both are dry.
// RAYGEN: the variant that traces level 0 of a Camera::render frame and makes its rays from the camera (a template flag, not
// a third value of `from_parent`: with the ray_for_pixel code in the fetch of every launch the variants of scenes with cull
// ranges or CSG went from 33-44 to 85-94 spilled registers; this way the kernels of levels 1.. are what they were).
template <int SC, bool NEED_FIRST, bool COUNT, bool RAYGEN = false>
__global__ __launch_bounds__(TBLOCK, RT_TWAVES) void k_trace(DScene S, Level L, long long n_cap, const unsigned long long* n_dev, int remaining,
                                                  unsigned long long* counters, Level P, unsigned long long* prev_counters, int from_parent,
                                                  const CameraParams* __restrict__ cam) {
    const long long n = n_cap > 0 ? level_count(n_cap, n_dev) : 0;
    const long long n_shadow = prev_counters ? (long long)prev_counters[CNT_SHADOW_Q] * S.n_lights : 0;
    if (n == 0 && n_shadow == 0) return;  // uniform over the grid: nothing staged, nothing fed
    Feed feed, shadow_feed;
    __shared__ FeedShared shadow_shared;
    shadow_feed.init(prev_counters ? prev_counters + CNT_FEED_SHADOW : nullptr, (n_shadow + 63) / 64, (RT_LDS FeedShared*)&shadow_shared);
    const TravMem M = trav_setup(S, feed, counters + CNT_FEED_TRACE, (n + 63) / 64);
    Work w = {}, ws = {};
    // A chunk's rays are requested one chunk ahead (just before the previous chunk's epilogue, whose dependent look-ups
    // they overlap): a wavefront is one of four on its SIMD, so nothing else hides that round trip.
    auto fetch = [&](long long c, V3& o, V3& d) {
        const long long i = c * 64 + (threadIdx.x & 63);
        o = mk(dead_ray(), 0., 0.);  // padding slot / dead ray
        d = mk(0., 0., 1.);
        if (c >= 0 && i < n) {
            if (RAYGEN) {  // level 0 of Camera::render: ray_for_pixel (src/camera.rs:37-60) for the slot's pixel; no k_raygen
                                     // launch, no pass over the level's ray columns (a padding slot stays the dead ray)
                // The camera is read HERE, by scalar loads from its device copy (k_set_camera): as a by-value kernel argument
                // it is loaded at the kernel's entry and its ~40 scalar registers stay live through the traversal (the variants
                // of scenes with cull ranges went from 33 to 85 spilled registers); the opaque offset keeps the loads in place.
                int zero = 0;
                asm volatile("" : "+s"(zero));
                const RT_CONSTANT CameraParams* q = as_constant(cam) + zero;
                CameraParams cp;  // (field by field: a struct cannot be copied out of the constant address space as a whole)
#pragma unroll
                for (int k = 0; k < 12; ++k) cp.inv[k] = q->inv[k];
                cp.half_width = q->half_width; cp.half_height = q->half_height; cp.pixel_size = q->pixel_size;
                cp.hsize = q->hsize; cp.vsize = q->vsize; cp.tiles_x = q->tiles_x; cp.tiles_y = q->tiles_y;
                cp.rank = q->rank; cp.n_ranks = q->n_ranks;
                int px, py;
                if (slot_to_pixel(cp, i, px, py)) ray_for_pixel(cp, px, py, o, d);
            } else if (from_parent) {
                spawn_ray(P, L.hkey[i], o, d);
            } else {
                o = mk(L.ox[i], L.oy[i], L.oz[i]);
                d = mk(L.dx[i], L.dy[i], L.dz[i]);
            }
        }
    };
    // the hit a lane's shadow ray belongs to: the queue's indirection is taken one chunk ahead too, so that a shadow
    // chunk starts with one round trip (point and normal), not two.  Same-box A/B per scene class: C4 (CSG + cull
    // ranges) 6.78 -> 6.53 ms, C5 (cull ranges) 18.1 -> 17.5 ms, but the headline's plain variant 1.85 -> 1.90 ms (one
    // more value alive through the traversal costs it more in spills than the round trip saves): plain scenes keep
    // the late look-up
    constexpr bool AHEAD = (SC & (SC_CSG | SC_CULL)) != 0;
    auto shadow_index = [&](long long chunk) -> int {
        const long long j = chunk * 64 + (threadIdx.x & 63);
        return chunk >= 0 && j < n_shadow ? P.shadow_q[j / S.n_lights] : 0;
    };
    long long c = feed.next();
    long long cs = shadow_feed.next();
    int shadow_i = AHEAD ? shadow_index(cs) : 0;
    V3 o, d;
    fetch(c, o, d);
    while (c >= 0 || cs >= 0) {
        if (c >= 0) {
            const long long i = c * 64 + (threadIdx.x & 63);
            const bool alive = !is_dead_ray(o.x);
            if (!alive) o = mk(0., 0., 0.);
            ClosestSink<NEED_FIRST> sink;
            sink.init(INFINITY);
            traverse<SC, COUNT>(S, M, o, d, sink, w, alive);
            const long long c_next = feed.next();
            V3 o_next, d_next;
            fetch(c_next, o_next, d_next);
            // ---- epilogue: prepare_computations (src/intersection.rs:38-48) for the lanes that hit something
            const int key = alive ? sink.best_key : -1;
            const bool hit = key >= 0;
            const int node = hit ? key >> 2 : 0;
            NodeRec rec = {0, 0, 0, 0};
            if (hit) rec = S.node_rec[node];
            // what the stage bits need to know about the hit's material rides in the record (NODE_* bits, set by the host
            // with the device's own near_eq): one dependent look-up less per chunk
            const int mbits = rec.kind >> 8;
            rec.kind &= 0xff;
            double u = sink.bu, v = sink.bv;
            unsigned flags = alive ? sink.flags : 0;  // a NaN t panics the reference's sort whether or not anything is hit
            if (NEED_FIRST && hit && rec.kind == RT_SMOOTH_TRIANGLE) {
                if (sink.first_uv) { u = sink.fu; v = sink.fv; }  // u,v of intersections[0] (F5)
                else flags |= RT_FLAG_REF_PANIC_UV;                // reference would unwrap a None
            }
            const V3 point = vadd(o, vmul(d, sink.best_t));  // Ray::position
            V3 normal_vector = normal_at_wave(S, hit, rec, point, u, v);
            int stage = 0;
            if (hit) {
                const V3 eye_vector = vneg(d);
                if (dot(normal_vector, eye_vector) < 0.) normal_vector = vneg(normal_vector);  // inside
                L.px[i] = point.x; L.py[i] = point.y; L.pz[i] = point.z;
                L.nx[i] = normal_vector.x; L.ny[i] = normal_vector.y; L.nz[i] = normal_vector.z;
                L.ht[i] = sink.best_t;
                if (!RAYGEN && from_parent) {  // a ray made here exists nowhere else: k_walk, k_shade and the next level's spawn read a HIT's ray
                    L.ox[i] = o.x; L.oy[i] = o.y; L.oz[i] = o.z;
                    L.dx[i] = d.x; L.dy[i] = d.y; L.dz[i] = d.z;
                }
                stage = ST_HIT;
                if ((mbits & NODE_WALKS) && needs_walk_at(S, remaining)) stage |= ST_WALK;
                if ((mbits & NODE_REFLECTS) && remaining > 0) stage |= ST_REFLECT;      // src/world.rs:110
                if ((mbits & NODE_REFRACTS) && remaining > 0) stage |= ST_REFRACT;      // src/world.rs:121
            }
            // level 0's rays made here: the overflow fix-up (k_collect_overflow) also reads a flagged ray that hit nothing
            if (RAYGEN && (hit || (flags & RT_FLAG_LIST_OVERFLOW))) {
                L.ox[i] = o.x; L.oy[i] = o.y; L.oz[i] = o.z;
                L.dx[i] = d.x; L.dy[i] = d.y; L.dz[i] = d.z;
            }
            if (i < n) {
                L.hkey[i] = key;
                L.mode[i] = stage;
                L.flags[i] = (uint8_t)flags;
            }
            c = c_next;
            o = o_next;
            d = d_next;
        }
        if (cs >= 0) {  // World::is_shadowed for 64 (hit, light) pairs of the previous level
            const long long j = cs * 64 + (threadIdx.x & 63);
            const bool alive = j < n_shadow;
            int light = 0;
            const long long i = AHEAD ? shadow_i : shadow_index(cs);  // asked for one chunk ahead
            V3 over_point = mk(0., 0., 0.);
            if (alive) {
                light = (int)(j % S.n_lights);
                over_point = over_point_of(mk(P.px[i], P.py[i], P.pz[i]), mk(P.nx[i], P.ny[i], P.nz[i]));
            }
            unsigned flags = 0;
            const bool sh = is_shadowed<SC, COUNT>(S, M, over_point, light, ws, flags, alive);
            if (alive) {
                P.shadowed[i * S.n_lights + light] = sh ? 1 : 0;
                if (flags) atomicOr(P.flags32 + (i >> 2), flags << (8 * (i & 3)));
            }
            cs = shadow_feed.next();
            if (AHEAD) shadow_i = shadow_index(cs);
        }
    }
    if (COUNT) {
        flush_work(w, counters + CNT_W_XFORMS);
        if (prev_counters) flush_work(ws, prev_counters + CNT_WS_XFORMS);
    }
}

// Queue appends without an atomic per wavefront (returning atomics on one word are serialised in L2, 5-10 ns each
// chip-wide): a workgroup owns a TILE of consecutive rays, counts its appends in a first pass over the stage bits,
// reserves its slots with ONE atomic per queue, then writes the entries in a second pass with slots known.
// A tile is ITERS x QBLOCK rays: 8192 for large levels (few atomics), 2048 for small ones (a level of 300k rays would
// otherwise occupy 37 of the 256 compute units and take as long as one of 2M).
constexpr int QBLOCK = 1024;  // 16 wavefronts share a tile: 8 (2) rays per lane, short dependent chains
constexpr int TILE_ITERS = 8;
constexpr int TILE_ITERS_SMALL = 2;
struct TileSlots {
    int off[4][TILE_ITERS][QBLOCK / 64];  // pass 1: counts; after the scan: offsets inside the reservation
    long long base[3];                   // shadow queue, container queue, next level
};

template <int ITERS>
__global__ __launch_bounds__(QBLOCK) void k_queue(Level L, long long n_cap, const unsigned long long* n_dev, long long next_cap,
                                                 unsigned long long* counters, int32_t* __restrict__ next_tag) {
    constexpr int TILE = ITERS * QBLOCK;
    const long long n = level_count(n_cap, n_dev);
    if ((long long)blockIdx.x * TILE >= n) return;  // whole workgroup beyond the level
    __shared__ TileSlots ts;
    const long long tile0 = (long long)blockIdx.x * TILE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (ITERS < TILE_ITERS) {  // the scan below runs over all TILE_ITERS x 4 counts: the unused ones are zero
        int* flat = &ts.off[0][0][0];
        for (int q = 0; q < 4; ++q)
            for (int k = ITERS * (QBLOCK / 64) + threadIdx.x; k < TILE_ITERS * (QBLOCK / 64); k += QBLOCK) flat[q * TILE_ITERS * (QBLOCK / 64) + k] = 0;
    }
    int stage[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {  // independent coalesced loads, all in flight together
        long long i = tile0 + (long long)it * QBLOCK + threadIdx.x;
        stage[it] = i < n ? L.mode[i] : 0;
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        unsigned long long b0 = __ballot(stage[it] & ST_HIT), b1 = __ballot(stage[it] & ST_WALK);
        unsigned long long b2 = __ballot(stage[it] & ST_REFLECT), b3 = __ballot(stage[it] & ST_REFRACT);
        if (lane == 0) {
            ts.off[0][it][wave] = __popcll(b0); ts.off[1][it][wave] = __popcll(b1);
            ts.off[2][it][wave] = __popcll(b2); ts.off[3][it][wave] = __popcll(b3);
        }
    }
    __syncthreads();
    // exclusive scans of the per-(iteration, wavefront) counts, one wavefront per queue (a single thread walking the
    // 128 entries through LDS cost ~10 us per tile): lane l owns entries 2l and 2l+1
    {
        constexpr int PER = TILE_ITERS * (QBLOCK / 64) / 64;  // entries per lane
        static_assert(PER == 2, "k_queue's scan assumes 128 entries per queue");
        int* flat0 = &ts.off[0][0][0];
        auto wave_exclusive = [&](int v) {  // exclusive prefix sum over the wavefront; returns (prefix, total via reference)
            int inc = v;
            for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
            return inc - v;
        };
        if (wave < 2) {  // shadow queue, container queue
            int* f = flat0 + wave * TILE_ITERS * (QBLOCK / 64);
            const int c0 = f[2 * lane], c1 = f[2 * lane + 1];
            const int pre = wave_exclusive(c0 + c1);
            f[2 * lane] = pre; f[2 * lane + 1] = pre + c0;
            const int total = __shfl(pre + c0 + c1, 63, 64);
            if (lane == 0) ts.base[wave] = total ? (long long)atomicAdd(counters + (wave == 0 ? CNT_SHADOW_Q : CNT_CONTAINER_Q), (unsigned long long)total) : 0;
        } else if (wave == 2) {  // both ray kinds share the next level's queue: reflection rays of a wavefront first
            int* f2 = flat0 + 2 * TILE_ITERS * (QBLOCK / 64);
            int* f3 = flat0 + 3 * TILE_ITERS * (QBLOCK / 64);
            const int a0 = f2[2 * lane], b0 = f3[2 * lane], a1 = f2[2 * lane + 1], b1 = f3[2 * lane + 1];
            const int pre = wave_exclusive(a0 + b0 + a1 + b1);
            f2[2 * lane] = pre; f3[2 * lane] = pre + a0; f2[2 * lane + 1] = pre + a0 + b0; f3[2 * lane + 1] = pre + a0 + b0 + a1;
            const int total = __shfl(pre + a0 + b0 + a1 + b1, 63, 64);
            int n_refl = a0 + a1, n_refr = b0 + b1;
            for (int off = 32; off > 0; off >>= 1) { n_refl += __shfl_xor(n_refl, off, 64); n_refr += __shfl_xor(n_refr, off, 64); }
            if (lane == 0) {
                ts.base[2] = total ? (long long)atomicAdd(counters + CNT_NEXT, (unsigned long long)total) : 0;
                if (n_refl) atomicAdd(counters + CNT_REFLECT, (unsigned long long)n_refl);
                if (n_refr) atomicAdd(counters + CNT_REFRACT, (unsigned long long)n_refr);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        long long i = tile0 + (long long)it * QBLOCK + threadIdx.x;
        const int st = stage[it];
        unsigned long long b0 = __ballot(st & ST_HIT), b1 = __ballot(st & ST_WALK);
        unsigned long long b2 = __ballot(st & ST_REFLECT), b3 = __ballot(st & ST_REFRACT);
        if (st & ST_HIT) L.shadow_q[ts.base[0] + ts.off[0][it][wave] + __popcll(b0 & lanes_below())] = (int32_t)i;
        if (st & ST_WALK) L.cont_q[ts.base[1] + ts.off[1][it][wave] + __popcll(b1 & lanes_below())] = (int32_t)i;
        if (i < n) {
            long long s0 = ts.base[2] + ts.off[2][it][wave] + __popcll(b2 & lanes_below());
            long long s1 = ts.base[2] + ts.off[3][it][wave] + __popcll(b3 & lanes_below());
            const bool refl = (st & ST_REFLECT) && s0 < next_cap, refr = (st & ST_REFRACT) && s1 < next_cap;
            L.child_refl[i] = refl ? (int)s0 : -1;
            L.child_refr[i] = refr ? (int)s1 : -1;
            // the slots name their parent (spawn_ray): the next level's hit-key column is free until its k_trace
            if (refl) next_tag[s0] = (int32_t)(2 * i);
            if (refr) next_tag[s1] = (int32_t)(2 * i + 1);
        }
    }
}

// The n1/n2 container walk of prepare_computations (src/intersection.rs:54-80) for every queued transparent hit of a
// level, and with it the total-internal-reflection test of refracted_color (src/world.rs:125-130), so that k_spawn
// knows which refraction slots stay dead.  On the critical path of the ray tree (the refracted ray's direction needs
// n1/n2), unlike the shadow rays, hence a launch of its own between k_queue and k_spawn.
template <int SC, bool COUNT>
__global__ __launch_bounds__(TBLOCK, RT_TWAVES) void k_walk(DScene S, Level L, unsigned long long* counters) {
    const long long n_walk = (long long)counters[CNT_CONTAINER_Q];
    if (n_walk == 0) return;
    Feed walk_feed;
    const TravMem M = trav_setup(S, walk_feed, counters + CNT_FEED_CONTAINER, (n_walk + 63) / 64);
    Work w = {};
    for (long long c = walk_feed.next(); c >= 0; c = walk_feed.next()) {
        const long long j = c * 64 + (threadIdx.x & 63);
        const bool alive = j < n_walk;
        long long i = 0;
        V3 o = mk(0., 0., 0.), d = mk(0., 0., 1.);
        double hit_t = 0.;
        int hit_node = 0;
        if (alive) {
            i = L.cont_q[j];
            o = mk(L.ox[i], L.oy[i], L.oz[i]);
            d = mk(L.dx[i], L.dy[i], L.dz[i]);
            hit_t = L.ht[i];
            hit_node = L.hkey[i] >> 2;
        }
        unsigned flags = 0;
        double n1, n2;
        bool tir = false;
        container_walk<SC, COUNT>(S, M, o, d, hit_t, hit_node, w, n1, n2, flags, alive);
        if (alive) {
            L.hu[i] = n1;  // prepare has consumed the hit's u,v: the columns now carry n1, n2
            L.hv[i] = n2;
            {
                V3 eye_vector = vneg(d), normal_vector = mk(L.nx[i], L.ny[i], L.nz[i]);
                double n_ratio = n1 / n2;
                double cos_i = dot(eye_vector, normal_vector);
                double sin2_t = (n_ratio * n_ratio) * (1. - cos_i * cos_i);
                tir = sin2_t > 1. && (L.mode[i] & ST_REFRACT);
                if (tir) L.mode[i] |= ST_TIR;
            }
            if (flags) atomicOr(L.flags32 + (i >> 2), flags << (8 * (i & 3)));
        }
        const unsigned long long bt = __ballot(tir);  // refraction rays k_queue counted that will never fly
        if (bt && (threadIdx.x & 63) == 0) atomicAdd(counters + CNT_TIR, (unsigned long long)__popcll(bt));
    }
    if (COUNT) flush_work(w, counters + CNT_WC_XFORMS);
}

// spawn_ray for every slot of the next level, as a launch of its own (RT_AMD_FUSE_SPAWN=0: the A/B of k_trace's fused
// spawn).  One ray per lane, no atomics, no barriers.
__global__ __launch_bounds__(BLOCK) void k_spawn(Level L, Level N, long long n_cap, const unsigned long long* n_dev) {
    const long long n = level_count(n_cap, n_dev);
    const long long c = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (c >= n) return;
    V3 o = mk(dead_ray(), 0., 0.), d = mk(0., 0., 1.);
    spawn_ray(L, N.hkey[c], o, d);
    N.ox[c] = o.x; N.oy[c] = o.y; N.oz[c] = o.z;
    N.dx[c] = d.x; N.dy[c] = d.y; N.dz[c] = d.z;
}

// Bottom-up: shade_hit for every node of one level (src/world.rs:57-79) == Material::lighting over the lights (the
// shadow flags are in by now), schlick, and the blend with the children's resolved colours in the reference's own
// operand order; color_at's miss case -> BLACK.  Surface colour, coefficients and reflectance never leave registers.
// Held to 128 VGPRs = 4 wavefronts per SIMD (the second launch-bound is HIP's minimum wavefronts per execution unit):
// left alone the kernel takes 138 and runs 3; at 4 it spills 4 registers and is 6-8 % faster (same-box A/B), at 5, 6 and
// 8 the spills (38 / 48 / 66 registers) make it 1.7x, 1.9x and 2.6x slower.
#ifndef RT_SHADE_MIN_WAVES
#define RT_SHADE_MIN_WAVES 4
#endif
// Where level 0's resolved colours go when k_shade writes them out itself (OUT: what k_output does, without its launch
// and without reading the colours back): shard buffers (slot order) or, with `frame`, whole row-major frames.
struct OutArgs {
    double* rgb;
    int32_t* hit_id;
    uint8_t* flags;
    CameraParams cam;
    int frame;
};
__device__ __forceinline__ void write_output(const DScene& S, const OutArgs& O, long long i, double r, double g, double b, int key,
                                             unsigned flags) {
    long long o = i;
    if (O.frame) {
        int px, py;
        if (!slot_to_pixel(O.cam, i, px, py)) return;
        o = (long long)py * O.cam.hsize + px;
    }
    O.rgb[3 * o] = r; O.rgb[3 * o + 1] = g; O.rgb[3 * o + 2] = b;
    if (O.hit_id) O.hit_id[o] = key >= 0 ? S.node_object_id[key >> 2] : 0;
    if (O.flags) O.flags[o] = (uint8_t)flags;
}
template <bool OUT>
__global__ __launch_bounds__(BLOCK, RT_SHADE_MIN_WAVES) void k_shade(DScene S, Level L, Level N, long long n_cap, const unsigned long long* n_dev, int has_next,
                                                 unsigned long long* overflow_count, OutArgs O) {
    const long long n = level_count(n_cap, n_dev);
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int key = L.hkey[i];
    if (key < 0) {  // miss -> BLACK
        L.cr[i] = 0.; L.cg[i] = 0.; L.cb[i] = 0.;
        const unsigned miss_flags = L.flags[i];
        if (overflow_count && (miss_flags & RT_FLAG_LIST_OVERFLOW)) atomicAdd(overflow_count, 1ull);
        if (OUT) write_output(S, O, i, 0., 0., 0., key, miss_flags);
        return;
    }
    const int stage = L.mode[i];
    const int node = key >> 2;
    const V3 d = mk(L.dx[i], L.dy[i], L.dz[i]);
    const V3 point = mk(L.px[i], L.py[i], L.pz[i]), normal_vector = mk(L.nx[i], L.ny[i], L.nz[i]);
    const V3 eye_vector = vneg(d);
    const V3 over_point = over_point_of(point, normal_vector);
    const int mi = S.node_material[node];
    const double* __restrict__ m = S.mat + (size_t)mi * RT_MAT_STRIDE;
    const int pat = S.mat_pattern[mi];
    const double reflective = m[RT_MAT_REFLECTIVE], transparency = m[RT_MAT_TRANSPARENCY];
    const bool walked = stage & ST_WALK;
    const double n1 = walked ? L.hu[i] : 0., n2 = walked ? L.hv[i] : 0.;
    // src/world.rs:57-66
    V3 surface = mk(0., 0., 0.);
    for (int l = 0; l < S.n_lights; ++l)
        surface = vadd(surface, lighting(S, m, pat, node, l, over_point, eye_vector, normal_vector, L.shadowed[i * S.n_lights + l] != 0));
    V3 reflected = mk(0., 0., 0.), refracted = mk(0., 0., 0.);
    unsigned flags = L.flags[i];
    const int c0 = L.child_refl[i], c1 = L.child_refr[i];
    if (has_next && c0 >= 0) {
        reflected = vmul(mk(N.cr[c0], N.cg[c0], N.cb[c0]), reflective);
        flags |= N.flags[c0];
    }
    if (has_next && c1 >= 0) {
        refracted = vmul(mk(N.cr[c1], N.cg[c1], N.cb[c1]), transparency);
        flags |= N.flags[c1];
    }
    V3 out;
    if (reflective > 0. && transparency > 0.) {  // src/world.rs:71-75
        const double reflectance = walked ? schlick(eye_vector, normal_vector, n1, n2) : 0.;  // see needs_walk
        out = vadd(vadd(surface, vmul(reflected, reflectance)), vmul(refracted, (1. - reflectance)));
    } else {
        out = vadd(vadd(surface, reflected), refracted);
    }
    L.cr[i] = out.x; L.cg[i] = out.y; L.cb[i] = out.z;
    L.flags[i] = (uint8_t)flags;
    // a level-0 ray whose tree ran into a bounded list: the host traces those again with the BIG variants
    if (overflow_count && (flags & RT_FLAG_LIST_OVERFLOW)) atomicAdd(overflow_count, 1ull);
    if (OUT) write_output(S, O, i, out.x, out.y, out.z, key, flags);
}

// Level-0 rays whose resolved flags carry RT_FLAG_LIST_OVERFLOW: their indices (any order) and rays, for the fix-up pass.
__global__ __launch_bounds__(BLOCK) void k_collect_overflow(Level L, long long n, unsigned long long* cursor, int32_t* index_out,
                                                            Level F) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || !(L.flags[i] & RT_FLAG_LIST_OVERFLOW) || is_dead_ray(L.ox[i])) return;
    const long long j = (long long)atomicAdd(cursor, 1ull);
    index_out[j] = (int32_t)i;
    F.ox[j] = L.ox[i]; F.oy[j] = L.oy[i]; F.oz[j] = L.oz[i];
    F.dx[j] = L.dx[i]; F.dy[j] = L.dy[i]; F.dz[j] = L.dz[i];
}
// ... and their re-traced results back into place
__global__ __launch_bounds__(BLOCK) void k_patch_overflow(Level L, Level F, long long n_fix, const int32_t* __restrict__ index) {
    const long long j = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_fix) return;
    const long long i = index[j];
    L.cr[i] = F.cr[j]; L.cg[i] = F.cg[j]; L.cb[i] = F.cb[j];
    L.hkey[i] = F.hkey[j];
    L.flags[i] = F.flags[j];
}

// Kernel: level-0 colours -> caller's buffers, interleaved RGB.  frame == 0: shard-local order; frame == 1:
// scatter into the full row-major frame (Canvas layout, src/canvas.rs:25-27).
// `chains` > 1: this level holds sub-shard `chain` of the shard (every chains-th 8x8 tile); its 64-slot tiles are
// written to their interleaved places in the shard buffer.
// `frame` != 0: the outputs are whole row-major frames (Canvas layout) and slot i goes to its pixel — `cam` is the
// camera of the (sub-)shard this level holds; padding slots write nothing.
__global__ __launch_bounds__(BLOCK) void k_output(DScene S, Level L, long long n, double* rgb, int32_t* hit_id, uint8_t* flags,
                                                  int chains, int chain, CameraParams cam, int frame) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    long long o = ((i >> 6) * chains + chain) * 64 + (i & 63);
    if (frame) {
        int px, py;
        if (!slot_to_pixel(cam, i, px, py)) return;
        o = (long long)py * cam.hsize + px;
    }
    rgb[3 * o] = L.cr[i]; rgb[3 * o + 1] = L.cg[i]; rgb[3 * o + 2] = L.cb[i];
    if (hit_id) {
        int key = L.hkey[i];
        hit_id[o] = key >= 0 ? S.node_object_id[key >> 2] : 0;
    }
    if (flags) flags[o] = L.flags[i];
}

__global__ __launch_bounds__(BLOCK) void k_assemble(CameraParams c, long long n_slots, const double* __restrict__ rgb_in,
                                                    const int32_t* __restrict__ id_in, const uint8_t* __restrict__ fl_in,
                                                    double* rgb_out, int32_t* id_out, uint8_t* fl_out) {
    long long slot = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= n_slots) return;
    int px, py;
    if (!slot_to_pixel(c, slot, px, py)) return;
    long long p = (long long)py * c.hsize + px;
    rgb_out[3 * p] = rgb_in[3 * slot]; rgb_out[3 * p + 1] = rgb_in[3 * slot + 1]; rgb_out[3 * p + 2] = rgb_in[3 * slot + 2];
    if (id_out && id_in) id_out[p] = id_in[slot];
    if (fl_out && fl_in) fl_out[p] = fl_in[slot];
}

// The same for every per-pixel array a multi-GPU gather brings to the root (rt_render_multi): f64 RGB, quantised u8 RGB,
// hit ids, flags; null inputs are skipped.
__global__ __launch_bounds__(BLOCK) void k_assemble_all(CameraParams c, long long n_slots, const double* __restrict__ rgb_in,
                                                        const uint8_t* __restrict__ q_in, const int32_t* __restrict__ id_in,
                                                        const uint8_t* __restrict__ fl_in, double* rgb_out, uint8_t* q_out, int32_t* id_out,
                                                        uint8_t* fl_out) {
    long long slot = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (slot >= n_slots) return;
    int px, py;
    if (!slot_to_pixel(c, slot, px, py)) return;
    long long p = (long long)py * c.hsize + px;
    if (rgb_in) { rgb_out[3 * p] = rgb_in[3 * slot]; rgb_out[3 * p + 1] = rgb_in[3 * slot + 1]; rgb_out[3 * p + 2] = rgb_in[3 * slot + 2]; }
    if (q_in) { q_out[3 * p] = q_in[3 * slot]; q_out[3 * p + 1] = q_in[3 * slot + 1]; q_out[3 * p + 2] = q_in[3 * slot + 2]; }
    if (id_in) id_out[p] = id_in[slot];
    if (fl_in) fl_out[p] = fl_in[slot];
}

// Kernel: World::intersect_world as a sorted list per ray (rt_intersect_world).
template <int SC>
__global__ __launch_bounds__(TBLOCK, RT_TWAVES) void k_list(DScene S, long long n, const double* __restrict__ origins,
                                                 const double* __restrict__ dirs, int max_hits, int32_t* count, double* t_out,
                                                 int32_t* id_out, uint8_t* flags_out) {
    constexpr bool BIG = (SC & SC_BIG) != 0;
    Feed feed;
    const TravMem M = trav_setup(S, feed, nullptr, (n + 63) / 64);
    for (long long c = feed.next(); c >= 0; c = feed.next()) {
        const long long i = c * 64 + (threadIdx.x & 63);
        const bool alive = i < n;
        double lt_private[BIG ? 1 : LIST_CAP];
        int lk_private[BIG ? 1 : LIST_CAP];
        double* lt = lt_private;
        int* lk = lk_private;
        int list_cap = LIST_CAP;
        if (BIG) {
            list_cap = S.arena_list_cap;
            lt = reinterpret_cast<double*>(arena_list(S));
            lk = reinterpret_cast<int*>(lt + list_cap);
        }
        ListSink sink;
        sink.t = lt; sink.key = lk; sink.n = 0; sink.cap = list_cap; sink.limit = INFINITY; sink.flags = 0;
        Work w = {};
        V3 o = mk(0., 0., 0.), d = mk(0., 0., 1.);
        if (alive) {
            o = mk(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
            d = mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        }
        traverse<SC, false>(S, M, o, d, sink, w, alive);
        if (!alive) continue;
        int m = sink.n < list_cap ? sink.n : list_cap;
        sort_list(lt, lk, m);
        count[i] = sink.n;
        flags_out[i] = (uint8_t)sink.flags;
        for (int k = 0; k < m && k < max_hits; ++k) {
            t_out[i * max_hits + k] = lt[k];
            id_out[i * max_hits + k] = S.node_object_id[lk[k] >> 2];
        }
    }
}

// Kernel: World::is_shadowed for a batch of points (rt_is_shadowed).
template <int SC>
__global__ __launch_bounds__(TBLOCK, RT_TWAVES) void k_shadow_points(DScene S, long long n, const double* __restrict__ points, uint8_t* out,
                                                          uint8_t* flags_out) {
    Feed feed;
    const TravMem M = trav_setup(S, feed, nullptr, (n + 63) / 64);
    for (long long c = feed.next(); c >= 0; c = feed.next()) {
        const long long i = c * 64 + (threadIdx.x & 63);
        const bool alive = i < n;
        Work w = {};
        unsigned flags = 0;
        V3 p = alive ? mk(points[3 * i], points[3 * i + 1], points[3 * i + 2]) : mk(0., 0., 0.);
        for (int l = 0; l < S.n_lights; ++l) {
            bool sh = is_shadowed<SC, false>(S, M, p, l, w, flags, alive);
            if (alive) out[i * S.n_lights + l] = sh ? 1 : 0;
        }
        if (alive) flags_out[i] = (uint8_t)flags;
    }
}

// Kernel: Intersection::prepare_computations (src/intersection.rs:38-95) and Intersection::schlick (:97-115) for the hit
// of each ray, every field of `Computations` (src/computations.rs:6-18) handed back (rt_prepare_computations).  The
// same device functions the frame pipeline runs — traverse, normal_at_wave, container_walk, schlick — with nothing
// skipped: the n1/n2 walk runs for every hit, as in the reference, whatever the material.
template <int SC>
__global__ __launch_bounds__(TBLOCK, RT_TWAVES) void k_comps(DScene S, long long n, const double* __restrict__ origins,
                                                            const double* __restrict__ dirs, int32_t* hit_id_out, double* out, uint8_t* flags_out) {
    Feed feed;
    const TravMem M = trav_setup(S, feed, nullptr, (n + 63) / 64);
    for (long long c = feed.next(); c >= 0; c = feed.next()) {
        const long long i = c * 64 + (threadIdx.x & 63);
        const bool alive = i < n;
        Work w = {};
        V3 o = mk(0., 0., 0.), d = mk(0., 0., 1.);
        if (alive) {
            o = mk(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
            d = mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        }
        ClosestSink<true> sink;
        sink.init(INFINITY);
        traverse<SC, false>(S, M, o, d, sink, w, alive);
        const int key = alive ? sink.best_key : -1;
        const bool hit = key >= 0;
        const int node = hit ? key >> 2 : 0;
        NodeRec rec = {0, 0, 0, 0};
        if (hit) rec = S.node_rec[node];
        rec.kind &= 0xff;
        double u = sink.bu, v = sink.bv;
        unsigned flags = alive ? sink.flags : 0;
        if (hit && rec.kind == RT_SMOOTH_TRIANGLE) {
            if (sink.first_uv) { u = sink.fu; v = sink.fv; }  // u,v of intersections[0] (F5)
            else flags |= RT_FLAG_REF_PANIC_UV;
        }
        const V3 point = vadd(o, vmul(d, sink.best_t));
        V3 normal_vector = normal_at_wave(S, hit, rec, point, u, v);
        const V3 eye_vector = vneg(d);
        const bool inside = hit && dot(normal_vector, eye_vector) < 0.;
        if (inside) normal_vector = vneg(normal_vector);
        double n1 = 0., n2 = 0.;
        container_walk<SC, false>(S, M, o, d, hit ? sink.best_t : 0., node, w, n1, n2, flags, hit);
        if (!alive) continue;
        hit_id_out[i] = hit ? S.node_object_id[node] : 0;
        flags_out[i] = (uint8_t)flags;
        double* r = out + i * RT_COMP_COUNT;
        if (!hit) {
            for (int k = 0; k < RT_COMP_COUNT; ++k) r[k] = 0.;
            continue;
        }
        const V3 over_point = over_point_of(point, normal_vector), under_point = under_point_of(point, normal_vector);
        const V3 reflect_vector = reflect(d, normal_vector);
        r[RT_COMP_T] = sink.best_t;
        r[RT_COMP_POINT] = point.x; r[RT_COMP_POINT + 1] = point.y; r[RT_COMP_POINT + 2] = point.z;
        r[RT_COMP_EYE] = eye_vector.x; r[RT_COMP_EYE + 1] = eye_vector.y; r[RT_COMP_EYE + 2] = eye_vector.z;
        r[RT_COMP_NORMAL] = normal_vector.x; r[RT_COMP_NORMAL + 1] = normal_vector.y; r[RT_COMP_NORMAL + 2] = normal_vector.z;
        r[RT_COMP_INSIDE] = inside ? 1. : 0.;
        r[RT_COMP_OVER] = over_point.x; r[RT_COMP_OVER + 1] = over_point.y; r[RT_COMP_OVER + 2] = over_point.z;
        r[RT_COMP_UNDER] = under_point.x; r[RT_COMP_UNDER + 1] = under_point.y; r[RT_COMP_UNDER + 2] = under_point.z;
        r[RT_COMP_REFLECT] = reflect_vector.x; r[RT_COMP_REFLECT + 1] = reflect_vector.y; r[RT_COMP_REFLECT + 2] = reflect_vector.z;
        r[RT_COMP_N1] = n1;
        r[RT_COMP_N2] = n2;
        r[RT_COMP_SCHLICK] = schlick(eye_vector, normal_vector, n1, n2);
        r[RT_COMP_U] = u;
        r[RT_COMP_V] = v;
    }
}

// Kernel: Shape::normal_at (src/shape.rs:92-108) of one leaf node for a batch of world points (rt_normal_at); `uv` [n][2]
// or null: the u, v of the intersection handed to SmoothTriangle::normal_at.  Whole wavefronts execute (normal_at_wave).
__global__ __launch_bounds__(BLOCK) void k_normal_at(DScene S, int node, long long n, const double* __restrict__ points,
                                                     const double* __restrict__ uv, double* out) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    const bool on = i < n;
    NodeRec rec = S.node_rec[node];
    rec.kind &= 0xff;
    const V3 p = on ? mk(points[3 * i], points[3 * i + 1], points[3 * i + 2]) : mk(0., 0., 0.);
    const double u = on && uv ? uv[2 * i] : 0., v = on && uv ? uv[2 * i + 1] : 0.;
    const V3 nv = normal_at_wave(S, on, rec, p, u, v);
    if (on) { out[3 * i] = nv.x; out[3 * i + 1] = nv.y; out[3 * i + 2] = nv.z; }
}

// Kernel: Material::lighting (src/material.rs:97-129) with the material (and pattern, evaluated in the leaf's object
// space: Pattern::pattern_at_shape, src/pattern.rs:23-35) of one leaf node, for a batch of (point, eye, normal, in_shadow).
__global__ __launch_bounds__(BLOCK) void k_lighting(DScene S, int node, int light, long long n, const double* __restrict__ points,
                                                    const double* __restrict__ eyes, const double* __restrict__ normals,
                                                    const uint8_t* __restrict__ in_shadow, double* out) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const int mi = S.node_material[node];
    const V3 c = lighting(S, S.mat + (size_t)mi * RT_MAT_STRIDE, S.mat_pattern[mi], node, light, mk(points[3 * i], points[3 * i + 1], points[3 * i + 2]),
                          mk(eyes[3 * i], eyes[3 * i + 1], eyes[3 * i + 2]), mk(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]),
                          in_shadow[i] != 0);
    out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
}

// ---------------------------------------------------------------------------------------------------------
// Canvas::canvas_to_ppm on the device (src/canvas.rs:33-98).  Every value is followed by exactly one separator
// (' ', or '\n' where the reference wraps or ends the row), so a row's byte length is the sum of (digits + 1) over
// its 3*w values whatever the wrapping: row offsets come from a scan, only the choice of separator is sequential
// (greedy line fill restarted at every row).

__device__ __forceinline__ int scale_color(double color_value) {  // src/canvas.rs:86-98
    if (color_value > 1.0) return 255;
    if (color_value < 0.0) return 0;
    double scaled = color_value * 255.;
    if (scaled != scaled) return 0;  // `as i32` maps NaN to 0
    return (int)round(scaled);       // f64::round: half away from zero
}

__global__ __launch_bounds__(BLOCK) void k_quantise(const double* __restrict__ rgb, long long n, uint8_t* __restrict__ q) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) q[i] = (uint8_t)scale_color(rgb[i]);
}

__device__ __forceinline__ int digits_of(int v) { return v >= 100 ? 3 : v >= 10 ? 2 : 1; }

// one thread per canvas row: bytes of the row's text
__global__ __launch_bounds__(BLOCK) void k_ppm_row_bytes(const uint8_t* __restrict__ q, int values_per_row, int rows, long long* __restrict__ bytes) {
    int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= rows) return;
    const uint8_t* __restrict__ p = q + (size_t)r * values_per_row;
    long long total = 0;
    for (int k = 0; k < values_per_row; ++k) total += digits_of(p[k]) + 1;
    bytes[r] = total;
}

// single workgroup: exclusive scan of the row sizes in place; bytes[rows] receives the total
__global__ __launch_bounds__(BLOCK) void k_ppm_scan(long long* __restrict__ bytes, int rows) {
    __shared__ long long part[BLOCK];
    const int per = (rows + BLOCK - 1) / BLOCK;
    const int lo = threadIdx.x * per, hi = min(lo + per, rows);
    long long sum = 0;
    for (int r = lo; r < hi; ++r) sum += bytes[r];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int k = 0; k < BLOCK; ++k) { long long c = part[k]; part[k] = run; run += c; }
        bytes[rows] = run;
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (int r = lo; r < hi; ++r) { long long c = bytes[r]; bytes[r] = run; run += c; }
}

// one thread per canvas row: the row's text at its offset
// `body_cap` >= 0: the text buffer holds that many bytes; a body that does not fit is not written at all (the row
// scan's total, offsets[rows], is already known on the device — the entry point that allocates nothing cannot ask the
// host first).  < 0: the caller has sized the buffer from the total.
__global__ __launch_bounds__(BLOCK) void k_ppm_write(const uint8_t* __restrict__ q, int values_per_row, int rows,
                                                     const long long* __restrict__ offsets, char* __restrict__ text,
                                                     long long body_cap) {
    int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= rows) return;
    if (body_cap >= 0 && offsets[rows] > body_cap) return;
    const uint8_t* __restrict__ p = q + (size_t)r * values_per_row;
    char* out = text + offsets[r];
    int line = 0;  // characters in the current line, trailing space included (String::len in the reference)
    for (int k = 0; k < values_per_row; ++k) {
        const int v = p[k], nd = digits_of(v);
        if (k > 0) {
            if (line + nd >= 70) { *out++ = '\n'; line = 0; }  // the trailing space becomes the line break
            else *out++ = ' ';
        }
        if (nd == 3) *out++ = char('0' + v / 100);
        if (nd >= 2) *out++ = char('0' + (v / 10) % 10);
        *out++ = char('0' + v % 10);
        line += nd + 1;
    }
    *out++ = '\n';
}

// The PPM header "P3\n{w} {h}\n255\n" written on the device (rt_canvas_to_ppm_noalloc: the whole text stays in HBM), and
// the total length of the text for callers that read it in stream order.  One thread.
__global__ void k_ppm_header(int hsize, int vsize, const long long* __restrict__ offsets, int rows, char* __restrict__ text,
                             long long cap, long long* __restrict__ len_out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    char h[48];
    int n = 0;
    auto put_int = [&](int v) {
        char d[12];
        int k = 0;
        do { d[k++] = char('0' + v % 10); v /= 10; } while (v > 0);
        while (k > 0) h[n++] = d[--k];
    };
    h[n++] = 'P'; h[n++] = '3'; h[n++] = '\n';
    put_int(hsize); h[n++] = ' '; put_int(vsize); h[n++] = '\n';
    h[n++] = '2'; h[n++] = '5'; h[n++] = '5'; h[n++] = '\n';
    const long long total = n + offsets[rows];
    if (len_out) *len_out = total;
    if (total <= cap)
        for (int k = 0; k < n; ++k) text[k] = h[k];
}

// Kernel: load caller rays into level 0 (rt_color_at).
__global__ __launch_bounds__(BLOCK) void k_load_rays(Level L, long long n, const double* __restrict__ origins,
                                                     const double* __restrict__ dirs) {
    long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    L.ox[i] = origins[3 * i]; L.oy[i] = origins[3 * i + 1]; L.oz[i] = origins[3 * i + 2];
    L.dx[i] = dirs[3 * i]; L.dy[i] = dirs[3 * i + 1]; L.dz[i] = dirs[3 * i + 2];
}

}  // namespace rt
