// capi.cpp — extern "C" handle API (include/rt_host.h) over the C++ host mirror.
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/rt_host.h"
#include "../csrc/rt_internal.h"
#include "ray_tracer.hpp"

using namespace ray_tracer;

namespace {

struct WorldSlot {
    World world;
    std::unique_ptr<FlatScene> flat;
    rt_scene_desc desc;
};

std::vector<std::unique_ptr<Shape>>& shapes() {
    static std::vector<std::unique_ptr<Shape>> v;
    return v;
}
std::vector<std::unique_ptr<WorldSlot>>& worlds() {
    static std::vector<std::unique_ptr<WorldSlot>> v;
    return v;
}

int put_shape(Shape&& s) {
    auto& v = shapes();
    for (size_t i = 0; i < v.size(); ++i)
        if (!v[i]) {
            v[i] = std::make_unique<Shape>(std::move(s));
            return int(i);
        }
    v.push_back(std::make_unique<Shape>(std::move(s)));
    return int(v.size() - 1);
}
Shape& get_shape(int h) {
    auto& v = shapes();
    if (h < 0 || size_t(h) >= v.size() || !v[size_t(h)]) throw std::out_of_range("bad shape handle");
    return *v[size_t(h)];
}
WorldSlot& get_world(int h) {
    auto& v = worlds();
    if (h < 0 || size_t(h) >= v.size() || !v[size_t(h)]) throw std::out_of_range("bad world handle");
    return *v[size_t(h)];
}

Matrix mat_from(const double* m16) { return Matrix(4, 4, std::vector<double>(m16, m16 + 16)); }
void mat_to(const Matrix& m, double* out16) { std::memcpy(out16, m.data(), 16 * sizeof(double)); }
Tuple pt(const double* p) { return Tuple::point(p[0], p[1], p[2]); }
Tuple vec(const double* p) { return Tuple::vector(p[0], p[1], p[2]); }

Material mat_in(const rth_material* m) {
    Material r;
    r.color = Color{m->color[0], m->color[1], m->color[2]};
    r.ambient = m->ambient;
    r.diffuse = m->diffuse;
    r.specular = m->specular;
    r.shininess = m->shininess;
    r.reflective = m->reflective;
    r.transparency = m->transparency;
    r.refractive_index = m->refractive_index;
    if (m->pattern_kind >= 0) {
        if (m->pattern_kind > RT_PAT_TEST) throw std::invalid_argument("bad pattern kind");
        Pattern p{PatternKind(m->pattern_kind), Color{m->pattern_a[0], m->pattern_a[1], m->pattern_a[2]},
                  Color{m->pattern_b[0], m->pattern_b[1], m->pattern_b[2]}};
        p.transform = mat_from(m->pattern_transform);
        r.pattern = p;
    }
    return r;
}
void mat_out(const Material& r, rth_material* m) {
    std::memset(m, 0, sizeof *m);
    m->color[0] = r.color.red; m->color[1] = r.color.green; m->color[2] = r.color.blue;
    m->ambient = r.ambient; m->diffuse = r.diffuse; m->specular = r.specular; m->shininess = r.shininess;
    m->reflective = r.reflective; m->transparency = r.transparency; m->refractive_index = r.refractive_index;
    m->pattern_kind = -1;
    rth_identity(m->pattern_transform);
    if (r.pattern) {
        m->pattern_kind = int32_t(r.pattern->kind);
        m->pattern_a[0] = r.pattern->a.red; m->pattern_a[1] = r.pattern->a.green; m->pattern_a[2] = r.pattern->a.blue;
        m->pattern_b[0] = r.pattern->b.red; m->pattern_b[1] = r.pattern->b.green; m->pattern_b[2] = r.pattern->b.blue;
        mat_to(r.pattern->transform, m->pattern_transform);
    }
}
void box_out(const Bound& b, double* o) {
    o[0] = b.minimum.x; o[1] = b.minimum.y; o[2] = b.minimum.z;
    o[3] = b.maximum.x; o[4] = b.maximum.y; o[5] = b.maximum.z;
}

// Runs f; maps exceptions to error codes (nothing unwinds across the C boundary).
template <class F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const Panic& p) {
        rt_set_error("reference panic: %s", p.what());
        return RT_ERR_UNSUPPORTED;
    } catch (const std::exception& e) {
        rt_set_error("%s", e.what());
        return RT_ERR_BAD_ARG;
    } catch (...) {
        rt_set_error("unknown C++ exception");
        return RT_ERR_BAD_ARG;
    }
}

}  // namespace

extern "C" {

void rth_identity(double* o) { mat_to(Matrix::identity(4), o); }
void rth_translate(double x, double y, double z, double* o) { mat_to(translate(x, y, z), o); }
void rth_scale(double x, double y, double z, double* o) { mat_to(scale(x, y, z), o); }
void rth_rotate(int axis, double radians, double* o) {
    mat_to(rotate(radians, axis == 0 ? Axis::X : axis == 1 ? Axis::Y : Axis::Z), o);
}
void rth_shear(double xy, double xz, double yx, double yz, double zx, double zy, double* o) {
    mat_to(shear(xy, xz, yx, yz, zx, zy), o);
}
void rth_view_transform(const double* from3, const double* to3, const double* up3, double* o) {
    mat_to(view_transform(pt(from3), pt(to3), vec(up3)), o);
}
void rth_matmul(const double* a, const double* b, double* o) { mat_to(mat_from(a) * mat_from(b), o); }
int rth_inverse(const double* m, double* o) {
    std::optional<Matrix> inv = mat_from(m).inverse();
    if (!inv) return 0;
    mat_to(*inv, o);
    return 1;
}

void rth_material_default(rth_material* out) { mat_out(Material(), out); }

int rth_shape_new(int kind) {
    return guarded([&] {
        switch (kind) {
            case RT_SPHERE: return put_shape(Shape::Sphere());
            case RT_PLANE: return put_shape(Shape::Plane());
            case RT_CUBE: return put_shape(Shape::Cube());
            case RT_CYLINDER: return put_shape(Shape::Cylinder());
            case RT_CONE: return put_shape(Shape::Cone());
            case RT_GROUP: return put_shape(Shape::Group());
            default: throw std::invalid_argument("rth_shape_new: kind needs its own constructor");
        }
    });
}
int rth_glass_sphere(void) { return guarded([&] { return put_shape(Shape::GlassSphere()); }); }
int rth_triangle_new(const double* p1, const double* p2, const double* p3) {
    return guarded([&] { return put_shape(Shape::Triangle(pt(p1), pt(p2), pt(p3))); });
}
int rth_smooth_triangle_new(const double* p1, const double* p2, const double* p3, const double* n1, const double* n2,
                            const double* n3) {
    return guarded(
        [&] { return put_shape(Shape::SmoothTriangle(pt(p1), pt(p2), pt(p3), vec(n1), vec(n2), vec(n3))); });
}
int rth_csg_new(const char* operation, int left, int right) {
    return guarded([&] { return put_shape(Shape::CSG(operation, get_shape(left), get_shape(right))); });
}
int rth_hexagon(void) { return guarded([&] { return put_shape(hexagon()); }); }
int rth_shape_clone(int h) { return guarded([&] { return put_shape(Shape(get_shape(h))); }); }
void rth_shape_free(int h) {
    auto& v = shapes();
    if (h >= 0 && size_t(h) < v.size()) v[size_t(h)].reset();
}
int rth_shape_kind(int h) { return guarded([&] { return int(get_shape(h).kind()); }); }
int rth_shape_id(int h) { return guarded([&] { return get_shape(h).get_id(); }); }
int rth_shape_parent(int h) { return guarded([&] { return get_shape(h).get_parent().value_or(0); }); }
int rth_shape_set_transform(int h, const double* m16) {
    return guarded([&] { get_shape(h).set_transform(mat_from(m16)); return 0; });
}
int rth_shape_set_material(int h, const rth_material* m) {
    return guarded([&] { get_shape(h).set_material(mat_in(m)); return 0; });
}
int rth_shape_get_material(int h, rth_material* out) {
    return guarded([&] { mat_out(get_shape(h).get_material(), out); return 0; });
}
int rth_shape_set_casts_shadow(int h, int v) { return guarded([&] { get_shape(h).set_casts_shadow(v != 0); return 0; }); }
int rth_shape_set_minimum(int h, double v) { return guarded([&] { get_shape(h).set_minimum(v); return 0; }); }
int rth_shape_set_maximum(int h, double v) { return guarded([&] { get_shape(h).set_maximum(v); return 0; }); }
int rth_shape_set_closed(int h, int v) { return guarded([&] { get_shape(h).set_closed(v != 0); return 0; }); }
int rth_group_add_child(int g, int c) { return guarded([&] { get_shape(g).add_child(get_shape(c)); return 0; }); }
int rth_shape_child_count(int h) { return guarded([&] { return int(get_shape(h).get_shapes().size()); }); }
int rth_shape_child(int h, int index) {
    return guarded([&] { return put_shape(Shape(get_shape(h).get_shapes().at(size_t(index)))); });
}
int rth_shape_divide(int h, int threshold) {
    return guarded([&] { get_shape(h).divide(size_t(threshold)); return 0; });
}
int rth_shape_bounds_of(int h, double* out6) { return guarded([&] { box_out(get_shape(h).bounds_of(), out6); return 0; }); }
int rth_shape_parent_space_bounds_of(int h, double* out6) {
    return guarded([&] { box_out(get_shape(h).parent_space_bounds_of(), out6); return 0; });
}
int rth_shape_includes(int h, int other) { return guarded([&] { return get_shape(h).includes(get_shape(other)) ? 1 : 0; }); }
int rth_shape_eq(int a, int b) { return guarded([&] { return get_shape(a) == get_shape(b) ? 1 : 0; }); }
int rth_triangle_data(int h, double* o) {
    return guarded([&] {
        const Shape& s = get_shape(h);
        if (s.kind() != ShapeKind::Triangle && s.kind() != ShapeKind::SmoothTriangle)
            throw Panic("Points are available only for triangles.");
        const Tuple* t[9] = {&s.point1, &s.point2, &s.point3, &s.edge_vector1, &s.edge_vector2, &s.normal_vector,
                             &s.normal_vector1, &s.normal_vector2, &s.normal_vector3};
        for (int i = 0; i < 9; ++i) { o[3 * i] = t[i]->x; o[3 * i + 1] = t[i]->y; o[3 * i + 2] = t[i]->z; }
        return 0;
    });
}

int rth_obj_to_group(const char* text, int64_t len, int32_t* counts_out) {
    return guarded([&] {
        Parser p = parse_obj_file(std::string(text, size_t(len)));
        if (counts_out) {
            counts_out[0] = p.lines_ignored;
            counts_out[1] = int32_t(p.vertices.size());
            counts_out[2] = int32_t(p.normals.size());
            counts_out[3] = int32_t(p.groups.size());
            int32_t tris = 0;
            for (const Shape& g : p.groups) tris += int32_t(g.get_shapes().size());
            counts_out[4] = tris;
        }
        return put_shape(obj_to_group(std::move(p)));
    });
}

int rth_obj_file_to_group(const char* text, int64_t len, int32_t* counts_out) {
    return guarded([&] { return put_shape(obj_file_to_group(std::string(text, size_t(len)), counts_out)); });
}

int rth_world_create(const int* shape_handles, int n_shapes, const double* lp, const double* li, int n_lights) {
    return guarded([&] {
        auto slot = std::make_unique<WorldSlot>();
        for (int i = 0; i < n_shapes; ++i) slot->world.objects.push_back(get_shape(shape_handles[i]));
        for (int i = 0; i < n_lights; ++i)
            slot->world.lights.push_back(Light::point_light(pt(lp + 3 * i), Color{li[3 * i], li[3 * i + 1], li[3 * i + 2]}));
        auto& v = worlds();
        for (size_t i = 0; i < v.size(); ++i)
            if (!v[i]) { v[i] = std::move(slot); return int(i); }
        v.push_back(std::move(slot));
        return int(v.size() - 1);
    });
}
void rth_world_free(int w) {
    auto& v = worlds();
    if (w >= 0 && size_t(w) < v.size()) v[size_t(w)].reset();
}
const rt_scene_desc* rth_world_flatten(int w) {
    const rt_scene_desc* out = nullptr;
    guarded([&] {
        WorldSlot& s = get_world(w);
        if (!s.flat) {
            s.flat = std::make_unique<FlatScene>(s.world.flatten());
            s.desc = s.flat->desc();
        }
        out = &s.desc;
        return 0;
    });
    return out;
}
rt_scene* rth_world_scene(int w, int device) {
    rt_scene* out = nullptr;
    guarded([&] {
        const rt_scene_desc* d = rth_world_flatten(w);
        if (!d) return 0;
        WorldSlot& s = get_world(w);
        // reuse World's own lazily-built device copy so World::color_at & co. share it
        (void)d;
        out = s.world.device_scene(device);
        return 0;
    });
    return out;
}

int rth_camera_desc(int hsize, int vsize, double fov, const double* transform16, rt_camera_desc* out) {
    return guarded([&] {
        if (hsize <= 0 || vsize <= 0) throw std::invalid_argument("camera size must be positive");
        Camera c{size_t(hsize), size_t(vsize), fov};
        c.transform = mat_from(transform16);
        *out = c.desc();
        return 0;
    });
}

int64_t rth_canvas_to_ppm(const double* rgb, int width, int height, char* out, int64_t cap) {
    int64_t n = -1;
    guarded([&] {
        Canvas c{size_t(width), size_t(height)};
        std::vector<Color>& px = c.pixels();
        for (size_t i = 0; i < px.size(); ++i) px[i] = Color{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        std::string s = c.canvas_to_ppm();
        n = int64_t(s.size());
        if (out && cap > 0) std::memcpy(out, s.data(), size_t(n < cap ? n : cap));
        return 0;
    });
    return n;
}

}  // extern "C"
