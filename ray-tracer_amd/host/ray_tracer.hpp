// ray_tracer.hpp — C++ host-side mirror of the reference crate's scene-construction API.
//
// The reference (ashcolecarr/Ray-Tracer) is a Rust crate; there is no Rust toolchain in the build image, so
// the host side above the C ABI (include/rt_abi.h) is written in C++ with the same type names, method names,
// argument meaning and error behaviour as the crate's public items.  What is mirrored here is everything a
// caller needs to BUILD a World and a Camera (tuple.rs, color.rs, matrix.rs, transformation.rs, material.rs,
// pattern.rs, light.rs, shape.rs + the nine shape modules, group.rs, csg.rs, bound.rs, obj_file.rs, canvas.rs,
// world.rs, camera.rs).  What is NOT here, on purpose, is any CPU implementation of the hot path
// (Shape::intersect, normal_at, World::color_at, shade_hit, ...): World and Camera forward those calls to the
// device through the C ABI after flattening the scene (flatten.cpp).  There is no CPU fallback.
//
// Value semantics follow Rust: a Shape is a value, `clone` is a deep copy, Group::add_child stores a CLONE of
// the child (src/group.rs:45-50) and the process-global id registry (src/lib.rs:46-49) keeps the latest version
// of every Group/CSG for parent-chain look-ups.
#pragma once

#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_abi.h"

namespace ray_tracer {

constexpr double EPSILON = 0.00001;      // src/lib.rs:42
constexpr int DEFAULT_RECURSION = 5;     // src/lib.rs:41
constexpr double INF = std::numeric_limits<double>::infinity();

inline bool near_eq(double a, double b) { return std::fabs(a - b) < EPSILON; }  // src/lib.rs:56-62

int generate_object_id();  // src/lib.rs:64-66
int generate_matrix_id();  // src/lib.rs:68-70

// A Rust panic at the same place in the reference becomes this exception; the C layers turn it into an error code.
struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------- tuple.rs
struct Tuple {
    double x = 0, y = 0, z = 0, w = 0;
    static Tuple point(double x, double y, double z) { return {x, y, z, 1.}; }
    static Tuple vector(double x, double y, double z) { return {x, y, z, 0.}; }
    double magnitude() const { return std::sqrt(x * x + y * y + z * z + w * w); }
    Tuple normalize() const {
        return {x / magnitude(), y / magnitude(), z / magnitude(), w / magnitude()};
    }
    double dot(const Tuple& o) const { return (x * o.x) + (y * o.y) + (z * o.z) + (w * o.w); }
    Tuple cross(const Tuple& o) const {
        return vector((y * o.z) - (z * o.y), (z * o.x) - (x * o.z), (x * o.y) - (y * o.x));
    }
    bool operator==(const Tuple& o) const {
        return near_eq(x, o.x) && near_eq(y, o.y) && near_eq(z, o.z) && near_eq(w, o.w);
    }
};
inline Tuple operator+(const Tuple& a, const Tuple& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline Tuple operator-(const Tuple& a, const Tuple& b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline Tuple operator-(const Tuple& a) { return {-a.x, -a.y, -a.z, -a.w}; }
inline Tuple operator*(const Tuple& a, double s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
inline Tuple operator/(const Tuple& a, double s) { return {a.x / s, a.y / s, a.z / s, a.w / s}; }
const Tuple ORIGIN = Tuple::point(0., 0., 0.);  // src/lib.rs:43

// ---------------------------------------------------------------- color.rs
struct Color {
    double red = 0, green = 0, blue = 0;
    bool operator==(const Color& o) const {
        return near_eq(red, o.red) && near_eq(green, o.green) && near_eq(blue, o.blue);
    }
};
inline Color operator+(const Color& a, const Color& b) { return {a.red + b.red, a.green + b.green, a.blue + b.blue}; }
inline Color operator-(const Color& a, const Color& b) { return {a.red - b.red, a.green - b.green, a.blue - b.blue}; }
inline Color operator*(const Color& a, double s) { return {a.red * s, a.green * s, a.blue * s}; }
inline Color operator*(const Color& a, const Color& b) { return {a.red * b.red, a.green * b.green, a.blue * b.blue}; }
const Color BLACK{0., 0., 0.};  // src/lib.rs:40
const Color WHITE{1., 1., 1.};  // src/lib.rs:44

// ---------------------------------------------------------------- matrix.rs
// rows x columns <= 4x4.  Carries the reference's matrix id: Matrix::inverse is cached per id in a process-global
// table (src/matrix.rs:124-154), so an inverse computed once is returned for that id from then on.
class Matrix {
public:
    Matrix(int rows, int columns, const std::vector<double>& values);
    static Matrix identity(int size);
    int get_rows() const { return rows_; }
    int get_columns() const { return columns_; }
    int get_id() const { return id_; }
    double& at(int r, int c) { return v_[r * columns_ + c]; }
    double at(int r, int c) const { return v_[r * columns_ + c]; }
    Matrix transpose() const;
    double determinant() const;
    Matrix submatrix(int row_to_remove, int column_to_remove) const;
    double minor(int row, int column) const;
    double cofactor(int row, int column) const;
    std::optional<Matrix> inverse() const;  // None when |det| < EPSILON (src/matrix.rs:133-136)
    bool operator==(const Matrix& o) const;
    const double* data() const { return v_; }

private:
    Matrix(int id, int rows, int columns) : id_(id), rows_(rows), columns_(columns) {
        for (double& d : v_) d = 0.;
    }
    int id_;
    int rows_, columns_;
    double v_[16];
    friend Matrix operator*(const Matrix&, const Matrix&);
};
Matrix operator*(const Matrix& a, const Matrix& b);  // src/matrix.rs:171-188
Tuple operator*(const Matrix& m, const Tuple& t);    // src/matrix.rs:190-200

// ---------------------------------------------------------------- transformation.rs
enum class Axis { X, Y, Z };
Matrix translate(double x, double y, double z);
Matrix scale(double x, double y, double z);
Matrix rotate(double radians, Axis axis);
Matrix shear(double x_y, double x_z, double y_x, double y_z, double z_x, double z_y);
Matrix view_transform(const Tuple& from, const Tuple& to, const Tuple& up);

// ---------------------------------------------------------------- pattern.rs
enum class PatternKind { Striped = 0, Gradient = 1, Ring = 2, Checkered = 3, RingGradient = 4, Test = 5 };
struct Pattern {
    PatternKind kind;
    Color a, b;
    Matrix transform = Matrix::identity(4);
    static Pattern Striped(Color a, Color b) { return {PatternKind::Striped, a, b}; }
    static Pattern Gradient(Color a, Color b) { return {PatternKind::Gradient, a, b}; }
    static Pattern Ring(Color a, Color b) { return {PatternKind::Ring, a, b}; }
    static Pattern Checkered(Color a, Color b) { return {PatternKind::Checkered, a, b}; }
    static Pattern RingGradient(Color a, Color b) { return {PatternKind::RingGradient, a, b}; }
    static Pattern Test() { return {PatternKind::Test, BLACK, BLACK}; }
    const Matrix& get_transform() const { return transform; }
    void set_transform(const Matrix& m) { transform = m; }
    bool operator==(const Pattern& o) const {  // #[derive(PartialEq)]
        if (kind != o.kind) return false;
        if (kind != PatternKind::Test && !(a == o.a && b == o.b)) return false;
        return transform == o.transform;
    }
};

// ---------------------------------------------------------------- material.rs
struct Material {
    Color color = WHITE;
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
    std::optional<Pattern> pattern;
    double reflective = 0., transparency = 0., refractive_index = 1.;
    Material& with_color(Color c) { color = c; return *this; }
    Material& with_ambient(double v) { ambient = v; return *this; }
    Material& with_diffuse(double v) { diffuse = v; return *this; }
    Material& with_specular(double v) { specular = v; return *this; }
    Material& with_shininess(double v) { shininess = v; return *this; }
    Material& with_pattern(const Pattern& p) { pattern = p; return *this; }
    Material& with_reflective(double v) { reflective = v; return *this; }
    Material& with_transparency(double v) { transparency = v; return *this; }
    Material& with_refractive_index(double v) { refractive_index = v; return *this; }
    // src/material.rs:39-45 — reflective / transparency / refractive_index are NOT compared.
    bool operator==(const Material& o) const {
        bool pat = (pattern.has_value() == o.pattern.has_value()) && (!pattern.has_value() || *pattern == *o.pattern);
        return color == o.color && near_eq(ambient, o.ambient) && near_eq(diffuse, o.diffuse) &&
               near_eq(specular, o.specular) && near_eq(shininess, o.shininess) && pat;
    }
};

// ---------------------------------------------------------------- light.rs
struct Light {
    Tuple position;
    Color intensity;
    static Light point_light(const Tuple& position, const Color& intensity) { return {position, intensity}; }
};

// ---------------------------------------------------------------- bound.rs
struct Bound {
    Tuple minimum, maximum;
    static Bound bounding_box_empty() { return {Tuple::point(INF, INF, INF), Tuple::point(-INF, -INF, -INF)}; }
    static Bound bounding_box_init(const Tuple& mn, const Tuple& mx) { return {mn, mx}; }
    void add_point(const Tuple& p);
    void add_box(const Bound& o) { add_point(o.minimum); add_point(o.maximum); }
    bool box_contains_point(const Tuple& p) const;
    bool box_contains_box(const Bound& o) const { return box_contains_point(o.minimum) && box_contains_point(o.maximum); }
    Bound transform(const Matrix& m) const;
    std::pair<Bound, Bound> split_bounds() const;
};

// ---------------------------------------------------------------- shape.rs + the shape modules
enum class ShapeKind { Sphere = 0, Plane, Cube, Cylinder, Cone, Triangle, SmoothTriangle, Group, CSG };

// Compact form of the Group tree parse_obj_file + obj_to_group build: one record per
// triangle instead of one Shape (a Shape carries a Matrix, a Material and an optional Pattern, and Group::add_child
// clones it).  Object ids are drawn in the order the reference draws them, so the flattened tables are the same, entry
// for entry, as those of the Shape tree (tests/test_host_api.py compares them).  Immutable once built, shared by clones.
struct PackedTri {
    int id;
    bool smooth;
    Tuple p1, p2, p3, e1, e2, normal;  // as Triangle::new computes them (src/triangle.rs:38-56)
    Tuple n1, n2, n3;                  // SmoothTriangle only
};
struct PackedGroup {  // one per `g` record, plus the parser's default group; identity transform, default material
    int id;
    std::vector<PackedTri> tris;
};
struct PackedMesh {
    std::vector<PackedGroup> groups;
    int id_first = 0, id_last = -1;  // the contiguous range of object ids drawn for groups and triangles
};

class Shape {
public:
    // constructors: Sphere::new(), Plane::new(), ... each draws a fresh object id
    static Shape Sphere();
    static Shape GlassSphere();  // Sphere::glass_sphere, src/sphere.rs:69-78
    static Shape Plane();
    static Shape Cube();
    static Shape Cylinder();
    static Shape Cone();
    static Shape Triangle(const Tuple& p1, const Tuple& p2, const Tuple& p3);
    static Shape SmoothTriangle(const Tuple& p1, const Tuple& p2, const Tuple& p3, const Tuple& n1, const Tuple& n2,
                                const Tuple& n3);
    static Shape Group();
    // CSG::new(operation, &mut left, &mut right): sets both parents, stores clones (src/csg.rs:33-53)
    static Shape CSG(const std::string& operation, Shape& left, Shape& right);

    Shape(const Shape& o);
    Shape& operator=(const Shape& o);
    Shape(Shape&&) = default;
    Shape& operator=(Shape&&) = default;

    ShapeKind kind() const { return kind_; }
    int get_id() const { return id_; }
    const Matrix& get_transform() const { return transform; }
    void set_transform(const Matrix& m);
    Material get_material() const;  // ROOT ancestor's material through the registry (src/shape.rs:163-184)
    void set_material(const Material& m);
    bool get_casts_shadow() const { return casts_shadow; }
    void set_casts_shadow(bool v);
    double get_minimum() const;
    void set_minimum(double v);
    double get_maximum() const;
    void set_maximum(double v);
    bool get_closed() const;
    void set_closed(bool v);
    std::optional<int> get_parent() const { return parent; }
    void set_parent(int p);
    const std::vector<Shape>& get_shapes() const;
    // A Group made by obj_file_to_group keeps its children packed until something asks for them as Shapes
    // (get_shapes, add_child, divide, includes, bounds_of ...): then they are materialised, ids and all.
    const PackedMesh* packed() const { return packed_.get(); }
    void unpack() const;
    static Shape FromPackedMesh(std::shared_ptr<const PackedMesh> mesh);  // draws the main group's id
    static Shape FromPackedTri(const PackedTri& t, int parent_id);
    void add_child(Shape& child);  // Group only
    bool includes(const Shape& other) const;
    Bound bounds_of() const;
    Bound parent_space_bounds_of() const { return bounds_of().transform(transform); }
    std::pair<std::vector<Shape>, std::vector<Shape>> partition_children();
    void make_subgroup(std::vector<Shape> shapes);
    void divide(size_t threshold);
    bool operator==(const Shape& o) const;  // the per-struct PartialEq impls

    // pub fields of the reference structs
    Matrix transform = Matrix::identity(4);
    Material material;
    bool casts_shadow = true;
    std::optional<int> parent;
    double minimum = -INF, maximum = INF;  // Cylinder / Cone
    bool closed = false;
    Tuple point1, point2, point3, edge_vector1, edge_vector2, normal_vector;  // Triangle / SmoothTriangle
    Tuple normal_vector1, normal_vector2, normal_vector3;                     // SmoothTriangle
    mutable std::vector<Shape> shapes;                                        // Group (see unpack)
    std::string operation;                                                    // CSG
    std::unique_ptr<Shape> left, right;                                       // CSG

private:
    Shape(ShapeKind k) : kind_(k), id_(generate_object_id()) {}
    Shape(ShapeKind k, int id) : kind_(k), id_(id) {}  // a Shape that already has its id (unpack)
    void update_reference() const;  // Group::update_group_reference / CSG::update_csg_reference
    ShapeKind kind_;
    int id_;
    mutable std::shared_ptr<const PackedMesh> packed_;
};

// The PARENT_REFERENCES registry (src/lib.rs:47-49), reduced to the fields ever read through it.
struct RegistryEntry {
    int id;
    Matrix transform;
    Material material;
    std::optional<int> parent;
};
const RegistryEntry* registry_find(int id);

// ---------------------------------------------------------------- obj_file.rs
struct Parser {
    std::vector<Tuple> vertices;  // 1-based: [0] is a dummy (src/obj_file.rs:33)
    std::vector<Shape> groups;
    std::vector<Tuple> normals;
    int lines_ignored = 0;
};
Parser parse_obj_file(const std::string& data);
Shape obj_to_group(Parser&& parser);
// parse_obj_file + obj_to_group in one go with the children kept packed (same parser code, same ids, same registry
// entries, same flattened tables; counts as Parser reports them: lines ignored, vertices, normals, groups, triangles)
Shape obj_file_to_group(const std::string& data, int32_t counts[5] = nullptr);
Shape hexagon();  // src/lib.rs:91-126

// ---------------------------------------------------------------- canvas.rs
class Canvas {
public:
    Canvas(size_t width, size_t height) : width_(width), height_(height), pixels_(width * height, BLACK) {}
    size_t get_width() const { return width_; }
    size_t get_height() const { return height_; }
    void write_pixel(uint32_t x, uint32_t y, const Color& c) { pixels_.at(size_t(y) * width_ + x) = c; }
    Color pixel_at(uint32_t x, uint32_t y) const { return pixels_.at(size_t(y) * width_ + x); }
    std::string canvas_to_ppm() const;
    static int scale_color(double color_value, int max_color_value);
    std::vector<Color>& pixels() { return pixels_; }

private:
    size_t width_, height_;
    std::vector<Color> pixels_;
};

// ---------------------------------------------------------------- flatten.cpp
// Owning storage behind an rt_scene_desc.
struct FlatScene {
    std::vector<int32_t> node_kind, node_object_id, node_parent, node_subtree_end, node_xform, node_ancestor,
        node_material, node_eq_class, node_param;
    std::vector<uint8_t> node_casts_shadow;
    std::vector<double> xform_inverse, xform_inverse_transpose;
    std::vector<int32_t> ancestor_xform, ancestor_parent;
    std::vector<double> group_box;
    std::vector<int32_t> csg_op;
    std::vector<double> conic_minimum, conic_maximum;
    std::vector<uint8_t> conic_closed;
    std::vector<double> tri_p1, tri_e1, tri_e2, tri_normal;
    std::vector<int32_t> tri_smooth;
    std::vector<double> smooth_n1, smooth_n2, smooth_n3;
    std::vector<double> material_params;
    std::vector<int32_t> material_pattern;
    std::vector<int32_t> pattern_kind;
    std::vector<double> pattern_a, pattern_b, pattern_inverse;
    std::vector<double> light_position, light_intensity;
    rt_scene_desc desc() const;
};

// ---------------------------------------------------------------- world.rs
struct Ray {
    Tuple origin, direction;
};
struct Intersection {  // what World::intersect_world reports back from the device
    double t;
    int object_id;
};

class World {
public:
    std::vector<Shape> objects;
    std::vector<Light> lights;
    static World default_world();  // impl Default for World, src/world.rs:20-39
    World() = default;
    World(const World& o) : objects(o.objects), lights(o.lights) {}  // the device copy is not shared
    World& operator=(const World& o) {
        if (this != &o) { invalidate(); objects = o.objects; lights = o.lights; }
        return *this;
    }
    ~World();

    FlatScene flatten() const;
    // Device-backed counterparts of the reference methods (src/world.rs:46,81,92).
    std::vector<Intersection> intersect_world(const Ray& ray) const;
    Color color_at(const Ray& ray, int remaining) const;
    std::vector<bool> is_shadowed(const Tuple& point) const;
    // Handle of the device copy of this world on `device` (built on first use, one per device, all kept until
    // invalidate() — call that after mutating the world — or destruction: a handle given out stays valid when another
    // device is asked for).
    rt_scene* device_scene(int device = 0) const;
    void invalidate() const;

private:
    mutable std::map<int, rt_scene*> scenes_;
};

// ---------------------------------------------------------------- camera.rs
class Camera {
public:
    size_t hsize, vsize;
    double field_of_view;
    Matrix transform = Matrix::identity(4);
    double half_width, half_height, pixel_size;
    Camera(size_t hsize, size_t vsize, double field_of_view);
    rt_camera_desc desc() const;
    // Camera::render (src/camera.rs:62-74) on the GPU: flatten -> rt_render -> Canvas.
    Canvas render(const World& world) const { return render(world, DEFAULT_RECURSION); }
    Canvas render(const World& world, int depth, std::vector<int32_t>* hit_ids = nullptr,
                  std::vector<uint8_t>* flags = nullptr, rt_stats* stats = nullptr) const;
};

}  // namespace ray_tracer
