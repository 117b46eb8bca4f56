// ray_tracer.cpp — host-side mirror of the reference crate's scene-construction API (see ray_tracer.hpp).
// Each block cites the reference file it follows.  No hot-path arithmetic lives here.
#include "ray_tracer.hpp"

#include <cstdlib>
#include <cstring>
#include <charconv>
#include <sstream>
#include <string_view>
#include <unordered_map>

namespace ray_tracer {

// ---------------------------------------------------------------- lib.rs globals
static int g_object_id = 1;  // ID_COUNT, src/lib.rs:46
static int g_matrix_id = 1;  // MATRIX_ID, src/lib.rs:51
int generate_object_id() { return g_object_id++; }
int generate_matrix_id() { return g_matrix_id++; }

static std::unordered_map<int, Matrix>& cached_inverses() {  // CACHED_INVERSES, src/lib.rs:52-54
    static std::unordered_map<int, Matrix> m;
    return m;
}
static std::unordered_map<int, RegistryEntry>& parent_references() {  // PARENT_REFERENCES, src/lib.rs:47-49
    static std::unordered_map<int, RegistryEntry> m;
    return m;
}
const RegistryEntry* registry_find(int id) {
    auto& r = parent_references();
    auto it = r.find(id);
    return it == r.end() ? nullptr : &it->second;
}

// ---------------------------------------------------------------- matrix.rs
Matrix::Matrix(int rows, int columns, const std::vector<double>& values)
    : id_(generate_matrix_id()), rows_(rows), columns_(columns) {
    if (rows < 1 || rows > 4 || columns < 1 || columns > 4 || values.size() != size_t(rows * columns))
        throw Panic("Matrix::new: bad dimensions");
    for (double& d : v_) d = 0.;
    for (int i = 0; i < rows * columns; ++i) v_[i] = values[i];
}

Matrix Matrix::identity(int size) {  // src/matrix.rs:52-60
    Matrix m(size, size, std::vector<double>(size_t(size * size), 0.));
    for (int r = 0; r < size; ++r) m.at(r, r) = 1.;
    return m;
}

Matrix Matrix::transpose() const {  // src/matrix.rs:62-73 (indexes [column][row] as written there)
    Matrix t(rows_, columns_, std::vector<double>(size_t(rows_ * columns_), 0.));
    for (int r = 0; r < rows_; ++r)
        for (int c = 0; c < columns_; ++c) t.at(r, c) = at(c, r);
    return t;
}

double Matrix::determinant() const {  // src/matrix.rs:75-86
    if (rows_ == 2 && columns_ == 2) return (at(0, 0) * at(1, 1)) - (at(0, 1) * at(1, 0));
    double determinant = 0.;
    for (int c = 0; c < columns_; ++c) determinant += at(0, c) * cofactor(0, c);
    return determinant;
}

Matrix Matrix::submatrix(int row_to_remove, int column_to_remove) const {  // src/matrix.rs:88-110
    Matrix s(rows_ - 1, columns_ - 1, std::vector<double>(size_t((rows_ - 1) * (columns_ - 1)), 0.));
    int sr = 0;
    for (int r = 0; r < rows_; ++r) {
        if (r == row_to_remove) continue;
        int sc = 0;
        for (int c = 0; c < columns_; ++c) {
            if (c == column_to_remove) continue;
            s.at(sr, sc) = at(r, c);
            ++sc;
        }
        ++sr;
    }
    return s;
}

double Matrix::minor(int row, int column) const { return submatrix(row, column).determinant(); }

double Matrix::cofactor(int row, int column) const {  // src/matrix.rs:118-122
    double m = minor(row, column);
    return ((row + column) % 2 == 1) ? m * -1. : m;
}

std::optional<Matrix> Matrix::inverse() const {  // src/matrix.rs:124-154
    auto& cache = cached_inverses();
    auto it = cache.find(id_);
    if (it != cache.end()) return it->second;
    double determinant = this->determinant();
    if (near_eq(determinant, 0.)) return std::nullopt;
    Matrix inv(id_, rows_, columns_);  // Matrix::new_inverse keeps the id of `self`
    for (int r = 0; r < rows_; ++r)
        for (int c = 0; c < columns_; ++c) inv.at(c, r) = cofactor(r, c) / determinant;
    cache.emplace(id_, inv);
    return inv;
}

bool Matrix::operator==(const Matrix& o) const {  // src/matrix.rs:17-33
    if (rows_ != o.rows_ || columns_ != o.columns_) return false;
    for (int r = 0; r < rows_; ++r)
        for (int c = 0; c < columns_; ++c)
            if (!near_eq(at(r, c), o.at(r, c))) return false;
    return true;
}

Matrix operator*(const Matrix& a, const Matrix& b) {  // src/matrix.rs:171-188
    Matrix p(a.rows_, b.columns_, std::vector<double>(size_t(a.rows_ * b.columns_), 0.));
    for (int r = 0; r < a.rows_; ++r)
        for (int c = 0; c < b.columns_; ++c)
            for (int e = 0; e < a.columns_; ++e) p.at(r, c) += a.at(r, e) * b.at(e, c);
    return p;
}

Tuple operator*(const Matrix& m, const Tuple& t) {  // src/matrix.rs:190-200
    Matrix col(4, 1, {t.x, t.y, t.z, t.w});
    Matrix p = m * col;
    return {p.at(0, 0), p.at(1, 0), p.at(2, 0), p.at(3, 0)};
}

// ---------------------------------------------------------------- transformation.rs
Matrix translate(double x, double y, double z) {
    Matrix m = Matrix::identity(4);
    m.at(0, 3) = x; m.at(1, 3) = y; m.at(2, 3) = z;
    return m;
}
Matrix scale(double x, double y, double z) {
    Matrix m = Matrix::identity(4);
    m.at(0, 0) = x; m.at(1, 1) = y; m.at(2, 2) = z;
    return m;
}
Matrix rotate(double radians, Axis axis) {  // src/transformation.rs:28-53
    Matrix m = Matrix::identity(4);
    switch (axis) {
        case Axis::X:
            m.at(1, 1) = std::cos(radians); m.at(1, 2) = -std::sin(radians);
            m.at(2, 1) = std::sin(radians); m.at(2, 2) = std::cos(radians);
            break;
        case Axis::Y:
            m.at(0, 0) = std::cos(radians); m.at(0, 2) = std::sin(radians);
            m.at(2, 0) = -std::sin(radians); m.at(2, 2) = std::cos(radians);
            break;
        case Axis::Z:
            m.at(0, 0) = std::cos(radians); m.at(0, 1) = -std::sin(radians);
            m.at(1, 0) = std::sin(radians); m.at(1, 1) = std::cos(radians);
            break;
    }
    return m;
}
Matrix shear(double x_y, double x_z, double y_x, double y_z, double z_x, double z_y) {
    Matrix m = Matrix::identity(4);
    m.at(0, 1) = x_y; m.at(0, 2) = x_z; m.at(1, 0) = y_x; m.at(1, 2) = y_z; m.at(2, 0) = z_x; m.at(2, 1) = z_y;
    return m;
}
Matrix view_transform(const Tuple& from, const Tuple& to, const Tuple& up) {  // src/transformation.rs:67-80
    Tuple forward = (to - from).normalize();
    Tuple up_normal = up.normalize();
    Tuple left = forward.cross(up_normal);
    Tuple true_up = left.cross(forward);
    Matrix orientation(4, 4, {left.x, left.y, left.z, 0., true_up.x, true_up.y, true_up.z, 0.,
                              -forward.x, -forward.y, -forward.z, 0., 0., 0., 0., 1.});
    return orientation * translate(-from.x, -from.y, -from.z);
}

// ---------------------------------------------------------------- bound.rs
void Bound::add_point(const Tuple& p) {  // src/bound.rs:35-42
    if (p.x < minimum.x) minimum.x = p.x;
    if (p.y < minimum.y) minimum.y = p.y;
    if (p.z < minimum.z) minimum.z = p.z;
    if (p.x > maximum.x) maximum.x = p.x;
    if (p.y > maximum.y) maximum.y = p.y;
    if (p.z > maximum.z) maximum.z = p.z;
}
bool Bound::box_contains_point(const Tuple& p) const {  // src/bound.rs:49-58
    bool cx = near_eq(minimum.x, p.x) || near_eq(maximum.x, p.x) || (minimum.x < p.x && p.x < maximum.x);
    bool cy = near_eq(minimum.y, p.y) || near_eq(maximum.y, p.y) || (minimum.y < p.y && p.y < maximum.y);
    bool cz = near_eq(minimum.z, p.z) || near_eq(maximum.z, p.z) || (minimum.z < p.z && p.z < maximum.z);
    return cx && cy && cz;
}
Bound Bound::transform(const Matrix& m) const {  // src/bound.rs:64-83
    const Tuple pts[8] = {minimum,
                          Tuple::point(minimum.x, minimum.y, maximum.z),
                          Tuple::point(minimum.x, maximum.y, minimum.z),
                          Tuple::point(minimum.x, maximum.y, maximum.z),
                          Tuple::point(maximum.x, minimum.y, minimum.z),
                          Tuple::point(maximum.x, minimum.y, maximum.z),
                          Tuple::point(maximum.x, maximum.y, minimum.z),
                          maximum};
    Bound nb = bounding_box_empty();
    for (const Tuple& p : pts) nb.add_point(m * p);
    return nb;
}
std::pair<Bound, Bound> Bound::split_bounds() const {  // src/bound.rs:118-146
    double dx = std::fabs(maximum.x - minimum.x), dy = std::fabs(maximum.y - minimum.y),
           dz = std::fabs(maximum.z - minimum.z);
    double greatest = std::fmax(std::fmax(std::fmax(std::nan(""), dx), dy), dz);
    double x0 = minimum.x, y0 = minimum.y, z0 = minimum.z, x1 = maximum.x, y1 = maximum.y, z1 = maximum.z;
    if (near_eq(greatest, dx)) { x0 = x0 + dx / 2.; x1 = x0; }
    else if (near_eq(greatest, dy)) { y0 = y0 + dy / 2.; y1 = y0; }
    else { z0 = z0 + dz / 2.; z1 = z0; }
    return {bounding_box_init(minimum, Tuple::point(x1, y1, z1)), bounding_box_init(Tuple::point(x0, y0, z0), maximum)};
}

// ---------------------------------------------------------------- shape.rs and the shape modules
Shape Shape::Sphere() { return Shape(ShapeKind::Sphere); }
Shape Shape::GlassSphere() {
    Shape s(ShapeKind::Sphere);
    s.material.transparency = 1.;
    s.material.refractive_index = 1.5;
    return s;
}
Shape Shape::Plane() { return Shape(ShapeKind::Plane); }
Shape Shape::Cube() { return Shape(ShapeKind::Cube); }
Shape Shape::Cylinder() { return Shape(ShapeKind::Cylinder); }
Shape Shape::Cone() { return Shape(ShapeKind::Cone); }
Shape Shape::Triangle(const Tuple& p1, const Tuple& p2, const Tuple& p3) {  // src/triangle.rs:38-56
    Shape s(ShapeKind::Triangle);
    s.point1 = p1; s.point2 = p2; s.point3 = p3;
    s.edge_vector1 = p2 - p1;
    s.edge_vector2 = p3 - p1;
    s.normal_vector = s.edge_vector2.cross(s.edge_vector1).normalize();
    return s;
}
Shape Shape::SmoothTriangle(const Tuple& p1, const Tuple& p2, const Tuple& p3, const Tuple& n1, const Tuple& n2,
                            const Tuple& n3) {  // src/smooth_triangle.rs:42-65
    Shape s(ShapeKind::SmoothTriangle);
    s.point1 = p1; s.point2 = p2; s.point3 = p3;
    s.normal_vector1 = n1; s.normal_vector2 = n2; s.normal_vector3 = n3;
    s.edge_vector1 = p2 - p1;
    s.edge_vector2 = p3 - p1;
    s.normal_vector = s.edge_vector2.cross(s.edge_vector1).normalize();
    return s;
}
Shape Shape::Group() { return Shape(ShapeKind::Group); }
Shape Shape::FromPackedTri(const PackedTri& t, int parent_id) {
    Shape s(t.smooth ? ShapeKind::SmoothTriangle : ShapeKind::Triangle, t.id);
    s.point1 = t.p1; s.point2 = t.p2; s.point3 = t.p3;
    s.edge_vector1 = t.e1; s.edge_vector2 = t.e2; s.normal_vector = t.normal;
    if (t.smooth) { s.normal_vector1 = t.n1; s.normal_vector2 = t.n2; s.normal_vector3 = t.n3; }
    s.parent = parent_id;
    return s;
}
Shape Shape::FromPackedMesh(std::shared_ptr<const PackedMesh> mesh) {
    Shape s(ShapeKind::Group);
    // what the add_child calls of parse_obj_file / obj_to_group leave in PARENT_REFERENCES: every sub-group with the main
    // group as its parent (src/group.rs:45-50, src/shape.rs:357-378), then the main group itself
    for (const PackedGroup& g : mesh->groups)
        parent_references().insert_or_assign(g.id, RegistryEntry{g.id, Matrix::identity(4), Material(), s.id_});
    s.packed_ = std::move(mesh);
    s.update_reference();
    return s;
}
// The Shape tree obj_to_group would have built: sub-groups and triangles with the ids, parents and fields the packed
// records carry (the registry entries of the sub-groups exist since the parse).
void Shape::unpack() const {
    if (!packed_) return;
    std::shared_ptr<const PackedMesh> m = std::move(packed_);
    packed_.reset();
    for (const PackedGroup& g : m->groups) {
        Shape sg(ShapeKind::Group, g.id);
        sg.parent = id_;
        sg.shapes.reserve(g.tris.size());
        for (const PackedTri& t : g.tris) sg.shapes.push_back(FromPackedTri(t, g.id));
        shapes.push_back(std::move(sg));
    }
}
Shape Shape::CSG(const std::string& operation, Shape& left, Shape& right) {  // src/csg.rs:33-53
    Shape s(ShapeKind::CSG);
    left.set_parent(s.id_);
    right.set_parent(s.id_);
    s.operation = operation;
    s.left = std::make_unique<Shape>(left);
    s.right = std::make_unique<Shape>(right);
    s.update_reference();
    return s;
}

Shape::Shape(const Shape& o)
    : transform(o.transform), material(o.material), casts_shadow(o.casts_shadow), parent(o.parent),
      minimum(o.minimum), maximum(o.maximum), closed(o.closed), point1(o.point1), point2(o.point2),
      point3(o.point3), edge_vector1(o.edge_vector1), edge_vector2(o.edge_vector2), normal_vector(o.normal_vector),
      normal_vector1(o.normal_vector1), normal_vector2(o.normal_vector2), normal_vector3(o.normal_vector3),
      shapes(o.shapes), operation(o.operation), left(o.left ? std::make_unique<Shape>(*o.left) : nullptr),
      right(o.right ? std::make_unique<Shape>(*o.right) : nullptr), kind_(o.kind_), id_(o.id_), packed_(o.packed_) {}
Shape& Shape::operator=(const Shape& o) {
    if (this != &o) {
        Shape tmp(o);
        *this = std::move(tmp);
    }
    return *this;
}

void Shape::update_reference() const {  // src/group.rs:71-84, src/csg.rs:80-93
    parent_references().insert_or_assign(id_, RegistryEntry{id_, transform, material, parent});
}
static bool is_container(ShapeKind k) { return k == ShapeKind::Group || k == ShapeKind::CSG; }

void Shape::set_transform(const Matrix& m) {  // src/shape.rs:140-161
    transform = m;
    if (is_container(kind_)) update_reference();
}
void Shape::set_material(const Material& m) {  // src/shape.rs:186-207
    material = m;
    if (is_container(kind_)) update_reference();
}
void Shape::set_casts_shadow(bool v) {  // src/shape.rs:224-245
    casts_shadow = v;
    if (is_container(kind_)) update_reference();
}
void Shape::set_parent(int p) {  // src/shape.rs:357-378
    parent = p;
    if (is_container(kind_)) update_reference();
}
Material Shape::get_material() const {  // src/shape.rs:163-184
    if (!parent) return material;
    const RegistryEntry* e = registry_find(*parent);
    while (true) {
        if (!e) throw Panic("called `Option::unwrap()` on a `None` value (parent id not in PARENT_REFERENCES)");
        if (!e->parent) return e->material;
        e = registry_find(*e->parent);
    }
}
static bool is_conic(ShapeKind k) { return k == ShapeKind::Cylinder || k == ShapeKind::Cone; }
double Shape::get_minimum() const {
    if (!is_conic(kind_)) throw Panic("Minimum property is not available for this shape.");
    return minimum;
}
void Shape::set_minimum(double v) {
    if (!is_conic(kind_)) throw Panic("Minimum property is not available for this shape.");
    minimum = v;
}
double Shape::get_maximum() const {
    if (!is_conic(kind_)) throw Panic("Maximum property is not available for this shape.");
    return maximum;
}
void Shape::set_maximum(double v) {
    if (!is_conic(kind_)) throw Panic("Maximum property is not available for this shape.");
    maximum = v;
}
bool Shape::get_closed() const {
    if (!is_conic(kind_)) throw Panic("Closed property is not available for this shape.");
    return closed;
}
void Shape::set_closed(bool v) {
    if (!is_conic(kind_)) throw Panic("Closed property is not available for this shape.");
    closed = v;
}
const std::vector<Shape>& Shape::get_shapes() const {
    if (kind_ != ShapeKind::Group) throw Panic("Only groups can contain children.");
    unpack();
    return shapes;
}
void Shape::add_child(Shape& child) {  // src/group.rs:45-50
    if (kind_ != ShapeKind::Group) throw Panic("Only groups can contain children.");
    unpack();
    child.set_parent(id_);
    shapes.push_back(child);
    update_reference();
}
bool Shape::includes(const Shape& other) const {  // src/shape.rs:422-435
    switch (kind_) {
        case ShapeKind::Group:
            unpack();
            for (const Shape& s : shapes)
                if (s.includes(other)) return true;
            return false;
        case ShapeKind::CSG:
            return left->includes(other) || right->includes(other);
        default:
            return id_ == other.id_;
    }
}
Bound Shape::bounds_of() const {
    switch (kind_) {
        case ShapeKind::Sphere:  // src/sphere.rs:80-83
        case ShapeKind::Cube:    // src/cube.rs:94-97
            return Bound::bounding_box_init(Tuple::point(-1., -1., -1.), Tuple::point(1., 1., 1.));
        case ShapeKind::Plane:  // src/plane.rs:58-61
            return Bound::bounding_box_init(Tuple::point(-INF, 0., -INF), Tuple::point(INF, 0., INF));
        case ShapeKind::Cylinder:  // src/cylinder.rs:129-132
            return Bound::bounding_box_init(Tuple::point(-1., minimum, -1.), Tuple::point(1., maximum, 1.));
        case ShapeKind::Cone: {  // src/cone.rs:137-144
            double limit = std::fmax(std::fabs(minimum), std::fabs(maximum));
            return Bound::bounding_box_init(Tuple::point(-limit, minimum, -limit), Tuple::point(limit, maximum, limit));
        }
        case ShapeKind::Group: {  // src/group.rs:86-95
            unpack();
            Bound b = Bound::bounding_box_empty();
            for (const Shape& s : shapes) b.add_box(s.parent_space_bounds_of());
            return b;
        }
        default:  // Triangle, SmoothTriangle, CSG: src/triangle.rs:93-95, src/smooth_triangle.rs:106-108, src/csg.rs:128-137
            return Bound::bounding_box_empty();
    }
}
std::pair<std::vector<Shape>, std::vector<Shape>> Shape::partition_children() {  // src/group.rs:97-123
    if (kind_ != ShapeKind::Group) throw Panic("Only groups can partition children.");
    unpack();
    auto [left_box, right_box] = bounds_of().split_bounds();
    std::vector<Shape> l, r, keep;
    for (Shape& s : shapes) {
        if (left_box.box_contains_box(s.parent_space_bounds_of())) l.push_back(s);
        else if (right_box.box_contains_box(s.parent_space_bounds_of())) r.push_back(s);
        else keep.push_back(s);
    }
    shapes = std::move(keep);
    update_reference();
    return {std::move(l), std::move(r)};
}
void Shape::make_subgroup(std::vector<Shape> list) {  // src/group.rs:125-132
    if (kind_ != ShapeKind::Group) throw Panic("Only groups can contain subgroups.");
    Shape sub = Shape::Group();
    for (Shape& s : list) sub.add_child(s);
    add_child(sub);
}
void Shape::divide(size_t threshold) {  // src/group.rs:134-149, src/shape.rs:470-475
    if (kind_ != ShapeKind::Group) return;
    unpack();
    if (threshold <= shapes.size()) {
        auto [l, r] = partition_children();
        if (!l.empty()) make_subgroup(std::move(l));
        if (!r.empty()) make_subgroup(std::move(r));
    }
    for (Shape& c : shapes) c.divide(threshold);
}
bool Shape::operator==(const Shape& o) const {
    if (kind_ != o.kind_) return false;  // #[derive(PartialEq)] on enum Shape, src/shape.rs:25
    bool common = id_ == o.id_ && transform == o.transform && material == o.material &&
                  casts_shadow == o.casts_shadow && parent == o.parent;
    if (!common) return false;
    switch (kind_) {
        case ShapeKind::Cylinder:  // src/cylinder.rs:26-33
        case ShapeKind::Cone:      // src/cone.rs:26-33
            return near_eq(minimum, o.minimum) && near_eq(maximum, o.maximum) && closed == o.closed;
        case ShapeKind::Triangle:  // src/triangle.rs:26-35
            return point1 == o.point1 && point2 == o.point2 && point3 == o.point3 &&
                   edge_vector1 == o.edge_vector1 && edge_vector2 == o.edge_vector2 && normal_vector == o.normal_vector;
        case ShapeKind::SmoothTriangle:  // src/smooth_triangle.rs:29-39
            return point1 == o.point1 && point2 == o.point2 && point3 == o.point3 &&
                   normal_vector1 == o.normal_vector1 && normal_vector2 == o.normal_vector2 &&
                   normal_vector3 == o.normal_vector3 && edge_vector1 == o.edge_vector1 &&
                   edge_vector2 == o.edge_vector2 && normal_vector == o.normal_vector;
        case ShapeKind::CSG:  // src/csg.rs:24-30
            return operation == o.operation;
        default:
            return true;
    }
}

// ---------------------------------------------------------------- lib.rs:91-126
static Shape hexagon_corner() {
    Shape corner = Shape::Sphere();
    corner.set_transform(translate(0., 0., -1.) * scale(0.25, 0.25, 0.25));
    return corner;
}
static Shape hexagon_edge() {
    Shape edge = Shape::Cylinder();
    edge.set_minimum(0.);
    edge.set_maximum(1.);
    edge.set_transform(translate(0., 0., -1.) * rotate(-(M_PI / 6.), Axis::Y) * rotate(-(M_PI / 2.), Axis::Z) *
                       scale(0.25, 1., 0.25));
    return edge;
}
static Shape hexagon_side() {
    Shape side = Shape::Group();
    Shape c = hexagon_corner(), e = hexagon_edge();
    side.add_child(c);
    side.add_child(e);
    return side;
}
Shape hexagon() {
    Shape hex = Shape::Group();
    for (int n = 0; n <= 5; ++n) {
        Shape side = hexagon_side();
        side.set_transform(rotate(double(n) * (M_PI / 3.), Axis::Y));
        hex.add_child(side);
    }
    return hex;
}

// ---------------------------------------------------------------- obj_file.rs
// Tokens are views into the file text: no allocation per line or per token (a 1 M-triangle OBJ has 4 M lines).
static void split_whitespace(std::string_view line, std::vector<std::string_view>& out) {
    out.clear();
    size_t i = 0, n = line.size();
    while (i < n) {
        while (i < n && std::isspace(static_cast<unsigned char>(line[i]))) ++i;
        size_t j = i;
        while (j < n && !std::isspace(static_cast<unsigned char>(line[j]))) ++j;
        if (j > i) out.push_back(line.substr(i, j - i));
        i = j;
    }
}
static bool parse_f64(std::string_view s, double& out) {  // str::parse::<f64>
    if (s.empty()) return false;
    for (char c : s)
        if (c == 'x' || c == 'X' || c == 'p' || c == 'P' || c == '(' || std::isspace(static_cast<unsigned char>(c)))
            return false;
    {   // fast lane: a plain decimal literal that std::from_chars takes whole (same correctly rounded value as strtod);
        // anything it stops early on, or finds out of range, is decided by strtod below as before
        double v;
        const std::from_chars_result r = std::from_chars(s.data(), s.data() + s.size(), v);
        if (r.ec == std::errc() && r.ptr == s.data() + s.size()) { out = v; return true; }
    }
    char small[64];
    std::string big;
    const char* z;
    if (s.size() < sizeof small) { std::memcpy(small, s.data(), s.size()); small[s.size()] = 0; z = small; }
    else { big.assign(s); z = big.c_str(); }
    char* end = nullptr;
    out = std::strtod(z, &end);
    return end == z + s.size();
}
static bool parse_usize(std::string_view s, size_t& out) {  // str::parse::<usize>
    size_t i = 0;
    if (!s.empty() && s[0] == '+') i = 1;
    if (i >= s.size()) return false;
    unsigned long long v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        unsigned long long nv = v * 10ull + static_cast<unsigned long long>(s[i] - '0');
        if (nv / 10ull != v) return false;
        v = nv;
    }
    out = static_cast<size_t>(v);
    return true;
}
template <class T>
static const T& idx(const std::vector<T>& v, size_t i) {
    if (i >= v.size()) throw Panic("index out of bounds");
    return v[i];
}

// The record loop of parse_obj_file (src/obj_file.rs:29-141).  What a face becomes goes to `sink`, so that the Shape
// tree (parse_obj_file) and the packed form (obj_file_to_group) come from the same code and draw the same ids.
template <class Sink>
static void parse_obj_records(const std::string& data, std::vector<Tuple>& vertices, std::vector<Tuple>& normals, int& lines_ignored,
                              Sink& sink) {
    vertices.push_back(ORIGIN);
    normals.push_back(Tuple::vector(0., 0., 0.));
    sink.group();
    size_t pos = 0;
    const std::string_view text(data);
    std::vector<std::string_view> record;
    std::vector<size_t> points, vertex_normals;
    while (pos < data.size()) {  // str::lines(): split on \n, strip one trailing \r
        size_t nl = text.find('\n', pos);
        std::string_view line = text.substr(pos, nl == std::string_view::npos ? std::string_view::npos : nl - pos);
        pos = (nl == std::string_view::npos) ? data.size() : nl + 1;
        if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
        split_whitespace(line, record);
        if (record.empty()) { lines_ignored += 1; continue; }
        if (record[0] == "v" || record[0] == "vn") {
            if (record.size() < 4) throw Panic("range end index 4 out of range for slice");
            double c[3];
            bool ok = true;
            for (int k = 0; k < 3 && ok; ++k) ok = parse_f64(record[size_t(k) + 1], c[k]);
            if (!ok) continue;
            if (record[0] == "v") vertices.push_back(Tuple::point(c[0], c[1], c[2]));
            else normals.push_back(Tuple::vector(c[0], c[1], c[2]));
        } else if (record[0] == "f") {
            points.clear();
            vertex_normals.clear();
            bool contains_vertex_normal = false, bad = false;
            for (size_t r = 1; r < record.size() && !bad; ++r) {
                const std::string_view rec = record[r];
                size_t start = 0, field = 0;
                while (true) {
                    size_t slash = rec.find('/', start);
                    std::string_view val = rec.substr(start, slash == std::string_view::npos ? std::string_view::npos : slash - start);
                    if (!val.empty()) {
                        size_t v;
                        if (!parse_usize(val, v)) { bad = true; break; }
                        if (field == 0) points.push_back(v);
                        else if (field == 2) { contains_vertex_normal = true; vertex_normals.push_back(v); }
                    }
                    if (slash == std::string_view::npos) break;
                    start = slash + 1;
                    ++field;
                }
            }
            if (bad) continue;
            if (contains_vertex_normal) {
                if (points.size() > 3) {  // fan_triangulation_smooth over the WHOLE vertex list (F3), src/obj_file.rs:153-162
                    sink.begin_fan();
                    for (size_t i = 2; i + 1 < vertices.size(); ++i)
                        sink.smooth(vertices[1], vertices.at(i), vertices.at(i + 1), normals.at(1), normals.at(i), normals.at(i + 1));
                    sink.end_fan();
                } else {
                    sink.begin_fan();
                    sink.smooth(idx(vertices, idx(points, 0)), idx(vertices, idx(points, 1)), idx(vertices, idx(points, 2)),
                                idx(normals, idx(vertex_normals, 0)), idx(normals, idx(vertex_normals, 1)), idx(normals, idx(vertex_normals, 2)));
                    sink.end_fan();
                }
            } else {
                if (points.size() > 3) {  // fan_triangulation over the WHOLE vertex list (F3), src/obj_file.rs:143-151
                    sink.begin_fan();
                    for (size_t i = 2; i + 1 < vertices.size(); ++i) sink.triangle(vertices[1], vertices[i], vertices[i + 1]);
                    sink.end_fan();
                } else {
                    sink.begin_fan();
                    sink.triangle(idx(vertices, idx(points, 0)), idx(vertices, idx(points, 1)), idx(vertices, idx(points, 2)));
                    sink.end_fan();
                }
            }
        } else if (record[0] == "g") {
            sink.group();
        } else {
            lines_ignored += 1;
        }
    }
}

Parser parse_obj_file(const std::string& data) {  // src/obj_file.rs:29-141
    Parser p;
    struct Sink {  // triangles of a face are all created first (each draws its id), then added to the current group
        Parser& p;
        std::vector<Shape> pending;
        void group() { p.groups.push_back(Shape::Group()); }
        void begin_fan() { pending.clear(); }
        void triangle(const Tuple& a, const Tuple& b, const Tuple& c) { pending.push_back(Shape::Triangle(a, b, c)); }
        void smooth(const Tuple& a, const Tuple& b, const Tuple& c, const Tuple& na, const Tuple& nb, const Tuple& nc) {
            pending.push_back(Shape::SmoothTriangle(a, b, c, na, nb, nc));
        }
        void end_fan() {
            for (Shape& t : pending) p.groups.back().add_child(t);
        }
    } sink{p, {}};
    parse_obj_records(data, p.vertices, p.normals, p.lines_ignored, sink);
    return p;
}

Shape obj_to_group(Parser&& parser) {  // src/obj_file.rs:164-172
    Shape main_group = Shape::Group();
    for (Shape& g : parser.groups) main_group.add_child(g);
    return main_group;
}

Shape obj_file_to_group(const std::string& data, int32_t counts[5]) {
    auto mesh = std::make_shared<PackedMesh>();
    struct Sink {
        PackedMesh& m;
        std::vector<PackedTri> pending;
        void group() { m.groups.push_back(PackedGroup{generate_object_id(), {}}); }
        void begin_fan() { pending.clear(); }
        void triangle(const Tuple& a, const Tuple& b, const Tuple& c) {  // Triangle::new, src/triangle.rs:38-56
            PackedTri t;
            t.id = generate_object_id();
            t.smooth = false;
            t.p1 = a; t.p2 = b; t.p3 = c;
            t.e1 = b - a;
            t.e2 = c - a;
            t.normal = t.e2.cross(t.e1).normalize();
            pending.push_back(t);
        }
        void smooth(const Tuple& a, const Tuple& b, const Tuple& c, const Tuple& na, const Tuple& nb, const Tuple& nc) {
            triangle(a, b, c);  // SmoothTriangle::new, src/smooth_triangle.rs:42-65: the same derived fields
            pending.back().smooth = true;
            pending.back().n1 = na; pending.back().n2 = nb; pending.back().n3 = nc;
        }
        void end_fan() { m.groups.back().tris.insert(m.groups.back().tris.end(), pending.begin(), pending.end()); }
    } sink{*mesh, {}};
    std::vector<Tuple> vertices, normals;
    int lines_ignored = 0;
    parse_obj_records(data, vertices, normals, lines_ignored, sink);
    int32_t tris = 0;
    mesh->id_first = mesh->groups.front().id;
    mesh->id_last = mesh->id_first - 1;
    for (const PackedGroup& g : mesh->groups) {
        tris += int32_t(g.tris.size());
        mesh->id_first = std::min(mesh->id_first, g.id);
        mesh->id_last = std::max(mesh->id_last, g.id);
        for (const PackedTri& t : g.tris) { mesh->id_first = std::min(mesh->id_first, t.id); mesh->id_last = std::max(mesh->id_last, t.id); }
    }
    if (counts) {
        counts[0] = lines_ignored;
        counts[1] = int32_t(vertices.size());
        counts[2] = int32_t(normals.size());
        counts[3] = int32_t(mesh->groups.size());
        counts[4] = tris;
    }
    return Shape::FromPackedMesh(std::move(mesh));  // obj_to_group: the main group's id is drawn last
}

// ---------------------------------------------------------------- canvas.rs
int Canvas::scale_color(double color_value, int max_color_value) {  // src/canvas.rs:86-98
    if (color_value > 1.0) return 255;
    if (color_value < 0.0) return 0;
    double scaled = color_value * double(max_color_value);
    if (std::isnan(scaled)) return 0;  // Rust `as i32` maps NaN to 0
    return int(std::round(scaled));
}

std::string Canvas::canvas_to_ppm() const {  // src/canvas.rs:33-84
    const int MAX_COLOR = 255;
    std::string ppm = "P3\n" + std::to_string(width_) + " " + std::to_string(height_) + "\n" +
                      std::to_string(MAX_COLOR) + "\n";
    std::string line;
    auto put = [&](const std::string& value) {
        if (line.size() + value.size() >= 70) {
            line.pop_back();
            line.push_back('\n');
            ppm += line;
            line.clear();
        }
        line += value + " ";
    };
    for (size_t pos = 0; pos < pixels_.size(); ++pos) {
        const Color& px = pixels_[pos];
        put(std::to_string(scale_color(px.red, MAX_COLOR)));
        put(std::to_string(scale_color(px.green, MAX_COLOR)));
        put(std::to_string(scale_color(px.blue, MAX_COLOR)));
        if ((pos + 1) % width_ == 0) {
            line.pop_back();
            line.push_back('\n');
            ppm += line;
            line.clear();
        }
    }
    return ppm;
}

// ---------------------------------------------------------------- world.rs
World World::default_world() {  // src/world.rs:20-39
    World w;
    w.lights.push_back(Light::point_light(Tuple::point(-10., 10., -10.), WHITE));
    Material m;
    m.color = Color{0.8, 1., 0.6};
    m.diffuse = 0.7;
    m.specular = 0.2;
    Shape s1 = Shape::Sphere();
    s1.set_material(m);
    Shape s2 = Shape::Sphere();
    s2.set_transform(scale(0.5, 0.5, 0.5));
    w.objects.push_back(std::move(s1));
    w.objects.push_back(std::move(s2));
    return w;
}

World::~World() { invalidate(); }

void World::invalidate() const {
    for (auto& kv : scenes_)
        if (kv.second) rt_scene_destroy(kv.second);
    scenes_.clear();
}

static void check_rt(int rc) {
    if (rc != RT_OK) throw Panic(std::string("rt error ") + std::to_string(rc) + ": " + rt_last_error());
}

rt_scene* World::device_scene(int device) const {
    auto it = scenes_.find(device);
    if (it != scenes_.end()) return it->second;
    FlatScene flat = flatten();
    rt_scene_desc d = flat.desc();
    rt_scene* scene = nullptr;
    check_rt(rt_scene_create(&d, device, &scene));
    scenes_[device] = scene;
    return scene;
}

std::vector<Intersection> World::intersect_world(const Ray& ray) const {  // src/world.rs:46-55
    rt_scene* s = device_scene();
    const int cap = 256;
    double o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
    int32_t count = 0;
    std::vector<double> t(cap);
    std::vector<int32_t> id(cap);
    check_rt(rt_intersect_world(s, 1, o, d, cap, &count, t.data(), id.data()));
    if (count > cap) throw Panic("intersect_world: more than 256 intersections on one ray");
    std::vector<Intersection> out;
    for (int i = 0; i < count; ++i) out.push_back({t[size_t(i)], id[size_t(i)]});
    return out;
}

Color World::color_at(const Ray& ray, int remaining) const {  // src/world.rs:81-90
    rt_scene* s = device_scene();
    double o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
    double rgb[3];
    check_rt(rt_color_at(s, 1, o, d, remaining, rgb, nullptr, nullptr, nullptr));
    return {rgb[0], rgb[1], rgb[2]};
}

std::vector<bool> World::is_shadowed(const Tuple& point) const {  // src/world.rs:92-107
    rt_scene* s = device_scene();
    double p[3] = {point.x, point.y, point.z};
    std::vector<uint8_t> out(lights.size());
    check_rt(rt_is_shadowed(s, 1, p, out.data()));
    return std::vector<bool>(out.begin(), out.end());
}

// ---------------------------------------------------------------- camera.rs
Camera::Camera(size_t hs, size_t vs, double fov) : hsize(hs), vsize(vs), field_of_view(fov) {  // src/camera.rs:21-35
    double half_view = std::tan(field_of_view / 2.);
    double aspect = double(hsize) / double(vsize);
    if (aspect > 1. || near_eq(aspect, 1.)) {
        half_width = half_view;
        half_height = half_view / aspect;
    } else {
        half_width = half_view * aspect;
        half_height = half_view;
    }
    pixel_size = (half_width * 2.) / double(hsize);
}

rt_camera_desc Camera::desc() const {
    rt_camera_desc d;
    std::memset(&d, 0, sizeof d);
    d.hsize = int32_t(hsize);
    d.vsize = int32_t(vsize);
    std::optional<Matrix> inv = transform.inverse();
    d.invertible = inv.has_value() ? 1 : 0;
    if (inv) std::memcpy(d.inverse, inv->data(), sizeof d.inverse);
    d.half_width = half_width;
    d.half_height = half_height;
    d.pixel_size = pixel_size;
    return d;
}

Canvas Camera::render(const World& world, int depth, std::vector<int32_t>* hit_ids, std::vector<uint8_t>* flags,
                      rt_stats* stats) const {  // src/camera.rs:62-74
    if (hsize == 0 || vsize == 0) throw Panic("attempt to subtract with overflow");  // `0..self.vsize - 1`
    rt_scene* s = world.device_scene();
    rt_camera_desc cd = desc();
    rt_render_opts opts;
    std::memset(&opts, 0, sizeof opts);
    opts.max_depth = depth;
    opts.rank = 0;
    opts.n_ranks = 1;
    std::vector<double> rgb(hsize * vsize * 3);
    if (hit_ids) hit_ids->assign(hsize * vsize, 0);
    if (flags) flags->assign(hsize * vsize, 0);
    check_rt(rt_render(s, &cd, &opts, rgb.data(), hit_ids ? hit_ids->data() : nullptr, flags ? flags->data() : nullptr,
                       stats));
    Canvas image(hsize, vsize);
    std::vector<Color>& px = image.pixels();
    for (size_t i = 0; i < px.size(); ++i) px[i] = Color{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    return image;
}

}  // namespace ray_tracer
