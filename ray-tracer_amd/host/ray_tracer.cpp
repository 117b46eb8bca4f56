// ray_tracer.cpp — host-side mirror of the reference crate's scene-construction API (see ray_tracer.hpp).
// Each block cites the reference file it follows.  No hot-path arithmetic lives here.
#include "ray_tracer.hpp"

#include <cstdlib>
#include <cstring>
#include <sstream>
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
      right(o.right ? std::make_unique<Shape>(*o.right) : nullptr), kind_(o.kind_), id_(o.id_) {}
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
    return shapes;
}
void Shape::add_child(Shape& child) {  // src/group.rs:45-50
    if (kind_ != ShapeKind::Group) throw Panic("Only groups can contain children.");
    child.set_parent(id_);
    shapes.push_back(child);
    update_reference();
}
bool Shape::includes(const Shape& other) const {  // src/shape.rs:422-435
    switch (kind_) {
        case ShapeKind::Group:
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
static std::vector<std::string> split_whitespace(const std::string& line) {
    std::vector<std::string> out;
    size_t i = 0, n = line.size();
    while (i < n) {
        while (i < n && std::isspace(static_cast<unsigned char>(line[i]))) ++i;
        size_t j = i;
        while (j < n && !std::isspace(static_cast<unsigned char>(line[j]))) ++j;
        if (j > i) out.emplace_back(line, i, j - i);
        i = j;
    }
    return out;
}
static bool parse_f64(const std::string& s, double& out) {  // str::parse::<f64>
    if (s.empty()) return false;
    for (char c : s)
        if (c == 'x' || c == 'X' || c == 'p' || c == 'P' || c == '(' || std::isspace(static_cast<unsigned char>(c)))
            return false;
    char* end = nullptr;
    out = std::strtod(s.c_str(), &end);
    return end == s.c_str() + s.size();
}
static bool parse_usize(const std::string& s, size_t& out) {  // str::parse::<usize>
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
static std::vector<Shape> fan_triangulation(const std::vector<Tuple>& v) {  // src/obj_file.rs:143-151 (F3: whole list)
    std::vector<Shape> t;
    for (size_t i = 2; i + 1 < v.size(); ++i) t.push_back(Shape::Triangle(v[1], v[i], v[i + 1]));
    return t;
}
static std::vector<Shape> fan_triangulation_smooth(const std::vector<Tuple>& v, const std::vector<Tuple>& n) {
    std::vector<Shape> t;  // src/obj_file.rs:153-162
    for (size_t i = 2; i + 1 < v.size(); ++i)
        t.push_back(Shape::SmoothTriangle(v[1], v.at(i), v.at(i + 1), n.at(1), n.at(i), n.at(i + 1)));
    return t;
}
template <class T>
static const T& idx(const std::vector<T>& v, size_t i) {
    if (i >= v.size()) throw Panic("index out of bounds");
    return v[i];
}

Parser parse_obj_file(const std::string& data) {  // src/obj_file.rs:29-141
    Parser p;
    p.vertices.push_back(ORIGIN);
    p.normals.push_back(Tuple::vector(0., 0., 0.));
    p.groups.push_back(Shape::Group());
    size_t pos = 0;
    while (pos < data.size()) {  // str::lines(): split on \n, strip one trailing \r
        size_t nl = data.find('\n', pos);
        std::string line = data.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
        pos = (nl == std::string::npos) ? data.size() : nl + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::vector<std::string> record = split_whitespace(line);
        if (record.empty()) { p.lines_ignored += 1; continue; }
        if (record[0] == "v" || record[0] == "vn") {
            if (record.size() < 4) throw Panic("range end index 4 out of range for slice");
            double c[3];
            bool ok = true;
            for (int k = 0; k < 3 && ok; ++k) ok = parse_f64(record[size_t(k) + 1], c[k]);
            if (!ok) continue;
            if (record[0] == "v") p.vertices.push_back(Tuple::point(c[0], c[1], c[2]));
            else p.normals.push_back(Tuple::vector(c[0], c[1], c[2]));
        } else if (record[0] == "f") {
            std::vector<size_t> points, vertex_normals;
            bool contains_vertex_normal = false, bad = false;
            for (size_t r = 1; r < record.size() && !bad; ++r) {
                const std::string& rec = record[r];
                size_t start = 0, field = 0;
                while (true) {
                    size_t slash = rec.find('/', start);
                    std::string val = rec.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (!val.empty()) {
                        size_t v;
                        if (!parse_usize(val, v)) { bad = true; break; }
                        if (field == 0) points.push_back(v);
                        else if (field == 2) { contains_vertex_normal = true; vertex_normals.push_back(v); }
                    }
                    if (slash == std::string::npos) break;
                    start = slash + 1;
                    ++field;
                }
            }
            if (bad) continue;
            Shape& group = p.groups.back();
            if (contains_vertex_normal) {
                if (points.size() > 3) {
                    for (Shape& t : fan_triangulation_smooth(p.vertices, p.normals)) group.add_child(t);
                } else {
                    Shape t = Shape::SmoothTriangle(idx(p.vertices, idx(points, 0)), idx(p.vertices, idx(points, 1)),
                                                    idx(p.vertices, idx(points, 2)), idx(p.normals, idx(vertex_normals, 0)),
                                                    idx(p.normals, idx(vertex_normals, 1)), idx(p.normals, idx(vertex_normals, 2)));
                    group.add_child(t);
                }
            } else {
                if (points.size() > 3) {
                    for (Shape& t : fan_triangulation(p.vertices)) group.add_child(t);
                } else {
                    Shape t = Shape::Triangle(idx(p.vertices, idx(points, 0)), idx(p.vertices, idx(points, 1)),
                                              idx(p.vertices, idx(points, 2)));
                    group.add_child(t);
                }
            }
        } else if (record[0] == "g") {
            p.groups.push_back(Shape::Group());
        } else {
            p.lines_ignored += 1;
        }
    }
    return p;
}

Shape obj_to_group(Parser&& parser) {  // src/obj_file.rs:164-172
    Shape main_group = Shape::Group();
    for (Shape& g : parser.groups) main_group.add_child(g);
    return main_group;
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
