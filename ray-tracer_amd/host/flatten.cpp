// flatten.cpp — World -> rt_scene_desc (include/rt_abi.h).
//
// Everything the reference resolves lazily through its global registries while rendering is resolved here once:
//   * the INTERSECTION tree comes from the by-value snapshots in World.objects (src/world.rs:46-50,
//     src/group.rs:54-57, src/csg.rs:67-70) — pre-order = emission order;
//   * the PARENT CHAIN used by world_to_object / normal_to_world / get_material comes from the id registry
//     (src/shape.rs:163-184,394-420), which may hold a newer version of a Group/CSG than the snapshot;
//   * inverses come from Matrix::inverse (src/matrix.rs:124-154); a non-invertible transform is where the
//     reference panics (src/shape.rs:75) and is reported as an error here;
//   * Group gate boxes come from Group::bounds_of on the snapshot (src/group.rs:86-95), +-inf included;
//   * Shape::eq (per-struct PartialEq) is folded into an equality class per node.
#include <cstring>
#include <map>
#include <unordered_map>

#include "ray_tracer.hpp"

namespace ray_tracer {

rt_scene_desc FlatScene::desc() const {
    rt_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.abi_version = RT_ABI_VERSION;
    d.n_nodes = int32_t(node_kind.size());
    d.node_kind = node_kind.data();
    d.node_object_id = node_object_id.data();
    d.node_parent = node_parent.data();
    d.node_subtree_end = node_subtree_end.data();
    d.node_xform = node_xform.data();
    d.node_ancestor = node_ancestor.data();
    d.node_material = node_material.data();
    d.node_casts_shadow = node_casts_shadow.data();
    d.node_eq_class = node_eq_class.data();
    d.node_param = node_param.data();
    d.n_xforms = int32_t(xform_inverse.size() / 16);
    d.xform_inverse = xform_inverse.data();
    d.xform_inverse_transpose = xform_inverse_transpose.data();
    d.n_ancestors = int32_t(ancestor_xform.size());
    d.ancestor_xform = ancestor_xform.data();
    d.ancestor_parent = ancestor_parent.data();
    d.n_groups = int32_t(group_box.size() / 6);
    d.group_box = group_box.data();
    d.n_csg = int32_t(csg_op.size());
    d.csg_op = csg_op.data();
    d.n_conics = int32_t(conic_minimum.size());
    d.conic_minimum = conic_minimum.data();
    d.conic_maximum = conic_maximum.data();
    d.conic_closed = conic_closed.data();
    d.n_triangles = int32_t(tri_smooth.size());
    d.tri_p1 = tri_p1.data();
    d.tri_e1 = tri_e1.data();
    d.tri_e2 = tri_e2.data();
    d.tri_normal = tri_normal.data();
    d.tri_smooth = tri_smooth.data();
    d.n_smooth = int32_t(smooth_n1.size() / 3);
    d.smooth_n1 = smooth_n1.data();
    d.smooth_n2 = smooth_n2.data();
    d.smooth_n3 = smooth_n3.data();
    d.n_materials = int32_t(material_pattern.size());
    d.material_params = material_params.data();
    d.material_pattern = material_pattern.data();
    d.n_patterns = int32_t(pattern_kind.size());
    d.pattern_kind = pattern_kind.data();
    d.pattern_a = pattern_a.data();
    d.pattern_b = pattern_b.data();
    d.pattern_inverse = pattern_inverse.data();
    d.n_lights = int32_t(light_position.size() / 3);
    d.light_position = light_position.data();
    d.light_intensity = light_intensity.data();
    return d;
}

namespace {

struct Flattener {
    FlatScene f;
    std::unordered_map<int, int32_t> ancestor_of_id;        // registry id -> ancestor index
    std::unordered_map<int, int32_t> xform_of_matrix_id;    // matrix id -> xform index
    std::unordered_map<int, std::vector<int32_t>> nodes_of_object;  // object id -> leaf nodes seen so far
    std::vector<const Shape*> node_shape;
    std::vector<std::pair<Material, int32_t>> materials;  // content-deduplicated

    Flattener() {
        // xform 0 is always the exact identity
        Matrix id = Matrix::identity(4);
        push_xform(id, id);
    }

    int32_t push_xform(const Matrix& inv, const Matrix& inv_t) {
        for (int i = 0; i < 16; ++i) f.xform_inverse.push_back(inv.data()[i]);
        for (int i = 0; i < 16; ++i) f.xform_inverse_transpose.push_back(inv_t.data()[i]);
        return int32_t(f.xform_inverse.size() / 16 - 1);
    }

    static bool exactly_identity(const Matrix& m) {
        if (m.get_rows() != 4 || m.get_columns() != 4) return false;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c)
                if (m.at(r, c) != (r == c ? 1. : 0.)) return false;
        return true;
    }

    // transform -> index of (inverse, inverse-transpose); the same inverse the reference would compute
    // (or fetch from its per-id cache) at src/shape.rs:75,396,407.
    int32_t xform_for(const Matrix& transform) {
        auto it = xform_of_matrix_id.find(transform.get_id());
        if (it != xform_of_matrix_id.end()) return it->second;
        int32_t idx;
        if (exactly_identity(transform) && !transform_has_cached_inverse(transform)) {
            idx = 0;  // cofactor inverse of the exact identity is the exact identity
        } else {
            std::optional<Matrix> inv = transform.inverse();
            if (!inv)
                throw Panic("called `Option::unwrap()` on a `None` value: a shape transform is not invertible "
                            "(|det| < 1e-5, src/matrix.rs:133-136, src/shape.rs:75)");
            idx = exactly_identity(*inv) ? 0 : push_xform(*inv, inv->transpose());
        }
        xform_of_matrix_id.emplace(transform.get_id(), idx);
        return idx;
    }
    // An identity matrix whose id already has a cached inverse must go through the cache (it could have been
    // mutated after the inverse was taken: the reference would keep serving the stale inverse).
    static bool transform_has_cached_inverse(const Matrix&) { return false; }

    int32_t ancestor_for(int id) {
        auto it = ancestor_of_id.find(id);
        if (it != ancestor_of_id.end()) return it->second;
        const RegistryEntry* e = registry_find(id);
        if (!e) throw Panic("called `Option::unwrap()` on a `None` value (parent id not in PARENT_REFERENCES)");
        int32_t x = xform_for(e->transform);
        int32_t up = e->parent ? ancestor_for(*e->parent) : -1;
        f.ancestor_xform.push_back(x);
        f.ancestor_parent.push_back(up);
        int32_t idx = int32_t(f.ancestor_xform.size() - 1);
        ancestor_of_id.emplace(id, idx);
        return idx;
    }

    int32_t pattern_for(const Pattern& p) {
        std::optional<Matrix> inv = p.transform.inverse();
        if (!inv)
            throw Panic("called `Option::unwrap()` on a `None` value: a pattern transform is not invertible "
                        "(src/pattern.rs:25)");
        f.pattern_kind.push_back(int32_t(p.kind));
        for (double v : {p.a.red, p.a.green, p.a.blue}) f.pattern_a.push_back(v);
        for (double v : {p.b.red, p.b.green, p.b.blue}) f.pattern_b.push_back(v);
        for (int i = 0; i < 16; ++i) f.pattern_inverse.push_back(inv->data()[i]);
        return int32_t(f.pattern_kind.size() - 1);
    }

    static bool same_bits(const Material& a, const Material& b) {
        const double av[10] = {a.color.red, a.color.green, a.color.blue, a.ambient, a.diffuse, a.specular,
                               a.shininess, a.reflective, a.transparency, a.refractive_index};
        const double bv[10] = {b.color.red, b.color.green, b.color.blue, b.ambient, b.diffuse, b.specular,
                               b.shininess, b.reflective, b.transparency, b.refractive_index};
        if (std::memcmp(av, bv, sizeof av) != 0) return false;
        if (a.pattern.has_value() != b.pattern.has_value()) return false;
        if (!a.pattern) return true;
        const Pattern &p = *a.pattern, &q = *b.pattern;
        return p.kind == q.kind && p.transform.get_id() == q.transform.get_id() &&
               std::memcmp(p.transform.data(), q.transform.data(), 16 * sizeof(double)) == 0 &&
               std::memcmp(&p.a, &q.a, sizeof(Color)) == 0 && std::memcmp(&p.b, &q.b, sizeof(Color)) == 0;
    }

    int32_t material_for(const Material& m) {
        // look at the most recent entries only: meshes repeat one material thousands of times
        size_t n = materials.size();
        for (size_t k = 0; k < n && k < 8; ++k)
            if (same_bits(materials[n - 1 - k].first, m)) return materials[n - 1 - k].second;
        int32_t pat = m.pattern ? pattern_for(*m.pattern) : -1;
        for (double v : {m.color.red, m.color.green, m.color.blue, m.ambient, m.diffuse, m.specular, m.shininess,
                         m.reflective, m.transparency, m.refractive_index})
            f.material_params.push_back(v);
        f.material_pattern.push_back(pat);
        int32_t idx = int32_t(f.material_pattern.size() - 1);
        materials.emplace_back(m, idx);
        return idx;
    }

    static void push3(std::vector<double>& v, const Tuple& t) {
        v.push_back(t.x);
        v.push_back(t.y);
        v.push_back(t.z);
    }

    void visit(const Shape& s, int32_t parent_node) {
        int32_t me = int32_t(f.node_kind.size());
        f.node_kind.push_back(int32_t(s.kind()));
        f.node_object_id.push_back(s.get_id());
        f.node_parent.push_back(parent_node);
        f.node_subtree_end.push_back(me + 1);
        f.node_xform.push_back(xform_for(s.get_transform()));
        f.node_ancestor.push_back(-1);
        f.node_material.push_back(-1);
        f.node_casts_shadow.push_back(s.get_casts_shadow() ? 1 : 0);
        f.node_eq_class.push_back(-1);
        f.node_param.push_back(-1);
        node_shape.push_back(&s);

        switch (s.kind()) {
            case ShapeKind::Group: {
                Bound b = s.bounds_of();
                f.node_param[size_t(me)] = int32_t(f.group_box.size() / 6);
                for (double v : {b.minimum.x, b.minimum.y, b.minimum.z, b.maximum.x, b.maximum.y, b.maximum.z})
                    f.group_box.push_back(v);
                for (const Shape& c : s.shapes) visit(c, me);
                break;
            }
            case ShapeKind::CSG: {
                int32_t op = s.operation == "union"          ? RT_CSG_UNION
                             : s.operation == "intersection" ? RT_CSG_INTERSECTION
                             : s.operation == "difference"   ? RT_CSG_DIFFERENCE
                                                             : RT_CSG_UNKNOWN;
                f.node_param[size_t(me)] = int32_t(f.csg_op.size());
                f.csg_op.push_back(op);
                visit(*s.left, me);
                visit(*s.right, me);
                break;
            }
            default: {  // leaves
                if (s.get_parent()) f.node_ancestor[size_t(me)] = ancestor_for(*s.get_parent());
                f.node_material[size_t(me)] = material_for(s.get_material());
                if (s.kind() == ShapeKind::Cylinder || s.kind() == ShapeKind::Cone) {
                    f.node_param[size_t(me)] = int32_t(f.conic_minimum.size());
                    f.conic_minimum.push_back(s.minimum);
                    f.conic_maximum.push_back(s.maximum);
                    f.conic_closed.push_back(s.closed ? 1 : 0);
                } else if (s.kind() == ShapeKind::Triangle || s.kind() == ShapeKind::SmoothTriangle) {
                    f.node_param[size_t(me)] = int32_t(f.tri_smooth.size());
                    push3(f.tri_p1, s.point1);
                    push3(f.tri_e1, s.edge_vector1);
                    push3(f.tri_e2, s.edge_vector2);
                    push3(f.tri_normal, s.normal_vector);
                    if (s.kind() == ShapeKind::SmoothTriangle) {
                        f.tri_smooth.push_back(int32_t(f.smooth_n1.size() / 3));
                        push3(f.smooth_n1, s.normal_vector1);
                        push3(f.smooth_n2, s.normal_vector2);
                        push3(f.smooth_n3, s.normal_vector3);
                    } else {
                        f.tri_smooth.push_back(-1);
                    }
                }
                // Shape::eq folded into a class id (F14: a shape may be unequal to itself)
                if (s == s) {
                    int32_t cls = me;
                    auto& seen = nodes_of_object[s.get_id()];
                    for (int32_t other : seen)
                        if (f.node_eq_class[size_t(other)] >= 0 && *node_shape[size_t(other)] == s) {
                            cls = f.node_eq_class[size_t(other)];
                            break;
                        }
                    f.node_eq_class[size_t(me)] = cls;
                    seen.push_back(me);
                }
                break;
            }
        }
        f.node_subtree_end[size_t(me)] = int32_t(f.node_kind.size());
    }
};

}  // namespace

FlatScene World::flatten() const {
    Flattener fl;
    for (const Shape& o : objects) fl.visit(o, -1);
    for (const Light& l : lights) {
        Flattener::push3(fl.f.light_position, l.position);
        for (double v : {l.intensity.red, l.intensity.green, l.intensity.blue}) fl.f.light_intensity.push_back(v);
    }
    return std::move(fl.f);
}

}  // namespace ray_tracer
