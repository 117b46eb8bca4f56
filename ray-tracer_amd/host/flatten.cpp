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
#include <deque>
#include <limits>
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
    std::vector<const Shape*> node_shape;  // null for the nodes of a packed mesh (see visit_packed)
    std::vector<std::pair<Material, int32_t>> materials;  // content-deduplicated
    // Packed meshes flattened so far: their (contiguous) id range and where their nodes are, so that a later leaf with
    // one of those ids finds the earlier node Shape::eq has to be asked about.
    struct PackedSeen {
        const PackedMesh* mesh;
        int32_t main_node;
    };
    std::vector<PackedSeen> packed_seen;
    std::deque<Shape> unpacked_copies;  // packed groups that had to take the Shape-by-Shape path (kept alive for node_shape)
    int min_leaf_id = std::numeric_limits<int>::max(), max_leaf_id = std::numeric_limits<int>::min();  // of unpacked leaves

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

    static bool self_equal(const PackedTri& t) {  // Triangle / SmoothTriangle PartialEq of a value with itself (F14)
        bool eq = t.p1 == t.p1 && t.p2 == t.p2 && t.p3 == t.p3 && t.e1 == t.e1 && t.e2 == t.e2 && t.normal == t.normal;
        if (t.smooth) eq = eq && t.n1 == t.n1 && t.n2 == t.n2 && t.n3 == t.n3;
        return eq;
    }

    // Nodes of earlier packed meshes that carry object id `id` (a mesh cloned into the world twice, or a leaf of an
    // unpacked clone): the node and a Shape made from its record.
    std::vector<std::pair<int32_t, Shape>> packed_nodes_with_id(int id) const {
        std::vector<std::pair<int32_t, Shape>> out;
        for (const PackedSeen& ps : packed_seen) {
            if (id < ps.mesh->id_first || id > ps.mesh->id_last) continue;
            int32_t node = ps.main_node + 1;
            for (const PackedGroup& g : ps.mesh->groups) {
                for (size_t k = 0; k < g.tris.size(); ++k)
                    if (g.tris[k].id == id) out.emplace_back(node + 1 + int32_t(k), Shape::FromPackedTri(g.tris[k], g.id));
                node += 1 + int32_t(g.tris.size());
            }
        }
        return out;
    }

    // Group::bounds_of of the main group (src/group.rs:86-95): the union of its sub-groups' parent-space boxes
    static Bound packed_main_bounds(const PackedMesh& m) {
        const Bound tri_box = Bound::bounding_box_empty().transform(Matrix::identity(4));
        Bound b = Bound::bounding_box_empty();
        for (const PackedGroup& g : m.groups) {
            Bound gb = Bound::bounding_box_empty();
            for (size_t k = 0; k < g.tris.size() && k < 3; ++k) gb.add_box(tri_box);
            b.add_box(gb.transform(Matrix::identity(4)));
        }
        return b;
    }

    // The children of a Group that obj_file_to_group built, straight from the packed records: the same nodes, in the
    // same order, with the same values as visit() produces for the unpacked tree.
    void visit_packed(const Shape& main, int32_t main_node) {
        const PackedMesh& m = *main.packed();
        // per-triangle constants of this mesh.  Every triangle and every sub-group has the identity transform (xform 0).
        // Group::bounds_of of a sub-group (src/group.rs:86-95): the union of its triangles' parent-space boxes, which are
        // all the same box — the empty box (src/triangle.rs:93-95) through the identity transform
        const Bound tri_box = Bound::bounding_box_empty().transform(Matrix::identity(4));
        packed_seen.push_back({&m, main_node});
        for (const PackedGroup& g : m.groups) {
            const int32_t gn = int32_t(f.node_kind.size());
            Bound gb = Bound::bounding_box_empty();
            for (size_t k = 0; k < g.tris.size() && k < 3; ++k) gb.add_box(tri_box);  // identical boxes: a fixed point after the first
            f.node_kind.push_back(int32_t(ShapeKind::Group));
            f.node_object_id.push_back(g.id);
            f.node_parent.push_back(main_node);
            f.node_subtree_end.push_back(gn + 1 + int32_t(g.tris.size()));
            f.node_xform.push_back(0);
            f.node_ancestor.push_back(-1);
            f.node_material.push_back(-1);
            f.node_casts_shadow.push_back(1);
            f.node_eq_class.push_back(-1);
            f.node_param.push_back(int32_t(f.group_box.size() / 6));
            for (double v : {gb.minimum.x, gb.minimum.y, gb.minimum.z, gb.maximum.x, gb.maximum.y, gb.maximum.z}) f.group_box.push_back(v);
            node_shape.push_back(nullptr);
            if (g.tris.empty()) continue;
            const int32_t anc = ancestor_for(g.id);
            // Shape::get_material of a triangle: the material of its root ancestor in the registry (src/shape.rs:163-184)
            const RegistryEntry* e = registry_find(g.id);
            while (true) {
                if (!e) throw Panic("called `Option::unwrap()` on a `None` value (parent id not in PARENT_REFERENCES)");
                if (!e->parent) break;
                e = registry_find(*e->parent);
            }
            const int32_t mat = material_for(e->material);
            const size_t n0 = f.node_kind.size(), nt = g.tris.size();
            f.node_kind.resize(n0 + nt);
            f.node_object_id.resize(n0 + nt);
            f.node_parent.resize(n0 + nt, gn);
            f.node_subtree_end.resize(n0 + nt);
            f.node_xform.resize(n0 + nt, 0);
            f.node_ancestor.resize(n0 + nt, anc);
            f.node_material.resize(n0 + nt, mat);
            f.node_casts_shadow.resize(n0 + nt, 1);
            f.node_eq_class.resize(n0 + nt);
            f.node_param.resize(n0 + nt);
            node_shape.resize(n0 + nt, nullptr);
            const size_t t0 = f.tri_smooth.size();
            f.tri_p1.resize(3 * (t0 + nt)); f.tri_e1.resize(3 * (t0 + nt)); f.tri_e2.resize(3 * (t0 + nt)); f.tri_normal.resize(3 * (t0 + nt));
            f.tri_smooth.resize(t0 + nt);
            for (size_t k = 0; k < nt; ++k) {
                const PackedTri& t = g.tris[k];
                const size_t n = n0 + k, ti = t0 + k;
                f.node_kind[n] = int32_t(t.smooth ? ShapeKind::SmoothTriangle : ShapeKind::Triangle);
                f.node_object_id[n] = t.id;
                f.node_subtree_end[n] = int32_t(n) + 1;
                f.node_param[n] = int32_t(ti);
                f.node_eq_class[n] = self_equal(t) ? int32_t(n) : -1;
                const Tuple* src[4] = {&t.p1, &t.e1, &t.e2, &t.normal};
                double* dst[4] = {f.tri_p1.data(), f.tri_e1.data(), f.tri_e2.data(), f.tri_normal.data()};
                for (int a = 0; a < 4; ++a) { dst[a][3 * ti] = src[a]->x; dst[a][3 * ti + 1] = src[a]->y; dst[a][3 * ti + 2] = src[a]->z; }
                if (t.smooth) {
                    f.tri_smooth[ti] = int32_t(f.smooth_n1.size() / 3);
                    push3(f.smooth_n1, t.n1);
                    push3(f.smooth_n2, t.n2);
                    push3(f.smooth_n3, t.n3);
                } else {
                    f.tri_smooth[ti] = -1;
                }
            }
        }
    }

    void visit(const Shape& s, int32_t parent_node) {
        if (s.kind() == ShapeKind::Group && s.packed()) {
            // The packed path needs no Shape::eq look-ups only if none of the mesh's ids has been seen in this world yet
            // (neither in an earlier packed mesh nor in an unpacked leaf); else a copy is unpacked and visited Shape by Shape
            const PackedMesh& m = *s.packed();
            bool clash = !(m.id_last < min_leaf_id || m.id_first > max_leaf_id);
            for (const PackedSeen& ps : packed_seen) clash = clash || !(m.id_last < ps.mesh->id_first || m.id_first > ps.mesh->id_last);
            if (clash) {
                unpacked_copies.emplace_back(s);
                unpacked_copies.back().unpack();
                visit(unpacked_copies.back(), parent_node);
                return;
            }
        }
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
                Bound b = s.packed() ? packed_main_bounds(*s.packed()) : s.bounds_of();
                f.node_param[size_t(me)] = int32_t(f.group_box.size() / 6);
                for (double v : {b.minimum.x, b.minimum.y, b.minimum.z, b.maximum.x, b.maximum.y, b.maximum.z})
                    f.group_box.push_back(v);
                if (s.packed()) visit_packed(s, me);
                else
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
                min_leaf_id = std::min(min_leaf_id, s.get_id());
                max_leaf_id = std::max(max_leaf_id, s.get_id());
                if (s == s) {
                    int32_t cls = me;
                    auto& seen = nodes_of_object[s.get_id()];
                    bool found = false;
                    for (int32_t other : seen)
                        if (f.node_eq_class[size_t(other)] >= 0 && *node_shape[size_t(other)] == s) {
                            cls = f.node_eq_class[size_t(other)];
                            found = true;
                            break;
                        }
                    if (!found && !packed_seen.empty())
                        for (const auto& cand : packed_nodes_with_id(s.get_id()))
                            if (f.node_eq_class[size_t(cand.first)] >= 0 && cand.second == s) {
                                cls = f.node_eq_class[size_t(cand.first)];
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
