// rt_program.cpp — descriptor validation, traversal-program compilation and BVH build (host side of
// rt_scene_create).  See rt_program.h for what the program is and why it cannot change results.
#include "rt_program.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <unordered_map>
#include <unordered_set>

namespace rt {
namespace {

constexpr double INF = std::numeric_limits<double>::infinity();

struct Compiler {
    const rt_scene_desc& d;
    Program& p;
    std::map<std::vector<int32_t>, int32_t> chain_index;
    std::map<std::pair<int32_t, int32_t>, int32_t> nchain_index;

    ProgramOpts opts;       // build_device >= 0 and device_bvh: large meshes are built on the GPU (rt_lbvh.hip)
    double t_dedup = 0., t_refs = 0., t_tree = 0.;  // seconds spent in build_mesh's phases (opts.verbose)
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    Compiler(const rt_scene_desc& desc, Program& prog) : d(desc), p(prog) {}

    int32_t chain_offset(const std::vector<int32_t>& chain) {
        auto it = chain_index.find(chain);
        if (it != chain_index.end()) return it->second;
        int32_t off = int32_t(p.chain_pool.size());
        p.chain_pool.push_back(int32_t(chain.size()));
        for (int32_t x : chain) p.chain_pool.push_back(x);
        chain_index.emplace(chain, off);
        p.max_chain = std::max(p.max_chain, int32_t(chain.size()));
        return off;
    }

    int32_t nchain_offset(int32_t node) {
        std::pair<int32_t, int32_t> key{d.node_xform[node], d.node_ancestor[node]};
        auto it = nchain_index.find(key);
        if (it != nchain_index.end()) return it->second;
        std::vector<int32_t> c{d.node_xform[node]};
        for (int32_t a = d.node_ancestor[node]; a >= 0; a = d.ancestor_parent[a]) c.push_back(d.ancestor_xform[a]);
        int32_t off = int32_t(p.chain_pool.size());
        p.chain_pool.push_back(int32_t(c.size()));
        for (int32_t x : c) p.chain_pool.push_back(x);
        nchain_index.emplace(key, off);
        p.max_chain = std::max(p.max_chain, int32_t(c.size()));
        return off;
    }

    static bool is_leaf_kind(int32_t k) { return k >= RT_SPHERE && k <= RT_SMOOTH_TRIANGLE; }
    static bool is_tri_kind(int32_t k) { return k == RT_TRIANGLE || k == RT_SMOOTH_TRIANGLE; }

    // ---- BVH -------------------------------------------------------------------------------------------
    struct TriRef {
        int32_t node;   // lowest node among the identical triangles this entry stands for
        int32_t dups;   // 0, or 1 + offset into p.mesh_dups of [count, node, node, ...]: the other nodes, ascending
        double lo[3], hi[3], c[3];
    };
    struct Box {
        double lo[3], hi[3];
        void reset() {
            for (int k = 0; k < 3; ++k) { lo[k] = INF; hi[k] = -INF; }
        }
        void add(const double* l, const double* h) {
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); }
        }
    };
    double pad = 0.;
    static constexpr int32_t EMPTY = std::numeric_limits<int32_t>::min();

    // binary tree of one mesh, before the collapse
    struct Node2 {
        Box box[2];
        int32_t child[2];  // >= 0: Node2 index; < 0: leaf reference (final encoding)
    };
    std::vector<Node2> n2;
    // BVH4 nodes of all meshes in creation order, with their depth; reordered by depth at the end
    std::vector<BvhNode> n4;
    std::vector<int32_t> n4_depth;
    std::vector<size_t> mesh_items;  // items whose `a` is a root index into n4
    int depth_bias = 0;              // collapse(): added to the depth a node is sorted by (the top-level tree: far below 0)
    Box last_mesh_box;               // padded bounds of the mesh build_mesh built last

    int32_t emit_leaf(std::vector<TriRef>& refs, size_t lo, size_t hi) {
        int32_t first = int32_t(p.mesh_tris.size());
        for (size_t i = lo; i < hi; ++i) {
            int32_t node = refs[i].node;
            int32_t t = d.node_param[node];
            MeshTri mt;
            for (int k = 0; k < 3; ++k) {
                mt.p1[k] = d.tri_p1[3 * t + k];
                mt.e1[k] = d.tri_e1[3 * t + k];
                mt.e2[k] = d.tri_e2[3 * t + k];
            }
            mt.node = node - mesh_base;  // relative to the mesh's group node: identical meshes share one tree (build_mesh)
            mt.smooth = (d.node_kind[node] == RT_SMOOTH_TRIANGLE ? 1 : 0) | (refs[i].dups << 1);
            p.mesh_tris.push_back(mt);
        }
        return ~((first << 2) | int32_t(hi - lo - 1));
    }

    // Builds [lo,hi) as a binary tree and returns the child reference plus its (padded) box.
    int32_t build(std::vector<TriRef>& refs, size_t lo, size_t hi, Box& box, int depth) {
        box.reset();
        Box cb;
        cb.reset();
        for (size_t i = lo; i < hi; ++i) {
            box.add(refs[i].lo, refs[i].hi);
            cb.add(refs[i].c, refs[i].c);
        }
        for (int k = 0; k < 3; ++k) { box.lo[k] -= pad; box.hi[k] += pad; }
        size_t n = hi - lo;
        if (n <= size_t(BVH_LEAF_MAX)) return emit_leaf(refs, lo, hi);

        // binned SAH over all three centroid axes; falls back to an index median on the longest axis when no bin
        // boundary separates the set
        int axis = 0;
        double ext = cb.hi[0] - cb.lo[0];
        for (int k = 1; k < 3; ++k)
            if (cb.hi[k] - cb.lo[k] > ext) { ext = cb.hi[k] - cb.lo[k]; axis = k; }
        size_t mid = lo + n / 2;
        bool split_done = false;
        int lg = 0;
        while ((size_t(1) << lg) < n) ++lg;
        // SAH may unbalance the tree; allow it only while a median-split subtree would still fit the depth bound
        if (ext > 0. && depth + lg + 2 < BVH2_MAX_DEPTH) {
            constexpr int NB = 32;
            auto area = [](const Box& b) {
                double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
                if (!(dx >= 0.)) return 0.;
                return dx * dy + dy * dz + dz * dx;
            };
            double best = INF;
            int best_b = -1, best_axis = -1;
            for (int ax = 0; ax < 3; ++ax) {
                const double e = cb.hi[ax] - cb.lo[ax];
                if (!(e > 0.)) continue;
                Box bb[NB];
                size_t cnt[NB] = {0};
                for (auto& b : bb) b.reset();
                const double scale = double(NB) / e;
                for (size_t i = lo; i < hi; ++i) {
                    int b = std::min(std::max(int((refs[i].c[ax] - cb.lo[ax]) * scale), 0), NB - 1);
                    bb[b].add(refs[i].lo, refs[i].hi);
                    cnt[b]++;
                }
                double right_area[NB];
                size_t right_cnt[NB];
                Box acc;
                acc.reset();
                size_t c = 0;
                for (int b = NB - 1; b > 0; --b) {
                    if (cnt[b]) acc.add(bb[b].lo, bb[b].hi);
                    c += cnt[b];
                    right_area[b] = area(acc);
                    right_cnt[b] = c;
                }
                acc.reset();
                c = 0;
                for (int b = 0; b + 1 < NB; ++b) {
                    if (cnt[b]) acc.add(bb[b].lo, bb[b].hi);
                    c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    double cost = area(acc) * double(c) + right_area[b + 1] * double(right_cnt[b + 1]);
                    if (cost < best) { best = cost; best_b = b; best_axis = ax; }
                }
            }
            if (best_b >= 0) {
                const double scale = double(NB) / (cb.hi[best_axis] - cb.lo[best_axis]);
                auto bin_of = [&](const TriRef& r) {
                    return std::min(std::max(int((r.c[best_axis] - cb.lo[best_axis]) * scale), 0), NB - 1);
                };
                auto it = std::partition(refs.begin() + long(lo), refs.begin() + long(hi),
                                         [&](const TriRef& r) { return bin_of(r) <= best_b; });
                size_t m = size_t(it - refs.begin());
                if (m > lo && m < hi) { mid = m; split_done = true; }
            }
        }
        if (!split_done) {
            if (ext > 0.)
                std::nth_element(refs.begin() + long(lo), refs.begin() + long(mid), refs.begin() + long(hi),
                                 [&](const TriRef& a, const TriRef& b) { return a.c[axis] < b.c[axis]; });
            // ext == 0: all centroids coincide (e.g. repeated fan triangles, F3) — halve by index
        }
        int32_t me = int32_t(n2.size());
        n2.emplace_back();
        Box b0, b1;
        int32_t c0 = build(refs, lo, mid, b0, depth + 1);
        int32_t c1 = build(refs, mid, hi, b1, depth + 1);
        Node2& nd = n2[size_t(me)];
        nd.box[0] = b0; nd.box[1] = b1;
        nd.child[0] = c0; nd.child[1] = c1;
        return me;
    }

    static float round_down(double x) {
        float f = float(x);
        if (double(f) > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
        return f;
    }
    static float round_up(double x) {
        float f = float(x);
        if (double(f) < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
        return f;
    }

    // Collapses the binary subtree under Node2 `root2` into one BVH4 node (children expanded breadth first, so every
    // BVH4 level consumes two binary levels wherever two exist) and recurses.  Returns the n4 index.
    int32_t collapse(int32_t root2, int depth) {
        struct Slot { int32_t ref; Box box; };
        std::vector<Slot> slots;
        const Node2 r = n2[size_t(root2)];
        auto expand = [&](const Slot& sl) {
            const Node2 c = n2[size_t(sl.ref)];
            slots.push_back({c.child[0], c.box[0]});
            slots.push_back({c.child[1], c.box[1]});
        };
        const Slot a{r.child[0], r.box[0]}, b{r.child[1], r.box[1]};
        if (a.ref >= 0) expand(a); else slots.push_back(a);
        if (b.ref >= 0) expand(b); else slots.push_back(b);
        if (slots.size() == 3) {  // one original was a leaf: a grandchild may be opened too
            for (size_t s = 0; s < 3; ++s)
                if (slots[s].ref >= 0) {
                    const Node2 c = n2[size_t(slots[s].ref)];
                    slots[s] = {c.child[0], c.box[0]};
                    slots.insert(slots.begin() + long(s) + 1, {c.child[1], c.box[1]});
                    break;
                }
        }
        int32_t me = int32_t(n4.size());
        n4.emplace_back();
        n4_depth.push_back(depth + depth_bias);
        if (depth_bias) p.tlas_levels = std::max(p.tlas_levels, depth + 1);   // the top-level tree: sorted in front of every mesh
        else p.bvh_max_depth = std::max(p.bvh_max_depth, depth + 1);
        BvhNode nd;
        std::memset(&nd, 0, sizeof nd);
        for (int c = 0; c < 4; ++c) {
            if (size_t(c) < slots.size()) {
                for (int k = 0; k < 3; ++k) { nd.lo[k][c] = round_down(slots[size_t(c)].box.lo[k]); nd.hi[k][c] = round_up(slots[size_t(c)].box.hi[k]); }
                nd.child[c] = slots[size_t(c)].ref >= 0 ? collapse(slots[size_t(c)].ref, depth + 1) : slots[size_t(c)].ref;
            } else {
                for (int k = 0; k < 3; ++k) { nd.lo[k][c] = float(INF); nd.hi[k][c] = float(-INF); }
                nd.child[c] = EMPTY;
            }
        }
        n4[size_t(me)] = nd;
        return me;
    }

    // Meshes already built, by content: the triangles of a mesh in node order, each as (node - group node, kind, p1, e1,
    // e2) — what the tree, its leaf records and its repeat lists are made of.  A group with the same content (the same
    // OBJ parsed again: C5's grid of teapots) takes the existing tree; only its ITEM_MESH differs (transform chain, group
    // node = the base the leaf records' relative node numbers are added to).  169 teapots: one tree that fits in LDS
    // instead of 161 k nodes of which 895 do.
    struct BuiltMesh {
        std::vector<unsigned char> content;
        int32_t root;
        Box box;
    };
    std::unordered_map<uint64_t, std::vector<BuiltMesh>> built_meshes;
    int32_t mesh_base = 0;  // group node of the mesh being built

    int32_t build_mesh(const std::vector<int32_t>& all_tri_nodes, int32_t group_node) {
        mesh_base = group_node;
        p.mesh_nodes.insert(p.mesh_nodes.end(), all_tri_nodes.begin(), all_tri_nodes.end());
        std::vector<unsigned char> content(all_tri_nodes.size() * (2 * sizeof(int32_t) + 9 * sizeof(double)));
        {
            unsigned char* w = content.data();
            for (int32_t node : all_tri_nodes) {
                const int32_t rel = node - group_node, kind = d.node_kind[node], t = d.node_param[node];
                std::memcpy(w, &rel, 4); std::memcpy(w + 4, &kind, 4);
                std::memcpy(w + 8, d.tri_p1 + 3 * size_t(t), 24); std::memcpy(w + 32, d.tri_e1 + 3 * size_t(t), 24);
                std::memcpy(w + 56, d.tri_e2 + 3 * size_t(t), 24);
                w += 80;
            }
        }
        uint64_t h = 1469598103934665603ull;  // FNV-1a over 8-byte words
        for (size_t i = 0; i + 8 <= content.size(); i += 8) { uint64_t v; std::memcpy(&v, content.data() + i, 8); h = (h ^ v) * 1099511628211ull; }
        std::vector<BuiltMesh>& same_hash = built_meshes[h];
        if (opts.share_meshes)
            for (const BuiltMesh& m : same_hash)
                if (m.content == content) {
                    last_mesh_box = m.box;
                    ++p.meshes_shared;
                    return m.root;
                }
        const int32_t root = build_mesh_tree(all_tri_nodes);
        if (opts.share_meshes) same_hash.push_back(BuiltMesh{std::move(content), root, last_mesh_box});
        return root;
    }

    int32_t build_mesh_tree(const std::vector<int32_t>& all_tri_nodes) {
        // Triangles with bit-identical p1, e1, e2 and kind give bit-identical test results for every ray: they are
        // tested once and the hit is emitted for each of their nodes (the reference's parser repeats triangles by the
        // hundred on polygon faces, F3: teapot-low.obj is 15,136 triangles but 151 distinct ones).
        std::vector<int32_t> tri_nodes;
        std::vector<int32_t> dups_of;
        const double t0 = now();
        {
            struct Key {
                double v[9];
                int32_t kind;
                bool operator==(const Key& o) const { return std::memcmp(v, o.v, sizeof v) == 0 && kind == o.kind; }
            };
            struct KeyHash {
                size_t operator()(const Key& k) const {  // FNV-1a over the bit patterns
                    uint64_t h = 1469598103934665603ull;
                    const unsigned char* b = reinterpret_cast<const unsigned char*>(k.v);
                    for (size_t i = 0; i < sizeof k.v; ++i) { h ^= b[i]; h *= 1099511628211ull; }
                    h ^= uint64_t(uint32_t(k.kind));
                    h *= 1099511628211ull;
                    return size_t(h);
                }
            };
            std::unordered_map<Key, size_t, KeyHash> seen;
            seen.reserve(all_tri_nodes.size());
            std::vector<std::vector<int32_t>> extra;
            for (int32_t node : all_tri_nodes) {
                int32_t t = d.node_param[node];
                Key k;
                for (int j = 0; j < 3; ++j) { k.v[j] = d.tri_p1[3 * t + j]; k.v[3 + j] = d.tri_e1[3 * t + j]; k.v[6 + j] = d.tri_e2[3 * t + j]; }
                k.kind = d.node_kind[node];
                auto ins = seen.emplace(k, tri_nodes.size());
                if (ins.second) { tri_nodes.push_back(node); extra.emplace_back(); }
                else extra[ins.first->second].push_back(node);  // all_tri_nodes ascends, so does every list
            }
            dups_of.assign(tri_nodes.size(), 0);
            for (size_t i = 0; i < tri_nodes.size(); ++i) {
                if (extra[i].empty()) continue;
                dups_of[i] = int32_t(p.mesh_dups.size()) + 1;
                p.mesh_dups.push_back(int32_t(extra[i].size()));
                for (int32_t other : extra[i]) p.mesh_dups.push_back(other - mesh_base);  // relative, like MeshTri::node
            }
        }
        const double t1 = now();
        t_dedup += t1 - t0;
        std::vector<TriRef> refs(tri_nodes.size());
        double maxabs = 0.;
        for (size_t i = 0; i < tri_nodes.size(); ++i) {
            TriRef& r = refs[i];
            r.node = tri_nodes[i];
            r.dups = dups_of[i];
            int32_t t = d.node_param[r.node];
            for (int k = 0; k < 3; ++k) {
                double a = d.tri_p1[3 * t + k], b = a + d.tri_e1[3 * t + k], c = a + d.tri_e2[3 * t + k];
                r.lo[k] = std::min(a, std::min(b, c));
                r.hi[k] = std::max(a, std::max(b, c));
                r.c[k] = 0.5 * r.lo[k] + 0.5 * r.hi[k];
                maxabs = std::max(maxabs, std::max(std::fabs(r.lo[k]), std::fabs(r.hi[k])));
            }
        }
        // Conservative padding: the reference accepts u,v in [0,1] as COMPUTED (relative error ~1e-16) and our
        // p2/p3 are re-derived from edges; 1e-7 of the mesh scale is nine orders of magnitude above both.
        pad = 1e-7 * maxabs + 1e-300;
        const double t2 = now();
        t_refs += t2 - t1;
        struct Stop { double& acc; double from; ~Stop() { acc += now() - from; } } stop{t_tree, t2};
        n2.clear();
        Box rb;
        int32_t root = 0;
        if (opts.build_device >= 0 && opts.device_bvh && refs.size() >= size_t(LBVH_MIN_TRIANGLES)) {
            // the whole tree on the GPU (rt_lbvh.hip): Morton order, radix tree, boxes, leaves, BVH4 collapse
            std::vector<double> tlo(refs.size() * 3), thi(refs.size() * 3);
            for (size_t i = 0; i < refs.size(); ++i)
                for (int k = 0; k < 3; ++k) { tlo[3 * i + size_t(k)] = refs[i].lo[k]; thi[3 * i + size_t(k)] = refs[i].hi[k]; }
            std::vector<int32_t> order, level_start;
            std::vector<BvhNode> nodes;
            double root_box[6];
            const int32_t tri_base = int32_t(p.mesh_tris.size());
            if (lbvh_build(opts.build_device, refs.size(), tlo.data(), thi.data(), pad, tri_base, (BVH_STACK - 1) / 3, order, nodes, level_start,
                           root_box)) {
                bool ok = order.size() == refs.size() && !nodes.empty() && level_start.size() >= 2;
                std::vector<char> seen(refs.size(), 0);
                for (size_t q = 0; q < order.size() && ok; ++q) {
                    const size_t i = size_t(order[q]);
                    ok = order[q] >= 0 && i < refs.size() && !seen[i];
                    if (ok) seen[i] = 1;
                }
                if (ok) {
                    std::vector<TriRef> sorted(refs.size());
                    for (size_t q = 0; q < order.size(); ++q) sorted[q] = refs[size_t(order[q])];
                    emit_leaf(sorted, 0, sorted.size());  // the MeshTri records in leaf order; the device encoded the leaf links
                    const int32_t base = int32_t(n4.size());
                    for (size_t lv = 0; lv + 1 < level_start.size(); ++lv)
                        for (int32_t i = level_start[lv]; i < level_start[lv + 1]; ++i) {
                            BvhNode nd = nodes[size_t(i)];
                            for (int c = 0; c < 4; ++c)
                                if (nd.child[c] >= 0) nd.child[c] += base;
                            n4.push_back(nd);
                            n4_depth.push_back(int(lv));
                        }
                    p.bvh_max_depth = std::max(p.bvh_max_depth, int32_t(level_start.size()) - 1);
                    for (int k = 0; k < 3; ++k) { last_mesh_box.lo[k] = root_box[k]; last_mesh_box.hi[k] = root_box[3 + k]; }
                    ++p.meshes_built_on_gpu;
                    return base;
                }
            }
        }
        root = build(refs, 0, refs.size(), rb, 0);
        last_mesh_box = rb;
        if (root < 0) {  // the whole mesh is one leaf: wrap it in a node
            Node2 nd;
            nd.box[0] = rb;
            nd.box[1].reset();
            nd.child[0] = root;
            nd.child[1] = EMPTY;
            n2.push_back(nd);
            root = int32_t(n2.size()) - 1;
        }
        return collapse(root, 0);
    }

    // Orders the BVH4 nodes of all meshes by depth (stable), rewrites child links and the mesh items' roots.
    void finish_bvh() {
        std::vector<int32_t> order(n4.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = int32_t(i);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return n4_depth[size_t(a)] < n4_depth[size_t(b)]; });
        std::vector<int32_t> where(n4.size());
        for (size_t i = 0; i < order.size(); ++i) where[size_t(order[i])] = int32_t(i);
        p.bvh.resize(n4.size());
        for (size_t i = 0; i < order.size(); ++i) {
            BvhNode nd = n4[size_t(order[i])];
            for (int c = 0; c < 4; ++c)
                if (nd.child[c] >= 0) nd.child[c] = where[size_t(nd.child[c])];
            p.bvh[i] = nd;
        }
        for (size_t it : mesh_items) p.items[it].a = where[size_t(p.items[it].a)];
        if (p.tlas_root >= 0) p.tlas_root = where[size_t(p.tlas_root)];
    }

    // ---- world-space bounds of top-level objects (ITEM_CULL) --------------------------------------------------
    struct WBox {
        double lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
        bool bounded = true;  // false: unbounded or not computable — never culled
        // How far (L-infinity, world units) a ray's origin may be from anything in the box for the box to be trusted: the
        // reference's quadratics lose ~1.1e-16 |o_obj|^2 in object space (c = o.o - 1, b^2 - 4ac), so from ~3e4 object
        // radii away it reports tangent hits further outside the unit shape than the 1e-7 padding of the boxes.  Each
        // leaf allows 1e3 object units (error ~1e-10); 1 / prod ||inverse||_F bounds the chain's smallest scale from below.
        double reach = INF;
        void add(const WBox& o) {
            bounded = bounded && o.bounded;
            reach = std::min(reach, o.reach);
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); }
        }
    };
    std::map<int32_t, std::vector<double>> forward_cache;  // xform index -> 12 doubles of the forward matrix, empty: singular
    const std::vector<double>& forward_of(int32_t x) {
        auto it = forward_cache.find(x);
        if (it != forward_cache.end()) return it->second;
        const double* m = d.xform_inverse + size_t(x) * 16;  // affine: rows 0..2 are [A | t], forward = [A^-1 | -A^-1 t]
        const double a = m[0], b = m[1], c = m[2], e = m[4], f = m[5], g = m[6], h = m[8], i = m[9], j = m[10];
        const double det = a * (f * j - g * i) - b * (e * j - g * h) + c * (e * i - f * h);
        std::vector<double> out;
        if (std::isfinite(det) && std::fabs(det) > 0.) {
            const double inv[9] = {(f * j - g * i) / det, (c * i - b * j) / det, (b * g - c * f) / det,
                                   (g * h - e * j) / det, (a * j - c * h) / det, (c * e - a * g) / det,
                                   (e * i - f * h) / det, (b * h - a * i) / det, (a * f - b * e) / det};
            out.resize(12);
            for (int r = 0; r < 3; ++r) {
                for (int k = 0; k < 3; ++k) out[size_t(4 * r + k)] = inv[3 * r + k];
                out[size_t(4 * r + 3)] = -(inv[3 * r] * m[3] + inv[3 * r + 1] * m[7] + inv[3 * r + 2] * m[11]);
            }
        }
        return forward_cache.emplace(x, out).first->second;
    }
    double chain_min_scale(const std::vector<int32_t>& chain) {  // lower bound of the smallest singular value of the forward chain
        double prod = 1.;
        for (int32_t x : chain) {
            const double* m = d.xform_inverse + size_t(x) * 16;
            double f2 = 0.;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) f2 += m[4 * r + c] * m[4 * r + c];
            prod *= std::sqrt(f2);
        }
        return prod > 0. && std::isfinite(prod) ? 1. / prod : 0.;
    }
    double chain_max_scale(const std::vector<int32_t>& chain) {  // upper bound of the largest singular value of the forward chain
        double prod = 1.;
        for (int32_t x : chain) {
            const std::vector<double>& fm = forward_of(x);
            if (fm.empty()) return INF;
            double f2 = 0.;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) f2 += fm[size_t(4 * r + c)] * fm[size_t(4 * r + c)];
            prod *= std::sqrt(f2);
        }
        return std::isfinite(prod) ? prod : INF;
    }
    // what a leaf under `chain` can leak beyond its box for a ray from D away (Program::far_a / far_b); kind: 0 quadratic
    // (sphere, cylinder), 1 cube, 2 triangles
    void note_far_leak(const std::vector<int32_t>& chain, int kind) {
        constexpr double EPS64 = 1.1102230246251565e-16, SAFETY = 8.;
        const double smin = chain_min_scale(chain), smax = chain_max_scale(chain);
        if (!(smin > 0.) || !std::isfinite(smax)) { p.far_a = INF; p.far_b = INF; return; }
        const double ratio = smax / smin;
        if (kind == 0) {
            // |o_obj| <= sqrt(3) D / smin radii; disc = b^2 - 4ac turns non-negative up to 2 eps R^2 outside the unit shape;
            // a near-tangent root is off by sqrt(8 eps) R / |d| in t, i.e. 3e-8 R object units along the ray
            p.far_a = std::max(p.far_a, SAFETY * 2. * EPS64 * 3. * smax / (smin * smin));
            p.far_b = std::max(p.far_b, SAFETY * 4e-8 * 1.7320508075688774 * ratio);
        } else if (kind == 1) {
            // EPSILON |t| object units with |t| <= 2 sqrt(3) D for directions at least 1/2 long
            p.far_b = std::max(p.far_b, SAFETY * 2. * 1.7320508075688774 * 0.00001 * smax);
        } else {
            p.far_b = std::max(p.far_b, SAFETY * 1e3 * EPS64 * ratio);  // u, v lose ~eps |o| / edge: lateral error ~eps |o|, x 1e3
        }
    }
    // object-space box -> world, through the non-identity transforms `chain` (outermost first); `size`: the object-space
    // length the leaf's arithmetic error is relative to (1 for the unit shapes)
    WBox to_world(const double* lo, const double* hi, const std::vector<int32_t>& chain, double size = 1.) {
        WBox w;
        w.reach = 1e3 * size * chain_min_scale(chain) / std::sqrt(3.);
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) { w.bounded = false; return w; }
        double pts[8][3];
        for (int c = 0; c < 8; ++c)
            for (int k = 0; k < 3; ++k) pts[c][k] = ((c >> k) & 1) ? hi[k] : lo[k];
        for (size_t lv = chain.size(); lv-- > 0;) {  // innermost first
            const std::vector<double>& fm = forward_of(chain[lv]);
            if (fm.empty()) { w.bounded = false; return w; }
            for (auto& p : pts) {
                const double q[3] = {p[0], p[1], p[2]};
                for (int r = 0; r < 3; ++r) p[r] = fm[size_t(4 * r)] * q[0] + fm[size_t(4 * r + 1)] * q[1] + fm[size_t(4 * r + 2)] * q[2] + fm[size_t(4 * r + 3)];
            }
        }
        for (const auto& p : pts)
            for (int k = 0; k < 3; ++k) {
                if (!std::isfinite(p[k])) { w.bounded = false; return w; }
                w.lo[k] = std::min(w.lo[k], p[k]);
                w.hi[k] = std::max(w.hi[k], p[k]);
            }
        return w;
    }
    WBox world_box(int32_t node, std::vector<int32_t> chain) {
        int32_t x = d.node_xform[node];
        if (x != 0) chain.push_back(x);
        const int32_t kind = d.node_kind[node];
        const double unit_lo[3] = {-1., -1., -1.}, unit_hi[3] = {1., 1., 1.};
        switch (kind) {
            // (sphere and cylinder: `size` 10 = trusted from 1e4 object radii.  A quadratic computed from an origin R radii away
            // reports tangent hits up to ~2 eps R^2 outside the unit shape: 2e-8 at 1e4, a tenth of the boxes' 1e-7 padding of the
            // object's own extent.  Round 2 allowed 1e3, which left a world of small spheres on a large floor with most of
            // its shadow rays untrusted.)
            case RT_SPHERE: note_far_leak(chain, 0); return to_world(unit_lo, unit_hi, chain, 10.);
            case RT_CUBE: {
                // check_axis treats an object-space direction component below EPSILON as parallel (src/cube.rs:64-78): that
                // axis' slab is then all of t when the ORIGIN lies between its planes, so the ray may drift out of the slab
                // by |d_k| |t| < EPSILON |t| (object units) before it meets the other axes' faces, and the reference reports
                // that hit — outside the unit cube, arbitrarily far for a distant origin (found by tests/test_gpu_fuzz.py: a
                // cube scaled by 17 seen from 1e5 away).  What the reference accepts at parameter t lies in the cube grown
                // by EPSILON |t|, so a box grown by CUBE_PAD covers every hit with |t| <= CUBE_PAD / EPSILON; and a ray whose
                // origin is within CUBE_REACH (L-infinity) of everything in the box, with a direction at least 1/2 long,
                // meets the box only at |t| <= 2 sqrt(3) CUBE_REACH <= CUBE_PAD / EPSILON.  Other rays do not trust the ranges
                // (cull_reach, cull_min_dir2).  Round 2 left cubes unbounded: C4's ranges then held three objects out of nine.
                constexpr double CUBE_PAD = 0.05, CUBE_REACH = 1400.;
                static_assert(2. * 1.7320508075688774 * CUBE_REACH <= CUBE_PAD / 0.00001, "cube padding does not cover its reach");
                const double h = 1. + CUBE_PAD;
                const double lo[3] = {-h, -h, -h}, hi[3] = {h, h, h};
                WBox w = to_world(lo, hi, chain);
                w.reach = std::min(w.reach, CUBE_REACH);
                // a cube that large, or that far out, would take the whole program's reach (a minimum) down with it: it
                // stays unbounded, as every cube was before
                for (int k = 0; k < 3 && w.bounded; ++k)
                    if (std::max(std::fabs(w.lo[k]), std::fabs(w.hi[k])) > CUBE_REACH / 2.) w.bounded = false;
                if (w.bounded) { p.cull_min_dir2 = 0.25; note_far_leak(chain, 1); }
                return w;
            }
            case RT_PLANE: { WBox w; w.bounded = false; return w; }
            case RT_CONE: {
                // never bounded: a ray parallel to the cone's side gets its single hit WITHOUT the minimum < y < maximum
                // test (src/cone.rs:61-64), i.e. anywhere on the infinite double cone
                WBox w;
                w.bounded = false;
                return w;
            }
            case RT_CYLINDER: {
                const int32_t q = d.node_param[node];
                const double mn = d.conic_minimum[q], mx = d.conic_maximum[q];
                // the caps accept x^2 + z^2 < 1 or within EPSILON of 1 (src/cylinder.rs:96-101): radius^2 up to 1 + 1e-5
                const double r = std::sqrt(1. + 2e-5);
                const double lo[3] = {-r, std::min(mn, mx), -r}, hi[3] = {r, std::max(mn, mx), r};
                note_far_leak(chain, 0);
                return to_world(lo, hi, chain, 10.);
            }
            case RT_TRIANGLE: case RT_SMOOTH_TRIANGLE: {
                note_far_leak(chain, 2);
                const int32_t t = d.node_param[node];
                double lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    const double a = d.tri_p1[3 * t + k], b = a + d.tri_e1[3 * t + k], c = a + d.tri_e2[3 * t + k];
                    lo[k] = std::min(a, std::min(b, c));
                    hi[k] = std::max(a, std::max(b, c));
                }
                // Moller-Trumbore's error is linear in the distance (u, v lose ~1e-16 |o| / edge, i.e. ~1e-16 |o| laterally),
                // against a padding of 1e-7 of the coordinates: trusted up to 1e6 extents
                return to_world(lo, hi, chain, 1e3 * std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2])));
            }
            case RT_GROUP: {
                // children without a transform of their own share this chain: unite them in object space first
                WBox w;
                double lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
                bool any_plain = false;
                for (int32_t c = node + 1; c < d.node_subtree_end[node]; c = d.node_subtree_end[c]) {
                    if (is_tri_kind(d.node_kind[c]) && d.node_xform[c] == 0) {
                        const int32_t t = d.node_param[c];
                        for (int k = 0; k < 3; ++k) {
                            const double a = d.tri_p1[3 * t + k], b = a + d.tri_e1[3 * t + k], e = a + d.tri_e2[3 * t + k];
                            lo[k] = std::min(lo[k], std::min(a, std::min(b, e)));
                            hi[k] = std::max(hi[k], std::max(a, std::max(b, e)));
                        }
                        any_plain = true;
                    } else {
                        w.add(world_box(c, chain));
                    }
                }
                // (triangles: see RT_TRIANGLE above; the mesh's own extent is the scale its padding is relative to)
                if (any_plain) { note_far_leak(chain, 2); w.add(to_world(lo, hi, chain, 1e3 * std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2])))); }
                return w;
            }
            default: {  // CSG: whatever survives the filter lies on an operand
                const int32_t left = node + 1, right = d.node_subtree_end[left];
                WBox w = world_box(left, chain);
                w.add(world_box(right, chain));
                return w;
            }
        }
    }
    // Emits the bounded top-level objects `roots[lo, hi)` as a tree of ITEM_CULL ranges (median split of the box
    // centres on their widest axis).  Only worth it in a world of many objects: a range costs every ray one more item
    // (a dependent scalar fetch), and with the headline scene's three bounded objects that lost 4 % of the frame, while
    // the 32-object grid of C5 renders 1.85x faster with it and the 8-object C4 1.17x.
    static constexpr size_t CULL_MIN_OBJECTS = 6;
    void emit_culled(std::vector<std::pair<int32_t, WBox>>& roots, size_t lo, size_t hi, bool always = false) {
        if (hi - lo == 0) return;
        if (!always && (roots.size() < CULL_MIN_OBJECTS || !opts.cull_ranges)) {
            for (size_t i = lo; i < hi; ++i) emit(roots[i].first, {});
            return;
        }
        WBox all;
        for (size_t i = lo; i < hi; ++i) all.add(roots[i].second);
        int32_t cull = -1;
        {
            Item it{};
            it.type = ITEM_CULL;
            double* b = cull_box(it);
            for (int k = 0; k < 3; ++k) {
                // hits are computed in object space from the inverse matrices: the bounds are padded far beyond that
                // arithmetic's error (as the BVH boxes are)
                const double pad = 1e-7 * std::max(std::max(std::fabs(all.lo[k]), std::fabs(all.hi[k])), all.hi[k] - all.lo[k]) + 1e-300;
                b[k] = all.lo[k] - pad;
                b[3 + k] = all.hi[k] + pad;
            }
            // the ranges are trusted only for rays that start within `reach` (L-infinity) of everything in them: folded
            // into one bound on the origin's largest |coordinate| for the whole program
            double far_plane = 0.;
            for (int k = 0; k < 6; ++k) far_plane = std::max(far_plane, std::fabs(b[k]));
            p.far_extent = std::max(p.far_extent, far_plane);
            const double r = std::max(0., all.reach - far_plane);
            p.cull_reach = p.has_cull ? std::min(p.cull_reach, r) : r;
            cull = int32_t(p.items.size());
            p.items.push_back(it);
            p.has_cull = true;
        }
        if (hi - lo == 1) {
            emit(roots[lo].first, {});
        } else {
            int axis = 0;
            double ext = -1.;
            for (int k = 0; k < 3; ++k) {
                double cl = INF, ch = -INF;
                for (size_t i = lo; i < hi; ++i) {
                    const double c = 0.5 * roots[i].second.lo[k] + 0.5 * roots[i].second.hi[k];
                    cl = std::min(cl, c);
                    ch = std::max(ch, c);
                }
                if (ch - cl > ext) { ext = ch - cl; axis = k; }
            }
            const size_t mid = lo + (hi - lo) / 2;
            std::nth_element(roots.begin() + long(lo), roots.begin() + long(mid), roots.begin() + long(hi),
                             [&](const std::pair<int32_t, WBox>& a, const std::pair<int32_t, WBox>& b) {
                                 return a.second.lo[axis] + a.second.hi[axis] < b.second.lo[axis] + b.second.hi[axis];
                             });
            emit_culled(roots, lo, mid, always);
            emit_culled(roots, mid, hi, always);
        }
        if (cull >= 0) p.items[size_t(cull)].skip = int32_t(p.items.size());
    }

    // The same bounded objects as a tree instead of nested ranges (worlds of >= TLAS_MIN_OBJECTS of them): each object
    // becomes [ITEM_CULL(its own padded world box, skip = end of the object), the object's items), and a BVH4 over the
    // boxes — binary median splits of the box centres collapsed like the mesh trees, boxes rounded outward to f32 —
    // names the ranges a ray has to visit.  The bounds are the cull ranges' bounds (world_box(): the reference's
    // ACCEPTANCE rules, not the ideal shapes), so are the padding and the reach guard: what holds for a range holds here.
    int32_t tlas_build(std::vector<std::pair<int32_t, WBox>>& roots, size_t lo, size_t hi, Box& box) {
        box.reset();
        for (size_t i = lo; i < hi; ++i) box.add(cull_box(p.items[size_t(tlas_header[i])]), cull_box(p.items[size_t(tlas_header[i])]) + 3);
        if (hi - lo == 1) {
            // an object that is nothing but one mesh under its transforms — [ITEM_CULL, ITEM_MESH) — is an INSTANCE: the
            // link names the ITEM_MESH itself and the kernels let every lane walk its own instance at the same time
            // (rt_kernels.hip.h: traverse), instead of the wavefront visiting the objects one after the other
            // ... and so is a bounded primitive that is an object of its own, [ITEM_CULL, ITEM_LEAF): each lane intersects the
            // sphere, cube or cylinder its own ray reached
            const int32_t h = tlas_header[lo];
            bool instance = false;
            if (RT_TLAS_PER_LANE && opts.tlas_instances && p.items[size_t(h)].skip == h + 2) {
                const Item& body = p.items[size_t(h) + 1];
                instance = body.type == ITEM_MESH ||
                           (body.type == ITEM_LEAF && (body.kind == RT_SPHERE || body.kind == RT_CUBE || body.kind == RT_CYLINDER));
            }
            return instance ? ~((h + 1) | TLAS_INSTANCE) : ~h;
        }
        int axis = 0;
        double ext = -1.;
        for (int k = 0; k < 3; ++k) {
            double cl = INF, ch = -INF;
            for (size_t i = lo; i < hi; ++i) {
                const double c = 0.5 * roots[i].second.lo[k] + 0.5 * roots[i].second.hi[k];
                cl = std::min(cl, c);
                ch = std::max(ch, c);
            }
            if (ch - cl > ext) { ext = ch - cl; axis = k; }
        }
        // (roots and header are permuted together: sort an index, then apply)
        const size_t mid = lo + (hi - lo) / 2;
        std::vector<size_t> idx(hi - lo);
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = lo + i;
        std::nth_element(idx.begin(), idx.begin() + long(mid - lo), idx.end(), [&](size_t a, size_t b) {
            const double ca = roots[a].second.lo[axis] + roots[a].second.hi[axis], cb = roots[b].second.lo[axis] + roots[b].second.hi[axis];
            return ca < cb || (ca == cb && a < b);
        });
        {
            std::vector<std::pair<int32_t, WBox>> r2(idx.size());
            std::vector<int32_t> h2(idx.size());
            for (size_t i = 0; i < idx.size(); ++i) { r2[i] = roots[idx[i]]; h2[i] = tlas_header[idx[i]]; }
            for (size_t i = 0; i < idx.size(); ++i) { roots[lo + i] = r2[i]; tlas_header[lo + i] = h2[i]; }
        }
        const int32_t me = int32_t(n2.size());
        n2.emplace_back();
        Box b0, b1;
        const int32_t c0 = tlas_build(roots, lo, mid, b0);
        const int32_t c1 = tlas_build(roots, mid, hi, b1);
        Node2& nd = n2[size_t(me)];
        nd.box[0] = b0; nd.box[1] = b1;
        nd.child[0] = c0; nd.child[1] = c1;
        return me;
    }
    std::vector<int32_t> tlas_header;  // per bounded object (same order as `roots`): index of its ITEM_CULL
    void emit_tlas(std::vector<std::pair<int32_t, WBox>>& roots) {
        p.n_linear = int32_t(p.items.size());
        tlas_header.assign(roots.size(), -1);
        for (size_t i = 0; i < roots.size(); ++i) {
            const size_t before = p.items.size();
            tiny_mesh_root = RT_TLAS_PER_LANE && opts.tlas_instances ? roots[i].first : -1;
            emit_culled(roots, i, i + 1, true);   // ITEM_CULL(own box) + the object + skip
            tlas_header[i] = int32_t(before);
        }
        tiny_mesh_root = -1;
        n2.clear();
        Box rb;
        const int32_t root2 = tlas_build(roots, 0, roots.size(), rb);
        depth_bias = -(1 << 20);
        p.tlas_root = collapse(root2, 0);
        depth_bias = 0;
    }

    // ---- traversal program -----------------------------------------------------------------------------
    void set_chain(Item& it, const std::vector<int32_t>& chain) {
        it.chain = chain_offset(chain);
        it.chain_len = int32_t(chain.size());
        it.x0 = chain.size() > 0 ? chain[0] : 0;
        it.x1 = chain.size() > 1 ? chain[1] : 0;
    }

    // Under a top-level tree (emit_tlas) a bounded object that is nothing but triangles — under one group or under several
    // nested ones that change nothing (no transform of their own, no gate: triangle groups have infinite boxes, F4), however
    // few: triangles.obj is two groups of one triangle each — becomes ONE ITEM_MESH, so that its tree leaf is an instance and
    // every lane meets the one ITS ray reached, instead of the wavefront visiting such objects one after the other.  (Keys are
    // node * 4 + push whoever tests the triangle, and normals use the per-leaf registry chains: results cannot change.)
    int32_t tiny_mesh_root = -1;  // the top-level object being emitted under emit_tlas (the rule applies to the object itself)
    bool only_plain_triangles(int32_t group, std::vector<int32_t>& tris) const {
        for (int32_t c = group + 1; c < d.node_subtree_end[group]; c = d.node_subtree_end[c]) {
            if (d.node_xform[c] != 0) return false;
            if (is_tri_kind(d.node_kind[c])) { tris.push_back(c); continue; }
            if (d.node_kind[c] != RT_GROUP) return false;
            const double* box = d.group_box + size_t(d.node_param[c]) * 6;
            bool infinite = true, empty = true;
            for (int k = 0; k < 3; ++k) {
                infinite = infinite && box[k] == -INF && box[3 + k] == INF;
                empty = empty && box[k] == INF && box[3 + k] == -INF;
            }
            if (!infinite && !empty) return false;
            if (!only_plain_triangles(c, tris)) return false;
        }
        return true;
    }

    void emit(int32_t node, std::vector<int32_t> chain) {
        int32_t x = d.node_xform[node];
        if (x != 0) chain.push_back(x);
        int32_t kind = d.node_kind[node];
        if (is_leaf_kind(kind)) {
            Item it{};
            it.type = ITEM_LEAF;
            it.node = node;
            it.kind = kind;
            set_chain(it, chain);
            if (kind == RT_CYLINDER || kind == RT_CONE) {
                int32_t q = d.node_param[node];
                it.p0 = d.conic_minimum[q];
                it.p1 = d.conic_maximum[q];
                it.op = d.conic_closed[q] ? 1 : 0;
            } else if (is_tri_kind(kind)) {
                it.a = d.node_param[node];
            }
            p.items.push_back(it);
            return;
        }
        if (kind == RT_GROUP) {
            const double* box = d.group_box + size_t(d.node_param[node]) * 6;
            bool infinite = true, empty = true;
            for (int k = 0; k < 3; ++k) {
                infinite = infinite && box[k] == -INF && box[3 + k] == INF;
                empty = empty && box[k] == INF && box[3 + k] == -INF;
            }
            int32_t gate = -1;
            if (!infinite && !empty) {  // both of those pass for every ray ("Group gate")
                Item it{};
                it.type = ITEM_GATE;
                it.node = node;
                set_chain(it, chain);
                it.a = d.node_param[node];
                gate = int32_t(p.items.size());
                p.items.push_back(it);
            }
            std::vector<int32_t> tris;
            if (node == tiny_mesh_root && gate < 0 && only_plain_triangles(node, tris) && !tris.empty()) {
                Item it{};
                it.type = ITEM_MESH;
                it.node = node;
                set_chain(it, chain);
                it.a = build_mesh(tris, node);
                it.b = int32_t(p.mesh_box.size() / 6);
                for (int k = 0; k < 3; ++k) p.mesh_box.push_back(last_mesh_box.lo[k]);
                for (int k = 0; k < 3; ++k) p.mesh_box.push_back(last_mesh_box.hi[k]);
                mesh_items.push_back(p.items.size());
                p.items.push_back(it);
                return;
            }
            tris.clear();
            for (int32_t c = node + 1; c < d.node_subtree_end[node]; c = d.node_subtree_end[c])
                if (is_tri_kind(d.node_kind[c]) && d.node_xform[c] == 0) tris.push_back(c);
            bool mesh = tris.size() >= size_t(MESH_MIN_TRIANGLES);
            if (mesh) {
                Item it{};
                it.type = ITEM_MESH;
                it.node = node;
                set_chain(it, chain);
                it.a = build_mesh(tris, node);
                it.b = int32_t(p.mesh_box.size() / 6);
                for (int k = 0; k < 3; ++k) p.mesh_box.push_back(last_mesh_box.lo[k]);
                for (int k = 0; k < 3; ++k) p.mesh_box.push_back(last_mesh_box.hi[k]);
                mesh_items.push_back(p.items.size());
                p.items.push_back(it);
            }
            for (int32_t c = node + 1; c < d.node_subtree_end[node]; c = d.node_subtree_end[c]) {
                if (mesh && is_tri_kind(d.node_kind[c]) && d.node_xform[c] == 0) continue;
                emit(c, chain);
            }
            if (gate >= 0) p.items[size_t(gate)].skip = int32_t(p.items.size());
            return;
        }
        // CSG
        int32_t left = node + 1, right = d.node_subtree_end[left];
        Item b{};
        b.type = ITEM_CSG_BEGIN;
        b.node = node;
        p.items.push_back(b);
        emit(left, chain);
        emit(right, chain);
        Item e{};
        e.type = ITEM_CSG_END;
        e.node = node;
        e.a = left;
        e.b = d.node_subtree_end[left];
        e.op = d.csg_op[d.node_param[node]];
        {
            // Shape::includes compares object ids (src/shape.rs:422-435), the device compares node ranges: the same thing
            // unless one id sits in both operands (csg(op, s, s.clone()) keeps the id).  The right operand's leaves
            // whose id also occurs in the left operand are listed (ascending) behind the item; normally there are none.
            const int32_t end = d.node_subtree_end[node];
            std::unordered_set<int32_t> left_ids;
            for (int32_t i = left; i < right; ++i)
                if (is_leaf_kind(d.node_kind[i])) left_ids.insert(d.node_object_id[i]);
            std::vector<int32_t> shared;
            for (int32_t i = right; i < end; ++i)
                if (is_leaf_kind(d.node_kind[i]) && left_ids.count(d.node_object_id[i])) shared.push_back(i);
            if (!shared.empty()) {
                e.chain = int32_t(p.chain_pool.size());
                e.chain_len = int32_t(shared.size());
                p.chain_pool.insert(p.chain_pool.end(), shared.begin(), shared.end());
            }
        }
        p.items.push_back(e);
        p.has_csg = true;
    }
};

bool affine_row(const double* m16) {
    return m16[12] == 0. && m16[13] == 0. && m16[14] == 0. && m16[15] == 1.;
}

#define FAIL(code, ...)                                  \
    do {                                                 \
        char buf_[512];                                  \
        std::snprintf(buf_, sizeof buf_, __VA_ARGS__);   \
        err = buf_;                                      \
        return code;                                     \
    } while (0)

int validate(const rt_scene_desc& d, std::string& err, bool& general) {
    general = false;
    if (d.abi_version != RT_ABI_VERSION) FAIL(RT_ERR_BAD_ARG, "abi_version %d, library speaks %d", d.abi_version, RT_ABI_VERSION);
    if (d.n_nodes < 0 || d.n_xforms < 1 || d.n_ancestors < 0 || d.n_groups < 0 || d.n_csg < 0 || d.n_conics < 0 ||
        d.n_triangles < 0 || d.n_smooth < 0 || d.n_materials < 0 || d.n_patterns < 0 || d.n_lights < 0)
        FAIL(RT_ERR_BAD_ARG, "negative table size (n_xforms must be >= 1: xform 0 is the identity)");
    if (d.n_nodes > (1 << 28)) FAIL(RT_ERR_UNSUPPORTED, "more than 2^28 nodes");
    if (d.n_nodes && (!d.node_kind || !d.node_object_id || !d.node_parent || !d.node_subtree_end || !d.node_xform ||
                      !d.node_ancestor || !d.node_material || !d.node_casts_shadow || !d.node_eq_class || !d.node_param))
        FAIL(RT_ERR_BAD_ARG, "null node table");
    if (!d.xform_inverse || !d.xform_inverse_transpose) FAIL(RT_ERR_BAD_ARG, "null xform table");
    if ((d.n_ancestors && (!d.ancestor_xform || !d.ancestor_parent)) || (d.n_groups && !d.group_box) ||
        (d.n_csg && !d.csg_op) || (d.n_conics && (!d.conic_minimum || !d.conic_maximum || !d.conic_closed)) ||
        (d.n_triangles && (!d.tri_p1 || !d.tri_e1 || !d.tri_e2 || !d.tri_normal || !d.tri_smooth)) ||
        (d.n_smooth && (!d.smooth_n1 || !d.smooth_n2 || !d.smooth_n3)) ||
        (d.n_materials && (!d.material_params || !d.material_pattern)) ||
        (d.n_patterns && (!d.pattern_kind || !d.pattern_a || !d.pattern_b || !d.pattern_inverse)) ||
        (d.n_lights && (!d.light_position || !d.light_intensity)))
        FAIL(RT_ERR_BAD_ARG, "null table with a non-zero count");
    for (int i = 0; i < 16; ++i)
        if (d.xform_inverse[i] != ((i % 5 == 0) ? 1. : 0.)) FAIL(RT_ERR_BAD_ARG, "xform 0 must be the exact identity");
    // An inverse whose last row is not 0,0,0,1 (only Matrix::new with hand-made values gets there: every product of
    // src/transformation.rs is affine, and so is its cofactor inverse) sends points out of Matrix * Tuple with w != 1: such
    // a scene runs on the 4-component path
    for (int32_t x = 0; x < d.n_xforms; ++x)
        if (!affine_row(d.xform_inverse + size_t(x) * 16)) general = true;
    for (int32_t q = 0; q < d.n_patterns; ++q) {
        if (!affine_row(d.pattern_inverse + size_t(q) * 16)) general = true;
        if (d.pattern_kind[q] < RT_PAT_STRIPED || d.pattern_kind[q] > RT_PAT_TEST) FAIL(RT_ERR_BAD_ARG, "pattern %d: bad kind", q);
    }
    for (int32_t a = 0; a < d.n_ancestors; ++a) {
        if (d.ancestor_xform[a] < 0 || d.ancestor_xform[a] >= d.n_xforms) FAIL(RT_ERR_BAD_ARG, "ancestor %d: xform out of range", a);
        int32_t up = d.ancestor_parent[a];
        if (up < -1 || up >= d.n_ancestors) FAIL(RT_ERR_BAD_ARG, "ancestor %d: parent out of range", a);
    }
    {   // no cycles (every entry range-checked above); chains may be as deep as the scene graph is (src/shape.rs:394-420 recurses
        // without bound): a group divide()d into a tall tree gets there.  state: 0 unseen, 1 on the current walk, 2 known to end
        std::vector<unsigned char> state(size_t(d.n_ancestors), 0);
        std::vector<int32_t> walk;
        for (int32_t a = 0; a < d.n_ancestors; ++a) {
            walk.clear();
            int32_t c = a;
            while (c >= 0 && state[size_t(c)] == 0) { state[size_t(c)] = 1; walk.push_back(c); c = d.ancestor_parent[c]; }
            if (c >= 0 && state[size_t(c)] == 1) FAIL(RT_ERR_BAD_ARG, "ancestor %d: parent chain is cyclic", a);
            for (int32_t v : walk) state[size_t(v)] = 2;
        }
    }
    for (int32_t m = 0; m < d.n_materials; ++m)
        if (d.material_pattern[m] < -1 || d.material_pattern[m] >= d.n_patterns) FAIL(RT_ERR_BAD_ARG, "material %d: pattern out of range", m);
    for (int32_t t = 0; t < d.n_triangles; ++t)
        if (d.tri_smooth[t] < -1 || d.tri_smooth[t] >= d.n_smooth) FAIL(RT_ERR_BAD_ARG, "triangle %d: smooth index out of range", t);
    for (int32_t n = 0; n < d.n_nodes; ++n) {
        int32_t k = d.node_kind[n];
        if (k < RT_SPHERE || k > RT_CSG) FAIL(RT_ERR_BAD_ARG, "node %d: bad kind %d", n, k);
        if (d.node_xform[n] < 0 || d.node_xform[n] >= d.n_xforms) FAIL(RT_ERR_BAD_ARG, "node %d: xform out of range", n);
        int32_t end = d.node_subtree_end[n];
        if (end <= n || end > d.n_nodes) FAIL(RT_ERR_BAD_ARG, "node %d: bad subtree_end", n);
        int32_t par = d.node_parent[n];
        if (par < -1 || par >= n) FAIL(RT_ERR_BAD_ARG, "node %d: bad parent", n);
        if (par >= 0 && end > d.node_subtree_end[par]) FAIL(RT_ERR_BAD_ARG, "node %d: subtree exceeds its parent's", n);
        int32_t prm = d.node_param[n];
        bool leaf = k <= RT_SMOOTH_TRIANGLE;
        if (leaf) {
            if (end != n + 1) FAIL(RT_ERR_BAD_ARG, "node %d: a primitive cannot have children", n);
            if (d.node_material[n] < 0 || d.node_material[n] >= d.n_materials) FAIL(RT_ERR_BAD_ARG, "node %d: material out of range", n);
            if (d.node_ancestor[n] < -1 || d.node_ancestor[n] >= d.n_ancestors) FAIL(RT_ERR_BAD_ARG, "node %d: ancestor out of range", n);
        }
        if ((k == RT_CYLINDER || k == RT_CONE) && (prm < 0 || prm >= d.n_conics)) FAIL(RT_ERR_BAD_ARG, "node %d: conic index out of range", n);
        if ((k == RT_TRIANGLE || k == RT_SMOOTH_TRIANGLE)) {
            if (prm < 0 || prm >= d.n_triangles) FAIL(RT_ERR_BAD_ARG, "node %d: triangle index out of range", n);
            if ((k == RT_SMOOTH_TRIANGLE) != (d.tri_smooth[prm] >= 0)) FAIL(RT_ERR_BAD_ARG, "node %d: kind and tri_smooth disagree", n);
        }
        if (k == RT_GROUP && (prm < 0 || prm >= d.n_groups)) FAIL(RT_ERR_BAD_ARG, "node %d: group index out of range", n);
        if (k == RT_CSG) {
            if (prm < 0 || prm >= d.n_csg) FAIL(RT_ERR_BAD_ARG, "node %d: csg index out of range", n);
            int32_t left = n + 1;
            if (left >= end) FAIL(RT_ERR_BAD_ARG, "node %d: CSG without operands", n);
            int32_t right = d.node_subtree_end[left];
            if (right <= left || right >= end || d.node_subtree_end[right] != end) FAIL(RT_ERR_BAD_ARG, "node %d: CSG needs exactly two operands", n);
        }
    }
    return RT_OK;
}

}  // namespace

int compile_program(const rt_scene_desc& d, Program& out, std::string& err, const ProgramOpts& opts) {
    const double t_begin = Compiler::now();
    int rc = validate(d, err, out.general);
    if (rc != RT_OK) return rc;
    if (out.general) {  // nothing to compile: the 4-component path walks the descriptor's own tables
        out.chain_pool.assign(1, 0);
        out.node_nchain.assign(size_t(d.n_nodes), 0);
        return RT_OK;
    }
    const double t_validated = Compiler::now();
    Compiler c(d, out);
    c.opts = opts;
    // chain 0 = the empty chain
    c.chain_offset({});
    out.node_nchain.assign(size_t(d.n_nodes), -1);
    // Emission order of the top-level objects is free (results are ordered by (t, node key), not by visit order): the
    // ones without a triangle mesh go first, so that by the time a ray reaches a mesh its closest-hit bound — which
    // prunes the mesh box and the BVH — already knows about the cheap primitives.
    auto has_mesh = [&](int32_t root) {
        for (int32_t g = root; g < d.node_subtree_end[root]; ++g) {
            if (d.node_kind[g] != RT_GROUP) continue;
            size_t tris = 0;
            for (int32_t ch = g + 1; ch < d.node_subtree_end[g]; ch = d.node_subtree_end[ch])
                if ((d.node_kind[ch] == RT_TRIANGLE || d.node_kind[ch] == RT_SMOOTH_TRIANGLE) && d.node_xform[ch] == 0) ++tris;
            if (tris >= size_t(MESH_MIN_TRIANGLES)) return true;
        }
        return false;
    };
    // Top-level objects with bounded extent additionally go under ITEM_CULL ranges (world-space boxes: a ray that
    // misses a box, or meets it beyond what its sink still accepts, skips everything under it); the unbounded ones
    // (planes, open cylinders and cones ...) come first, so that their hits already shorten that range.
    std::vector<std::pair<int32_t, Compiler::WBox>> bounded;
    for (int pass = 0; pass < 2; ++pass)
        for (int32_t n = 0; n < d.n_nodes; n = d.node_subtree_end[n]) {
            if (has_mesh(n) != (pass == 1)) continue;
            Compiler::WBox wb = c.world_box(n, {});
            if (wb.bounded && wb.lo[0] <= wb.hi[0]) bounded.emplace_back(n, wb);
            else c.emit(n, {});
        }
    // From TLAS_MIN_OBJECTS bounded objects on, the tree; from TLAS_MIN_INSTANCES on already when EVERY bounded object is one a
    // lane handles by itself (a lone sphere / cube / cylinder, or a group that is nothing but one mesh) and the scene has no
    // CSG (the tree kernels with CSG code are the heaviest variants).  Measured, ranges / tree, ms per frame: 32 teapot
    // instances scattered at random, 960x540: 3.57 / 1.88 (64: 5.58 / 2.73); 32 spheres: 0.81 / 0.68 (96: 1.59 / 1.03); but a
    // (16: 2.45 / 1.44; 16 spheres: 0.55 / 0.45; 12 spheres: 0.45 / 0.48); but a
    // regular grid of 16 teapots at 8K 13.1 / 14.3, of 36: 15.5 / 15.6; C5 at grid 4 (16 teapots + 16 two-triangle groups,
    // which the wavefront visits in turn) 14.1 / 15.3; C4 (9 objects, CSG and a hexagon among them) 5.4 / 8.3.
    size_t tlas_from = size_t(TLAS_MIN_OBJECTS);
    if (bounded.size() >= size_t(TLAS_MIN_INSTANCES) && bounded.size() < tlas_from) {
        bool all_by_lane = true;
        for (int32_t n = 0; n < d.n_nodes && all_by_lane; ++n) all_by_lane = d.node_kind[n] != RT_CSG;
        for (const auto& b : bounded) {
            if (!all_by_lane) break;
            const int32_t n = b.first, k = d.node_kind[n];
            if (k == RT_SPHERE || k == RT_CUBE || k == RT_CYLINDER) continue;
            // a group without a gate (infinite or empty box: see emit) that holds nothing but triangles — directly, or inside
            // one more such group, the way obj_to_group wraps an OBJ's default group — is emitted as one ITEM_MESH
            bool one_mesh = false;
            for (int32_t g = n; d.node_kind[g] == RT_GROUP;) {
                const double* box = d.group_box + size_t(d.node_param[g]) * 6;
                bool infinite = true, empty = true;
                for (int a = 0; a < 3; ++a) {
                    infinite = infinite && box[a] == -INF && box[3 + a] == INF;
                    empty = empty && box[a] == INF && box[3 + a] == -INF;
                }
                if (!infinite && !empty) break;
                const int32_t first = g + 1, end = d.node_subtree_end[g];
                if (first >= end) break;
                if (d.node_subtree_end[first] == end && d.node_kind[first] == RT_GROUP) { g = first; continue; }
                size_t tris = 0;
                bool all_tris = true;
                for (int32_t ch = first; ch < end && all_tris; ch = d.node_subtree_end[ch]) {
                    all_tris = Compiler::is_tri_kind(d.node_kind[ch]) && d.node_xform[ch] == 0;
                    ++tris;
                }
                one_mesh = all_tris && tris >= size_t(MESH_MIN_TRIANGLES);
                break;
            }
            all_by_lane = one_mesh;
        }
        if (all_by_lane) tlas_from = size_t(TLAS_MIN_INSTANCES);
    }
    if (opts.tlas_min_objects > 0) tlas_from = size_t(opts.tlas_min_objects);
    if (opts.cull_ranges && opts.top_level_tree && bounded.size() >= tlas_from) c.emit_tlas(bounded);
    else c.emit_culled(bounded, 0, bounded.size());
    if (out.tlas_root < 0) out.n_linear = int32_t(out.items.size());
    if (out.tlas_root < 0) {  // the n1/n2 walk may bound its mesh searches once every non-mesh leaf has been listed
        int32_t first = -1;
        bool only_meshes_behind = true;
        for (size_t i = 0; i < out.items.size(); ++i) {
            if (out.items[i].type == ITEM_MESH) { if (first < 0) first = int32_t(i); }
            else if (first >= 0) only_meshes_behind = false;
        }
        // (the walk is cut in two there: no group gate or cull range may reach across the cut)
        for (int32_t i = 0; i < first && only_meshes_behind; ++i)
            if ((out.items[size_t(i)].type == ITEM_GATE || out.items[size_t(i)].type == ITEM_CULL) && out.items[size_t(i)].skip > first)
                only_meshes_behind = false;
        out.walk_bound_item = only_meshes_behind ? first : -1;
    }
    for (int32_t n = 0; n < d.n_nodes; ++n) {
        int32_t k = d.node_kind[n];
        if (k <= RT_SMOOTH_TRIANGLE) {
            out.node_nchain[size_t(n)] = c.nchain_offset(n);
            if (k == RT_SMOOTH_TRIANGLE) out.has_smooth = true;
            if (!d.node_casts_shadow[n]) out.all_cast_shadow = false;
        }
    }
    const double t_emitted = Compiler::now();
    c.finish_bvh();
    if (opts.verbose)
        std::fprintf(stderr, "compile_program: validate %.3f s, program + meshes %.3f s (of which de-duplication %.3f, triangle bounds %.3f, trees %.3f), "
                             "depth ordering %.3f s; %d meshes built on the GPU, %d meshes took the tree of an identical earlier one, deepest tree %d levels\n",
                     t_validated - t_begin, t_emitted - t_validated, c.t_dedup, c.t_refs, c.t_tree, Compiler::now() - t_emitted,
                     out.meshes_built_on_gpu, out.meshes_shared, out.bvh_max_depth);
    if (3 * (out.bvh_max_depth + out.tlas_levels) + 1 > BVH_STACK) {
        err = "BVH deeper than the device traversal stack";
        return RT_ERR_UNSUPPORTED;
    }
    return RT_OK;
}

}  // namespace rt
