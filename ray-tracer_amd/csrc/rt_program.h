// rt_program.h — what rt_scene_create compiles the flattened scene (rt_abi.h) into before upload:
// a linear traversal program over the reference's scene graph plus a conservative BVH per triangle mesh.
//
// Nothing here changes results.  The reference's own Group box gate (src/group.rs:52-53, src/bound.rs:85-116) is
// semantic and is replayed literally as ITEM_GATE; the BVH is an acceleration over real triangle bounds that the
// reference does not have (its triangle groups have infinite boxes: src/triangle.rs:93-95 + src/bound.rs:35-47)
// and is built so that it never culls an intersection the reference arithmetic would report.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace rt {

enum ItemType : int32_t {
    ITEM_LEAF = 0,       // one primitive: node = leaf node index
    ITEM_GATE = 1,       // reference Group gate: node = group node, a = group box index, skip = item after the subtree
    ITEM_MESH = 2,       // BVH over the triangle children of one group: a = BVH root node index
    ITEM_CSG_BEGIN = 3,  // open a CSG scratch segment: node = csg node
    ITEM_CSG_END = 4,    // sort + filter the segment: node = csg node, a..b = node range of the LEFT operand, op = csg op;
                         // chain_len > 0: chain_pool[chain .. chain + chain_len) lists (ascending) the RIGHT operand's leaves
                         // whose object id also occurs in the left operand — Shape::includes is by id (src/shape.rs:422-435)
    ITEM_CULL = 5        // acceleration only: padded WORLD-space bounds of everything up to item `skip` (a set of top-level
                         // objects); a ray that misses them, or meets them beyond what its sink still accepts, skips there.
                         // The six bounds overlay bytes 16..63 of the record (cull_box()).
};

// One 64-byte record per item: everything the traversal needs for the item arrives with ONE scalar load (the
// item index is wavefront-uniform, see traverse() in rt_kernels.hip.h), and the first two transforms of the chain
// are named inline so that the common chains need no second dependent fetch.
struct Item {
    int32_t type;
    int32_t node;
    int32_t skip;       // ITEM_GATE: index of the item after the group's subtree
    int32_t kind;       // ITEM_LEAF: RT_* kind of the node
    int32_t a;          // GATE: group box index; MESH: BVH root; CSG_END: first node of the left operand;
                        // LEAF triangle: triangle index
    int32_t b;          // CSG_END: end of the left operand's node range;  MESH: index into mesh_box
    int32_t op;         // CSG_END: csg op;  LEAF cylinder / cone: closed
    int32_t chain_len;  // non-identity inverses to apply to the WORLD ray, outermost first
    int32_t x0, x1;     // the first two of them
    int32_t chain;      // offset into chain_pool of [len, xform...] (read only when chain_len > 2)
    int32_t _pad;
    double p0, p1;      // LEAF cylinder / cone: minimum, maximum
};
static_assert(sizeof(Item) == 64, "Item must be 64 bytes");
inline double* cull_box(Item& it) { return reinterpret_cast<double*>(reinterpret_cast<char*>(&it) + 16); }  // lo xyz, hi xyz

// BVH4: the four children's boxes are stored in the parent (one 128-byte fetch decides all four), as f32 rounded
// OUTWARD from the padded f64 boxes — the device converts back to f64 (exact) and tests in f64, so a stored box
// always contains the f64 box it came from.  Nodes of all meshes are ordered by depth, so that any prefix of the
// array is "the top levels" — that prefix is what the traversal kernels stage in LDS.
struct BvhNode {
    float lo[3][4];    // [axis][child]
    float hi[3][4];
    int32_t child[4];  // >= 0: inner node index;  < 0: leaf, ~c = (first_tri << 2) | (count - 1);  INT32_MIN: empty
    int32_t _pad[4];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode must be 128 bytes");

// Triangle copy in BVH leaf order (what the traversal kernel streams).
struct MeshTri {
    double p1[3], e1[3], e2[3];  // src/triangle.rs:17-21
    int32_t node;                // leaf node index MINUS the group node of its mesh (ITEM_MESH.node): identical meshes
                                 // share one tree; the traversal adds the base back (emission order + shading look-ups)
    int32_t smooth;              // bit 0: SmoothTriangle (keeps u,v);  bits 1..: 0, or 1 + offset into mesh_dups of
                                 // [count, node, ...] — further nodes with the bit-identical triangle (tested once)
};
static_assert(sizeof(MeshTri) == 80, "MeshTri must be 80 bytes");

struct Program {
    std::vector<Item> items;
    std::vector<int32_t> chain_pool;
    std::vector<int32_t> node_nchain;  // per leaf node: offset into chain_pool of [len, x_leaf, x_parent, ..., x_root]
                                       // (registry chain of src/shape.rs:394-420, identities kept)
    std::vector<BvhNode> bvh;
    std::vector<MeshTri> mesh_tris;
    std::vector<int32_t> mesh_dups;    // node lists of repeated triangles (MeshTri::smooth), relative like MeshTri::node
    std::vector<int32_t> mesh_nodes;   // every triangle node that lives in some mesh (all instances, repeats included)
    int32_t meshes_shared = 0;         // ITEM_MESH items that took the tree of an earlier, identical mesh
    std::vector<double> mesh_box;      // 6 per ITEM_MESH: padded f64 bounds of the whole mesh
    bool has_csg = false;
    bool general = false;              // some inverse is not affine: no program is compiled, the scene runs on the 4-component
                                       // path (rt_general.hip.h) and everything above stays empty
    int32_t meshes_built_on_gpu = 0;   // meshes whose tree topology came from rt_lbvh.hip
    // Worlds of many bounded top-level objects (>= TLAS_MIN_OBJECTS): instead of nested ITEM_CULL ranges walked in program
    // order, every bounded object is a range of its own — [ITEM_CULL with the object's world box, the object's items) —
    // behind the first `n_linear` items (the unbounded objects, walked by every ray as before), and a BVH4 over the
    // objects' world boxes (`tlas_root`, nodes in front of all mesh nodes so that they are always staged in LDS; a
    // leaf link is ~(index of the object's ITEM_CULL)) tells each ray which ranges to visit, nearest first.
    int32_t tlas_root = -1;            // -1: no top-level tree, the whole program is walked linearly
    int32_t n_linear = 0;              // items [0, n_linear) are walked by every ray (== items.size() without a tree)
    int32_t tlas_levels = 0;           // BVH4 levels of the top-level tree (its share of the traversal stack)
    bool has_cull = false;             // the program has ITEM_CULL ranges
    double cull_reach = 0.;            // ... trusted for rays whose origin has max |coordinate| below this (WBox::reach)
    // Rays from beyond the reach, under a top-level tree: every box is grown by (far_a D + far_b) D, D = the L-infinity distance
    // the origin can have from anything bounded (max |origin coordinate| + far_extent).  far_a bounds the quadratics' leak
    // (2 eps R^2 object units from R radii away), far_b the linear ones (a cube's EPSILON |t|, the t error of a near-tangent
    // root, Moller-Trumbore's u, v); both carry a safety factor of 8 (rt_program.cpp: world_box).
    double far_a = 0., far_b = 0., far_extent = 0.;
    double cull_min_dir2 = 0.;         // ... and whose direction is at least this long, squared (0.25 once a cube is bounded: its
                                       // padding covers the reference's leak only up to a parameter |t| of CUBE_PAD / EPSILON)
    int32_t walk_bound_item = -1;      // index of the first ITEM_MESH when nothing but ITEM_MESH follows it and there is no top-level
                                       // tree (every other leaf is listed by then: WalkSink::before_mesh), else -1
    bool has_smooth = false;
    bool all_cast_shadow = true;
    int32_t max_chain = 0;
    int32_t bvh_max_depth = 0;
};

// Validates the descriptor (pointers, index ranges, affine inverses, CSG operand ids) and compiles it.
// Returns RT_OK or an RT_ERR_* code with `err` set.
// `opts.build_device` >= 0 and `opts.device_bvh`: meshes of at least LBVH_MIN_TRIANGLES distinct triangles are built on
// that GPU (rt_lbvh.hip); everything else, and any mesh the GPU build declines, uses the host's binned-SAH build.
struct ProgramOpts {
    int build_device = -1;
    bool device_bvh = false;
    bool cull_ranges = true;
    bool tlas_instances = true;  // ... whose one-mesh objects are walked per lane (RT_AMD_TLAS=2: left to the wavefront, A/B)
    int tlas_min_objects = 0;    // > 0: build the tree from this many bounded objects on instead of the library's rule (RT_AMD_TLAS_MIN: A/B)
    bool top_level_tree = true;  // worlds with >= TLAS_MIN_OBJECTS bounded top-level objects get a tree over them (RT_AMD_TLAS=0: A/B)
    bool share_meshes = true;   // groups with identical triangle content take one tree (RT_AMD_SHARE_MESHES=0: A/B)
    bool verbose = false;   // phase times of the compilation on stderr
};
int compile_program(const rt_scene_desc& d, Program& out, std::string& err, const ProgramOpts& opts = ProgramOpts());

// rt_lbvh.hip: the whole BVH of one mesh on the GPU.  tri_lo / tri_hi [n][3]: unpadded bounds of the distinct
// triangles; tri_base: index the mesh's first MeshTri will have (leaves are encoded with it).  On success: order[p] =
// triangle at leaf-order position p (the caller emits its MeshTri records in that order), `nodes` = the BVH4 nodes in
// breadth-first order with mesh-local child indices (node 0 = root), level_start[k] .. level_start[k + 1] = the nodes of
// depth k, root_box = padded f64 bounds of the mesh.  false: declined (too small, too deep for `max_levels`, not
// finite, out of memory, HIP error) — the caller builds on the host.
struct BvhNode;
bool lbvh_build(int device, size_t n, const double* tri_lo, const double* tri_hi, double pad, int32_t tri_base, int max_levels,
                std::vector<int32_t>& order, std::vector<BvhNode>& nodes, std::vector<int32_t>& level_start, double root_box[6]);

// Per-lane walks of mesh instances under the top-level tree: compiled in only on request (-DRT_TLAS_PER_LANE=1).  Measured
// (DESIGN.md section 7, round 3): the second inlined copy of the mesh traversal takes the SC_TLAS kernels from 43 to 192
// spilled registers, which costs C5 more (grid 4: 15.2 -> 19.6 ms) than lanes working side by side on different teapots
// gain (grid 13: 27.4 -> 23.9 ms inside one binary, against 23.6 ms for the binary without the code).
#ifndef RT_TLAS_PER_LANE
#define RT_TLAS_PER_LANE 1
#endif
constexpr int32_t TLAS_INSTANCE = 1 << 29;  // in a top-level leaf link ~c: c & ~TLAS_INSTANCE is the index of an ITEM_MESH (a mesh
                                            // instance, walked per lane) rather than of an object's ITEM_CULL (walked by the wavefront)
constexpr int TLAS_MIN_OBJECTS = 64;     // fewer bounded top-level objects: nested ITEM_CULL ranges in program order (measured on C5
                                         // at 8K, ranges / tree: 32 objects 14.3 / 15.9 ms, 72: 17.1 / 17.0, 128: 19.0 / 18.0, 338: 24.8 / 20.5)
constexpr int TLAS_MIN_INSTANCES = 16;   // ... but already from this many when every one of them is walked per lane (rt_program.cpp: compile_program)
constexpr int LBVH_MIN_TRIANGLES = 1024; // smaller meshes are always built on the host
constexpr int MESH_MIN_TRIANGLES = 8;  // smaller triangle sets are tested one by one
constexpr int BVH_LEAF_MAX = 4;        // triangles per leaf (measured: 1 -> 2.74 ms, 2 -> 2.21 ms, 4 -> 2.12 ms per frame); fetched two at a time
constexpr int BVH_STACK = 64;          // most device traversal stack entries per ray a scene may need: 3 pending siblings per BVH4 level
                                       // (the spill area is sized by what the scene's trees do need: host-built trees stay within 31)
constexpr int BVH2_MAX_DEPTH = 20;     // binary build depth bound; the collapse halves it: 3 * 10 <= BVH_STACK

}  // namespace rt
