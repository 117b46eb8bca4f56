// rt_device.h — device-side tables and per-level queues shared by the kernels (rt_kernels.hip) and the
// orchestration (rt_api.hip).  Plain structs of device pointers, passed to kernels by value.
#pragma once
#include <cstdint>

#include "rt_program.h"

namespace rt {

constexpr double EPS = 0.00001;   // EPSILON, src/lib.rs:42
constexpr int BLOCK = 256;        // 4 wavefronts: streaming kernels
#ifndef RT_TBLOCK
#define RT_TBLOCK 1024
#endif
#ifndef RT_TWAVES
#define RT_TWAVES 4   // wavefronts per SIMD the traversal kernels are held to (512 / RT_TWAVES VGPRs each)
#endif
#ifndef RT_LDS_NODE_CAP
#define RT_LDS_NODE_CAP 895
#endif
constexpr int TBLOCK = RT_TBLOCK;      // 16 wavefronts: traversal kernels (they share one LDS copy of the BVH top levels)
constexpr int LDS_STACK = 12;     // traversal stack entries per lane kept in LDS; deeper entries spill to HBM (rare)
constexpr int LDS_NODE_CAP = RT_LDS_NODE_CAP; // BVH4 nodes staged in LDS: ~112 KiB + 48 KiB of per-lane stacks + the feed word <= 160 KiB
constexpr int FEED_PARTS = 8;     // chunk-feed counters per traversal kernel, one cache line each
constexpr int FEED_STRIDE = 16;   // in counters (128 bytes)
constexpr int FEED_ROWS = 64;     // consecutive feed positions are 1/FEED_ROWS of the level apart (see Feed::next)
constexpr int CSG_CAP = 32;       // per-ray scratch entries while inside CSG subtrees
constexpr int CSG_DEPTH = 128;    // nested CSG levels held per lane (deeper trees: the arena variants)
// (the overflow (BIG) kernel variants keep their CSG marks in the HBM arena: as many nested levels as pending entries)
#ifndef RT_LIST_CAP
#define RT_LIST_CAP 256
#endif
constexpr int LIST_CAP = RT_LIST_CAP;      // per-ray list for the n1/n2 container walk and rt_intersect_world
constexpr int CONTAINER_CAP = LIST_CAP; // `containers` vector of prepare_computations: never longer than the list
// (transform chains have no depth limit: Shape::world_to_object / normal_to_world recurse without bound, src/shape.rs:394-420,
// and so do the kernels — every chain is a [length, xform, ...] run of chain_pool walked to its end)

// What prepare_computations needs to know about the hit node, in one 16-byte fetch (the SoA columns below stay for
// the kernels that need a single field).
struct NodeRec {
    int32_t kind;  // RT_* kind in bits 0-7; NODE_* bits of the node's material above
    int32_t param, nchain, material;
};
// material predicates the ray-tree stage bits are made of (rt_kernels.hip.h: k_trace epilogue, needs_walk)
constexpr int NODE_REFLECTS = 1;  // !near_eq(reflective, 0)                              src/world.rs:110
constexpr int NODE_REFRACTS = 2;  // !near_eq(transparency, 0)                            src/world.rs:121
constexpr int NODE_WALKS = 4;     // n1/n2 can be observed: NODE_REFRACTS, or reflective > 0 && transparency > 0 (schlick)
static_assert(sizeof(NodeRec) == 16, "NodeRec must be 16 bytes");

// Scene tables in HBM.  SoA per table; rows padded to what one lane fetches at once.
struct DScene {
    const NodeRec* node_rec;
    // nodes (src/shape.rs:25-37)
    const int32_t* node_kind;
    const int32_t* node_object_id;
    const int32_t* node_material;
    const int32_t* node_eq_class;
    const int32_t* node_param;
    const int32_t* node_nchain;       // offset into chain_pool: [len, x_leaf, x_parent, ... x_root]
    const uint8_t* node_casts_shadow;
    const double* xf;                 // [n_xforms][12]: rows 0..2 of the inverse (row 3 is 0,0,0,1)
    const double* group_box;          // [n_groups][6]
    const double* conic;              // [n_conics][2] minimum, maximum
    const uint8_t* conic_closed;
    const double* tri;                // [n_triangles][12] p1, e1, e2, normal
    const int32_t* tri_smooth;
    const double* smooth;             // [n_smooth][9] n1, n2, n3
    const double* mat;                // [n_materials][10]
    const int32_t* mat_pattern;
    const int32_t* pat_kind;
    const double* pat_ab;             // [n_patterns][6] a, b
    const double* pat_inv;            // [n_patterns][12]
    const double* light;              // [n_lights][6] position, intensity
    int32_t n_lights;
    int32_t n_nodes;
    // traversal program + acceleration
    const Item* items;
    int32_t n_items;
    const int32_t* chain_pool;
    const BvhNode* bvh;
    const MeshTri* mesh_tris;
    const int32_t* mesh_dups;         // node lists of repeated triangles (MeshTri::smooth bits 1..)
    const double* mesh_box;           // [n_meshes][6] padded f64 box of each mesh (ITEM_MESH.b indexes it)
    int32_t all_cast_shadow;
    int32_t n_lds;                    // BVH nodes [0, n_lds) are staged in LDS by the traversal kernels
    int32_t walk_at_zero;             // 1: some refractive index is not finite-positive, or some leaf is not equal to
                                      //    itself (F14): the n1/n2 walk cannot be skipped at remaining == 0 (needs_walk)
    int32_t passive_mesh_tris;        // 1: every triangle of a mesh BVH is equal (Shape::eq) to no other node: WalkSink may drop
                                      //    all but the last of them below the hit (rt_kernels.hip.h)
    int32_t walk_bound_item;          // >= 0: the program's first ITEM_MESH, with nothing but meshes behind it, in a scene whose mesh
                                      //       triangles are passive and share one refractive index: WalkSink::before_mesh
    int32_t _pad_walk;
    double walk_passive_ior;          // ... that refractive index
    int32_t tlas_root;                // BVH4 over the bounded top-level objects (Program::tlas_root), -1: none
    int32_t n_linear;                 // items [0, n_linear) are walked by every ray; the rest only through the top-level tree
    double cull_reach;                // ITEM_CULL ranges are trusted for rays whose origin has max |coordinate| below this
    double far_a, far_b, far_extent;  // SC_TLAS kernels: box growth (far_a D + far_b) D for a ray that starts D = max|origin| + far_extent away
    double cull_min_dir2;             // ... and whose direction has a squared length of at least this (0: any; see Program)
    // HBM arena of the BIG kernel variants (rt_kernels.hip.h: CsgScratch<true>, arena_list): one slab per lane
    char* arena;
    long long arena_lane_bytes;       // csg_cap * 32 + list_cap * 16
    int32_t arena_csg_cap, arena_list_cap;
    int32_t* spill;                   // [3 * deepest tree + 1 - LDS_STACK][spill_lanes] traversal stack entries beyond LDS_STACK
    long long spill_lanes;
};

// One recursion level: the ray queue and, index for index, the nodes of the ray tree at that depth.
struct Level {
    double *ox, *oy, *oz, *dx, *dy, *dz;  // ray (origin.x is NaN for a padding slot)
    double *ht, *hu, *hv;                 // hit t and the (u,v) normal_at must use (F5)
    int32_t* hkey;                        // node*4 + push index of the hit, -1 = miss
    double *px, *py, *pz, *nx, *ny, *nz;  // world-space point / normal of the hit (prepare_computations), k_trace -> k_shade
    int32_t* mode;                        // stage bits ST_* (rt_kernels.hip.h)
    int32_t *child_refl, *child_refr;     // index in the next level or -1 (=> BLACK)
    double *cr, *cg, *cb;                 // resolved colour of this node
    uint8_t* flags;                       // RT_FLAG_* per ray
    uint32_t* flags32;                    // the same bytes, for atomicOr from the queue stages
    int32_t* shadow_q;                    // ray indices whose shadow rays the next k_trace launch traces
    int32_t* cont_q;                      // ray indices queued for k_walk
    uint8_t* shadowed;                    // [ray][light]
};

// Device counters (one int64 each).  A frame uses one block of CNT_TOTAL counters per recursion level (plus one for
// the launch after the last level), all zeroed when the frame starts.
enum Counter : int {
    // the three words k_queue reserves queue slots from, one returning atomic per tile each: every one on a cache line
    // (16 counters) of its own — returning atomics on one line are served one after the other, chip-wide
    CNT_NEXT = 0,           // rays appended to the next level
    CNT_SHADOW_Q = 16,      // entries of the level's shadow queue (= nodes that hit something)
    CNT_CONTAINER_Q = 32,   // entries of the level's container queue (n1/n2 walks)
    // plain tallies (non-returning adds)
    CNT_REFLECT = 48,  // reflection rays spawned
    CNT_REFRACT,       // refraction rays spawned
    CNT_TIR,           // refraction slots reserved by k_queue that total internal reflection left dead
    CNT_OVERFLOW,      // block 0 only: level-0 rays whose tree ran into a bounded list (RT_FLAG_LIST_OVERFLOW)
    // work of the instrumented build (rt_render_opts.want_counters), RT_WORK_COUNT rows per kernel class
    CNT_W_XFORMS = 64,                               // closest-hit part of k_trace for this level's rays
    CNT_WS_XFORMS = CNT_W_XFORMS + RT_WORK_COUNT,    // shadow rays of THIS level's hits (traced by the next k_trace)
    CNT_WC_XFORMS = CNT_WS_XFORMS + RT_WORK_COUNT,   // k_walk
    CNT_COUNT = CNT_WC_XFORMS + RT_WORK_COUNT,       // what the host reads back
    // chunk feeds of the persistent traversal kernels: FEED_PARTS counters per kernel, FEED_STRIDE apart
    CNT_FEED_TRACE = 112,
    CNT_FEED_SHADOW = CNT_FEED_TRACE + FEED_PARTS * FEED_STRIDE,
    CNT_FEED_CONTAINER = CNT_FEED_SHADOW + FEED_PARTS * FEED_STRIDE,
    CNT_TOTAL = CNT_FEED_CONTAINER + FEED_PARTS * FEED_STRIDE
};
static_assert(CNT_COUNT <= CNT_FEED_TRACE, "feed counters overlap the plain ones");

struct CameraParams {
    double inv[12];
    double half_width, half_height, pixel_size;
    int32_t hsize, vsize;
    int32_t tiles_x, tiles_y;
    int32_t rank, n_ranks;
};

}  // namespace rt
