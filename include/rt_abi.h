/*
 * rt_abi.h — C ABI of the MI355X-native render path (Camera::render -> World::color_at).
 *
 * The reference crate (ashcolecarr/Ray-Tracer, Rust) has no FFI layer; this header is the
 * boundary its host side binds (see INTEGRATION.md for the Rust `extern "C"` block).  The
 * host keeps World/Shape/Material/Camera and scene construction; at render time it flattens
 * the scene into the structure-of-arrays tables below and calls rt_render().  Everything on
 * the device side (ray generation, scene traversal, intersection, CSG filtering,
 * prepare_computations, shadows, Phong + patterns, reflection/refraction/Schlick) is
 * hand-written HIP for gfx950 behind these entry points.
 *
 * Conventions
 *   - All floating point is IEEE f64, ids and indices are int32 (reference: src/lib.rs:46,64-66).
 *   - Matrices are 4x4 row-major, 16 doubles.  The kernels the library is built around drop the
 *     w of every tuple, which is exact while every inverse is affine — last row exactly
 *     (0,0,0,1), what Matrix::inverse (src/matrix.rs:124-154) yields for every transform built
 *     from src/transformation.rs.  A note on w: a scene with any other inverse (Matrix::new with
 *     hand-made values), and any frame whose camera has no inverse (the reference then traces
 *     Ray(ORIGIN, ORIGIN), src/camera.rs:44-46) or a non-affine one, is rendered by a second,
 *     4-component path: the reference's algorithm with nothing taken away and no acceleration
 *     structure, on the GPU, orders of magnitude slower.  Same entry points, same outputs.
 *   - Points/vectors are 3 doubles; the implied w is 1 for points and 0 for vectors.
 *   - The caller owns every input array for the duration of the call; the library copies
 *     what it keeps.  The caller allocates every output.  No ownership crosses the boundary
 *     except the opaque rt_scene handle, which only rt_scene_destroy frees.
 *   - Return value 0 = RT_OK, negative = error class; rt_last_error() gives the text
 *     (thread-local).  Nothing unwinds across the boundary.
 *   - The reference is single-threaded with process-global registries (src/lib.rs:46-54);
 *     a given rt_scene handle must be used from one host thread at a time.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

/* error classes */
#define RT_OK 0
#define RT_ERR_BAD_ARG (-1)      /* null pointer, negative count, index out of range          */
#define RT_ERR_UNSUPPORTED (-2)  /* a size beyond a documented limit, an entry point a scene cannot serve */
#define RT_ERR_HIP (-3)          /* a HIP runtime call failed (text has hipGetErrorString)    */
#define RT_ERR_NO_DEVICE (-4)    /* no gfx950 device / device index out of range              */
#define RT_ERR_OVERFLOW (-5)     /* rt_intersect_world / rt_is_shadowed: a ray needs more list entries than even the HBM
                                    arena variants hold (32768; 8192 pending inside one CSG subtree)        */

/* node kinds — enum Shape, src/shape.rs:25-37 (TestShape has no device counterpart) */
#define RT_SPHERE 0          /* src/sphere.rs          */
#define RT_PLANE 1           /* src/plane.rs           */
#define RT_CUBE 2            /* src/cube.rs            */
#define RT_CYLINDER 3        /* src/cylinder.rs        */
#define RT_CONE 4            /* src/cone.rs            */
#define RT_TRIANGLE 5        /* src/triangle.rs        */
#define RT_SMOOTH_TRIANGLE 6 /* src/smooth_triangle.rs */
#define RT_GROUP 7           /* src/group.rs           */
#define RT_CSG 8             /* src/csg.rs             */

/* CSG operation — the `operation: String` of src/csg.rs:18; any other string is UNKNOWN,
 * for which intersection_allowed returns false (src/csg.rs:102). */
#define RT_CSG_UNION 0
#define RT_CSG_INTERSECTION 1
#define RT_CSG_DIFFERENCE 2
#define RT_CSG_UNKNOWN 3

/* pattern kinds — enum Pattern, src/pattern.rs:7-14 */
#define RT_PAT_STRIPED 0
#define RT_PAT_GRADIENT 1
#define RT_PAT_RING 2
#define RT_PAT_CHECKERED 3
#define RT_PAT_RING_GRADIENT 4
#define RT_PAT_TEST 5

/* columns of material_params — struct Material, src/material.rs:11-21 */
#define RT_MAT_R 0
#define RT_MAT_G 1
#define RT_MAT_B 2
#define RT_MAT_AMBIENT 3
#define RT_MAT_DIFFUSE 4
#define RT_MAT_SPECULAR 5
#define RT_MAT_SHININESS 6
#define RT_MAT_REFLECTIVE 7
#define RT_MAT_TRANSPARENCY 8
#define RT_MAT_REFRACTIVE_INDEX 9
#define RT_MAT_STRIDE 10

/* per-pixel / per-ray flag bits */
#define RT_FLAG_REF_PANIC_UV 1u  /* reference would panic: SmoothTriangle::normal_at unwraps the u/v of
                                    intersections[0], which is not a smooth-triangle hit
                                    (src/intersection.rs:40, src/smooth_triangle.rs:98-100); the hit's own
                                    u,v were used instead */
#define RT_FLAG_NAN_T 2u         /* a NaN t reached a sort: reference panics in partial_cmp().unwrap()
                                    (src/world.rs:52, src/group.rs:59, src/csg.rs:71)                    */
#define RT_FLAG_LIST_OVERFLOW 4u /* a bounded device list overflowed even in the library's last resort: the kernels keep
                                  * 32 pending intersections per CSG subtree (128 nested) and 256 list entries for the n1/n2
                                  * walk in per-lane memory; a ray that needs more is traced again — automatically, inside
                                  * the same call — by kernel variants whose lists live in an HBM arena (8192 pending, as many
                                  * nested / 32768).  Only beyond THOSE sizes is the flag ever seen (and the pixel not guaranteed). */

/*
 * Flattened scene.  `nodes` is the intersection tree exactly as World::intersect_world walks it
 * (src/world.rs:46-55): world.objects in order, each subtree in depth-first pre-order, Group
 * children in `shapes` order (src/group.rs:54-57), CSG left subtree then right subtree
 * (src/csg.rs:67-70).  The node index therefore IS the reference's emission order, which is what
 * the stable sorts fall back to on equal t.
 */
typedef struct rt_scene_desc {
    int32_t abi_version; /* RT_ABI_VERSION */

    /* ---- nodes: one per Shape value in the tree (src/shape.rs:25-37) ---- */
    int32_t n_nodes;
    const int32_t* node_kind;        /* RT_SPHERE ... RT_CSG                                           */
    const int32_t* node_object_id;   /* Shape::get_id (ids start at 1, src/lib.rs:46)                  */
    const int32_t* node_parent;      /* tree parent node index, -1 for world.objects entries           */
    const int32_t* node_subtree_end; /* one past the last descendant (pre-order skip pointer)          */
    const int32_t* node_xform;       /* index into xform_*: inverse of this node's own transform, as
                                        Shape::intersect uses it (src/shape.rs:75)                     */
    const int32_t* node_ancestor;    /* leaves: index into ancestor_* of the registry entry of the
                                        shape's `parent` id (src/shape.rs:394-420), -1 if None         */
    const int32_t* node_material;    /* leaves: index into material_*; the ROOT ancestor's material
                                        (Shape::get_material, src/shape.rs:163-184)                    */
    const uint8_t* node_casts_shadow; /* the leaf's own flag (src/shape.rs:209-222)                    */
    const int32_t* node_eq_class;    /* leaves: two nodes compare equal under Shape::eq iff their
                                        classes are equal and >= 0; -1 = not even equal to itself
                                        (NaN/inf in an epsilon-compared field, e.g. an unbounded
                                        cylinder: src/cylinder.rs:26-33)                               */
    const int32_t* node_param;       /* GROUP -> group index, CSG -> csg index, CYLINDER/CONE -> conic
                                        index, TRIANGLE/SMOOTH_TRIANGLE -> triangle index, else -1     */

    /* ---- transforms ---- */
    int32_t n_xforms;
    const double* xform_inverse;           /* [n_xforms][16]  transform.inverse()                       */
    const double* xform_inverse_transpose; /* [n_xforms][16]  transform.inverse().transpose()           */

    /* ---- registry chain used by world_to_object / normal_to_world (src/lib.rs:47-49) ---- */
    int32_t n_ancestors;
    const int32_t* ancestor_xform;  /* index into xform_*                                               */
    const int32_t* ancestor_parent; /* next entry up the chain, -1 at the root                          */

    /* ---- Group gate boxes: Group::bounds_of in group-local space (src/group.rs:86-95) ---- */
    int32_t n_groups;
    const double* group_box; /* [n_groups][6] = min.x,min.y,min.z,max.x,max.y,max.z (may be +-inf)      */

    /* ---- CSG ---- */
    int32_t n_csg;
    const int32_t* csg_op; /* RT_CSG_*                                                                  */

    /* ---- cylinders and cones (src/cylinder.rs:15-24, src/cone.rs:15-24) ---- */
    int32_t n_conics;
    const double* conic_minimum;
    const double* conic_maximum;
    const uint8_t* conic_closed;

    /* ---- triangles (src/triangle.rs:12-24, src/smooth_triangle.rs:12-27) ---- */
    int32_t n_triangles;
    const double* tri_p1;     /* [n_triangles][3] point1                                               */
    const double* tri_e1;     /* [n_triangles][3] edge_vector1 = p2 - p1                               */
    const double* tri_e2;     /* [n_triangles][3] edge_vector2 = p3 - p1                               */
    const double* tri_normal; /* [n_triangles][3] normalize(e2 x e1)                                   */
    const int32_t* tri_smooth; /* index into smooth_* or -1 for a flat triangle                        */
    int32_t n_smooth;
    const double* smooth_n1; /* [n_smooth][3] normal_vector1..3                                        */
    const double* smooth_n2;
    const double* smooth_n3;

    /* ---- materials and patterns (src/material.rs, src/pattern.rs) ---- */
    int32_t n_materials;
    const double* material_params;   /* [n_materials][RT_MAT_STRIDE]                                   */
    const int32_t* material_pattern; /* index into pattern_* or -1                                     */
    int32_t n_patterns;
    const int32_t* pattern_kind;   /* RT_PAT_*                                                         */
    const double* pattern_a;       /* [n_patterns][3]                                                  */
    const double* pattern_b;       /* [n_patterns][3]                                                  */
    const double* pattern_inverse; /* [n_patterns][16] pattern transform.inverse() (src/pattern.rs:25) */

    /* ---- lights (src/light.rs:5-20), in world.lights order ---- */
    int32_t n_lights;
    const double* light_position;  /* [n_lights][3] */
    const double* light_intensity; /* [n_lights][3] */
} rt_scene_desc;

/* struct Camera after Camera::new, src/camera.rs:10-35 */
typedef struct rt_camera_desc {
    int32_t hsize;
    int32_t vsize;
    int32_t invertible; /* transform.inverse().is_some(); 0: every pixel's ray is Ray(ORIGIN, ORIGIN), src/camera.rs:44-46 */
    int32_t _pad;
    double inverse[16]; /* transform.inverse() */
    double half_width;
    double half_height;
    double pixel_size;
} rt_camera_desc;

typedef struct rt_render_opts {
    int32_t max_depth;    /* `remaining` passed to color_at; Camera::render uses DEFAULT_RECURSION = 5
                             (src/camera.rs:68, src/lib.rs:41)                                          */
    int32_t rank;         /* tile shard of this call, 0 <= rank < n_ranks                               */
    int32_t n_ranks;      /* 1 = whole frame                                                            */
    int32_t want_counters; /* 1 = also run the instrumented traversal and fill rt_stats.work_*          */
    void* stream;         /* hipStream_t to enqueue on, NULL = the library's own stream                 */
} rt_render_opts;

/* rows of rt_stats.work_* */
#define RT_WORK_XFORMS 0     /* ray transforms (2 mat-vec each, src/ray.rs:25-30)   */
#define RT_WORK_BOX_TESTS 1  /* acceleration-BVH box tests                          */
#define RT_WORK_GATE_TESTS 2 /* reference Group gate tests (src/bound.rs:85)        */
#define RT_WORK_TRI_TESTS 3  /* triangle tests entered                              */
#define RT_WORK_TRI_DET_REJ 4 /*  ... rejected at |det| < EPSILON                   */
#define RT_WORK_TRI_U_REJ 5  /*   ... rejected at u                                 */
#define RT_WORK_TRI_V_REJ 6  /*   ... rejected at v                                 */
#define RT_WORK_PRIM0 7      /* +0..4: sphere, plane, cube, cylinder, cone tests    */
#define RT_WORK_BVH_NODES 12 /* BVH nodes fetched (128 B each)                      */
#define RT_WORK_COUNT 13

typedef struct rt_stats {
    /* rays actually traced, by class (BASELINE.md ray accounting) */
    int64_t rays_primary;
    int64_t rays_reflect;
    int64_t rays_refract;
    int64_t rays_shadow;
    int64_t rays_container; /* extra list-collection passes for the n1/n2 walk (not counted as rays) */
    int64_t shaded_nodes;   /* nodes of the recursion tree that hit something                         */
    int32_t levels;         /* recursion levels that held at least one ray                            */
    int32_t _pad;
    /* device time per kernel class, ms, measured with HIP events on the launch stream around every launch of the SAME
     * schedule the untimed calls run (no host wait between launches; the events are read after the frame).
     * Field names say which kernel they time (round 3 renamed them: the layout is round 2's, where the same slots were
     * called ms_collect / ms_shade / ms_resolve / n_shadow_launches after kernels that no longer exist — a reader of
     * the old names fails to compile instead of silently misattributing time). */
    double ms_total;
    double ms_raygen;  /* k_raygen; with rt_scene_opts.fuse_raygen (the default) level 0's k_trace makes the primary rays and
                          this is the few microseconds of k_set_camera — ray_for_pixel's time is then part of ms_trace   */
    double ms_trace;   /* k_trace launches: closest hits of level d + the shadow rays of level d-1, one launch per level,
                          plus one for the shadow rays of the last level                                                */
    double ms_shadow;  /* always 0: the shadow rays have no launch of their own (slot kept for layout stability)        */
    double ms_walk;    /* k_walk launches: the n1/n2 container walks                                                    */
    double ms_queue;   /* k_queue (queue compaction; + k_spawn when rt_scene_opts.fuse_spawn is 0)                      */
    double ms_kshade;  /* k_shade: lighting + schlick + bottom-up blend                                                 */
    int32_t n_trace_launches; /* k_trace launches                                                                       */
    int32_t n_walk_launches;  /* k_walk launches                                                                        */
    /* algorithmic work counted by the instrumented kernels (want_counters = 1), per kernel class; the
     * RT_WORK_* indices name the rows, SURVEY.md section 8(d) constants turn them into flops / bytes */
    int64_t work_trace[RT_WORK_COUNT];     /* closest-hit part of k_trace         */
    int64_t work_shadow[RT_WORK_COUNT];    /* shadow-ray part of k_trace          */
    int64_t work_container[RT_WORK_COUNT]; /* k_walk                              */
} rt_stats;

typedef struct rt_scene rt_scene; /* opaque: device-resident tables + acceleration structure + queues */

/* Number of usable HIP devices (0 when none / no driver).  No reference counterpart: the crate is CPU-only. */
int rt_device_count(void);

/* Copy the flattened scene to device `device`, build the (non-semantic, conservative) acceleration BVH over
 * real triangle bounds, and return a handle.  Replaces the per-render use the reference makes of World
 * (src/world.rs:15-18) and of the global registries (src/lib.rs:46-54) — all resolved by the flattener. */
int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);

/* Behaviour switches of a scene, fixed at creation (no counterpart in the reference: none of them can change a result,
 * every one has an on/off parity test).  In every field -1 means "library default"; the RT_AMD_* environment variable
 * named with a field is read only when the field is -1 and exists for A/B measurements of an unmodified host. */
typedef struct rt_scene_opts {
    int32_t struct_size;        /* sizeof(rt_scene_opts): lets the library accept an older, shorter struct            */
    int32_t cull_ranges;        /* world-space cull ranges over the top-level objects: -1/1 the library's rule (worlds
                                   with >= 6 bounded top-level objects), 0 never                    [RT_AMD_CULL]      */
    int32_t passive_triangles;  /* n1/n2 walk keeps only the last mesh triangle below the hit when that is provably
                                   enough: -1/1 yes, 0 lists every triangle                  [RT_AMD_PASSIVE_TRIS]     */
    int32_t frame_graph;        /* replay a frame's launches as one HIP graph: -1/1 yes, 0 plain enqueue [RT_AMD_GRAPH] */
    int32_t device_bvh;         /* build the BVH of large meshes (>= 1024 distinct triangles) on the GPU — Morton order,
                                   radix tree, boxes, leaves, BVH4 collapse — instead of the host's binned-SAH build:
                                   -1/0 no (the SAH tree traces faster), 1 yes              [RT_AMD_DEVICE_BVH]        */
    int32_t gather;             /* rt_render_multi: -1 RCCL when there is more than one device and it loads, else peer
                                   copies; 1 RCCL even for one device; 2 peer copies always        [RT_AMD_GATHER]     */
    int32_t verbose;            /* -1/0 quiet, 1 one line of scene statistics on stderr          [RT_AMD_VERBOSE]      */
    /* (fields below: struct_size >= 40; a shorter struct leaves them at the default) */
    int32_t share_meshes;       /* meshes with identical triangle content take one BVH: -1/1 yes, 0 a tree each
                                                                                              [RT_AMD_SHARE_MESHES]    */
    int32_t fuse_output;        /* level 0's shading pass writes the frame / shard itself: -1/1 yes, 0 a k_output launch
                                                                                               [RT_AMD_FUSE_OUTPUT]    */
    int32_t fuse_spawn;         /* the launch that traces a level makes its reflection / refraction rays: -1/1 yes, 0 a
                                   k_spawn launch writes them first                             [RT_AMD_FUSE_SPAWN]    */
    /* (struct_size >= 44) */
    int32_t top_level_tree;     /* worlds with 64 or more bounded top-level objects: a BVH4 over the objects' world boxes,
                                   walked per ray (nearest first, stack in LDS), names the objects a ray visits, instead of
                                   nested cull ranges walked in program order; lanes whose rays meet different instances of
                                   a mesh walk them side by side: -1/1 yes, 0 ranges only, 2 the tree without per-lane mesh
                                   instances (every object left to the wavefront)                          [RT_AMD_TLAS]      */
    /* (struct_size >= 48) */
    int32_t four_components;    /* -1/0: the library's rule — the 4-component path (see the note on w below) only for scenes
                                   and cameras that need it; 1: every call on this scene takes it, whatever its matrices
                                   (orders of magnitude slower: for checking that path against the other) [RT_AMD_GENERAL]    */
    /* (struct_size >= 52) */
    int32_t walk_bound;         /* the n1/n2 walk searches a mesh only as far back along the ray as a triangle there could still
                                   be the last container at the hit (worked out from the entries of the other shapes, which
                                   are listed first): -1/1 yes, 0 the whole line                       [RT_AMD_WALK_BOUND]    */
    /* (struct_size >= 56) */
    int32_t render_lanes;       /* rt_render_begin / rt_render_frame_begin / rt_render_shard_begin: how many frames in flight
                                   are RENDERED side by side (each on a stream and on queues of its own): 1..3, or -1 for the
                                   library's choice: 3 when the process runs with GPU_MAX_HW_QUEUES >= 6 (the HIP runtime's
                                   variable: streams that share a hardware queue run one after the other, and with the
                                   default 4 a third lane's stream shares one), else 2                  [RT_AMD_LANES]      */
    /* (struct_size >= 60) */
    int32_t fuse_raygen;        /* the launch that traces level 0 makes the primary rays from the camera (ray_for_pixel,
                                   src/camera.rs:37-60) as it fetches them: -1/1 yes, 0 a k_raygen launch writes them first
                                                                                              [RT_AMD_FUSE_RAYGEN]    */
} rt_scene_opts;

/* rt_scene_create / rt_scene_create_multi with explicit switches: n_gpus = 1 and devices = {device} is
 * rt_scene_create; opts NULL = all defaults. */
int rt_scene_create_opts(const rt_scene_desc* desc, int32_t n_gpus, const int32_t* devices, const rt_scene_opts* opts,
                         rt_scene** out);

/* Page-locked host memory for the output buffers of rt_render / rt_render_multi (no counterpart in the reference: it is
 * what `vec![0f64; n]` becomes in Camera::render_gpu).  The D2H copy of a frame into pageable memory is staged by the HIP
 * runtime at ~17 GB/s; into memory from rt_host_alloc it is one DMA at PCIe rate: rt_render of the 1920x1080 headline
 * frame takes 2.65 ms instead of 4.65 ms (49.8 MB of f64 RGB).  Any host pointer works for every entry point; this one
 * is just faster.  rt_host_free(NULL) is a no-op. */
int rt_host_alloc(size_t bytes, void** out);
void rt_host_free(void* p);

/* Free everything rt_scene_create and later renders allocated for this scene. */
void rt_scene_destroy(rt_scene* scene);

/* Camera::render (src/camera.rs:62-74): full frame, host output buffers.
 *   rgb_out     [vsize*hsize*3] row-major f64 RGB; last row and last column stay 0.0 (src/camera.rs:65-66)
 *   hit_id_out  nullable [vsize*hsize] object id of the primary hit, 0 = miss
 *   flags_out   nullable [vsize*hsize] RT_FLAG_* bits (OR over the pixel's whole ray tree)
 *   stats_out   nullable */
int rt_render(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* rgb_out,
              int32_t* hit_id_out, uint8_t* flags_out, rt_stats* stats_out);

/* Camera::render (src/camera.rs:62-74, which RETURNS a host Canvas) for callers that render frame after frame: the
 * same frame as rt_render into the same HOST buffers, with several frames in flight.  rt_render_begin hands the frame (whole
 * frames only: opts->rank 0 of 1, opts->stream ignored) to one of the render lanes (three at most, rt_scene_opts.render_lanes) and returns; a lane renders into a
 * device frame of its own on a stream of its own and copies to rgb_out / hit_id_out / flags_out (the last two nullable) on a
 * copy stream; the buffers hold the frame once the matching rt_render_end has returned.  At most SIX frames may be in
 * flight (with three lanes: three being rendered — frames are dealt to the render lanes in turn, and a begin waits for the
 * render of the frame three before it — and three being copied); rt_render_end retires the oldest one (and returns RT_OK at once when none
 * is) and returns ITS result: an error of a frame's render surfaces there, not at its begin (rt_last_error has the text).
 *     begin(A); begin(B); begin(C); begin(D); end() -> A complete; begin(E); end() -> B complete; begin(A'); ...
 * Round 2 overlapped the 0.85 ms copy of a 1920x1080 f64 frame with the next frame's render (2.5 -> 1.63 ms per frame at the
 * host); round 3 also renders the frames in flight side by side — each traversal launch ends in a tail of a few long rays
 * during which most of the chip is idle, and a launch of ANOTHER frame's stream fills it — so a
 * sequence of frames comes out at 1.05-1.1 ms per frame where one frame alone takes 1.5 (two lanes: 1.1; three: 1.06; the
 * smaller the frame, the more the third is worth: a rank's shard of an 8-way split 0.77 / 0.44 / 0.33 ms for one / two /
 * three at once).  Lanes two and three are shallow copies of the scene made at the first begin: they share the scene's
 * read-only tables and own their queues (each about as much device memory again as the scene's queues take, grown when the
 * lane is first used; rt_scene_opts.render_lanes = 1 or 2 for fewer).  A caller that keeps fewer frames in flight uses as
 * many lanes as it has frames in flight.  Buffers from rt_host_alloc keep the copy off the critical path (a pageable buffer
 * makes the runtime stage it).  The caller must not touch a buffer between its begin and the end that retires it; any other
 * call on the scene first waits for the renders in flight.  rt_scene_destroy retires everything.
 *
 * rt_render_frame_begin is the same with the Canvas left in DEVICE memory (rt_render_frame's arguments; frames in flight
 * must be different buffers; three in flight are all it can use); it shares the lanes, and rt_render_end retires its frames
 * too. */
int rt_render_begin(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* rgb_out,
                    int32_t* hit_id_out, uint8_t* flags_out);
int rt_render_frame_begin(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* d_rgb_frame,
                          int32_t* d_hit_id_frame, uint8_t* d_flags_frame);
int rt_render_end(rt_scene* scene);

/* Camera::render's pixel loop (src/camera.rs:65-71) for ONE SHARD of the frame, frame after frame with up to three shards
 * in flight: rt_render_shard's arguments and output (shard opts->rank of opts->n_ranks in shard-local pixel order, DEVICE
 * buffers, opts->stream ignored), handed to the same render lanes as rt_render_frame_begin and retired by the same
 * rt_render_end — when that returns the shard is complete in device memory and work enqueued on ANY stream afterwards (an
 * RCCL gather, rt_assemble) may read it.  For one-process-per-GPU drivers that render a sequence of frames: a rank's
 * shard of a 1920x1080 frame is nearly all launch tails, which the next frame's shard fills.
 * Shards in flight must be different buffers. */
int rt_render_shard_begin(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* d_rgb_out,
                          int32_t* d_hit_id_out, uint8_t* d_flags_out);

/* Number of pixels shard `rank` of `n_ranks` renders (8x8-pixel tiles of the (hsize-1)x(vsize-1) domain dealt
 * round-robin).  Negative on bad arguments. */
int64_t rt_shard_pixel_count(const rt_camera_desc* camera, int32_t rank, int32_t n_ranks);

/* One shard of Camera::render with DEVICE output pointers (for one-process-per-GPU drivers that gather the
 * shards with RCCL).  d_rgb_out [count*3], d_hit_id_out nullable [count], d_flags_out nullable [count], in
 * shard-local pixel order; count = rt_shard_pixel_count().  All work is ordered on opts->stream (the scene's own
 * stream when that is NULL).  The call returns once the shard's ray tree has been walked — the host reads the
 * per-level ray counts to learn whether a queue has to grow — so it is not a fire-and-forget enqueue; with a caller
 * stream the output writes are complete or ordered on that stream, and work enqueued on it afterwards (an RCCL
 * gather, rt_assemble) sees them without a host wait; with the scene's stream the call waits for them. */
int rt_render_shard(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* d_rgb_out,
                    int32_t* d_hit_id_out, uint8_t* d_flags_out, rt_stats* stats_out);

/* How many triangle meshes of the scene got their BVH topology from the GPU build (Morton sort + binary radix tree,
 * csrc/rt_lbvh.hip; opt-in with RT_AMD_DEVICE_BVH=1, meshes of 1024 triangles and more) rather than the host's binned
 * SAH.  No counterpart in the reference (its groups have no acceleration for triangles): this is the "what was built"
 * read-back for SURVEY.md section 8(f) row 1. */
int32_t rt_scene_meshes_built_on_gpu(const rt_scene* scene);

/* How many triangle meshes of the scene took the BVH of an earlier mesh with identical content (the same OBJ text
 * parsed again, a group cloned into several places: same triangles in the same order under their group) instead of
 * getting a tree of their own.  No counterpart in the reference; read-back for tests and tools. */
int32_t rt_scene_meshes_shared(const rt_scene* scene);

/* Camera::render (src/camera.rs:62-74) with the Canvas left in DEVICE memory: d_rgb_frame [vsize*hsize*3] row-major
 * (Canvas::write_pixel layout, src/canvas.rs:25-27), d_hit_id_frame / d_flags_frame nullable [vsize*hsize].  The
 * pixels of shard opts->rank of opts->n_ranks are written where they belong in the frame, the never-rendered last row
 * and column (F1) are cleared, everything else is left alone: with n_ranks == 1 this is the whole render, one call and
 * no intermediate copy; ranks of one process may also fill one frame in turn.  Stream ordering and blocking as for
 * rt_render_shard. */
int rt_render_frame(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* d_rgb_frame,
                    int32_t* d_hit_id_frame, uint8_t* d_flags_frame, rt_stats* stats_out);

/* ---- one process, several GPUs of one node (no counterpart in the reference, which is
 * single-threaded CPU code: this is the n_gpus of the Camera::render_gpu(world, depth, n_gpus) a host binds) ----
 *
 * rt_scene_create_multi: like rt_scene_create on each of n_gpus devices (`devices` [n_gpus], NULL = 0..n_gpus-1); the
 * handle is that of the replica on devices[0] and owns the others: rt_scene_destroy frees everything, and every other
 * entry point (rt_render, rt_color_at, ...) works on it as on a single-GPU scene of devices[0].
 * rt_render_multi: Camera::render (src/camera.rs:62-74) with the (hsize-1)x(vsize-1) pixel domain cut into 8x8 tiles
 * dealt round-robin to the devices (pixels are independent, src/camera.rs:65-71; a pixel's value does not depend on
 * the partition), one host thread per device, the shards gathered to devices[0] with ONE ncclGather per array (RCCL
 * over xGMI, loaded on first use; hipMemcpyPeerAsync when RCCL cannot be used — rt_scene_gather_kind tells which) and
 * scattered into the row-major frame there.  Outputs are HOST buffers as in rt_render, all nullable except that one of
 * rgb_out / quantised_out must be given: quantised_out [vsize*hsize*3] receives scale_color (src/canvas.rs:86-98) of
 * the frame, computed on each device before the gather (an 8x smaller gather and D2H copy than the f64 frame).
 * stats_per_gpu: NULL or [n_gpus].  opts->rank / n_ranks / stream are ignored. */
int rt_scene_create_multi(const rt_scene_desc* desc, int32_t n_gpus, const int32_t* devices, rt_scene** out);
int rt_render_multi(rt_scene* scene, const rt_camera_desc* camera, const rt_render_opts* opts, double* rgb_out,
                    int32_t* hit_id_out, uint8_t* flags_out, uint8_t* quantised_out, rt_stats* stats_per_gpu);
/* Replicas of the scene (1 for rt_scene_create), and how rt_render_multi gathers: 0 nothing to gather (one GPU),
 * 1 ncclGather (RCCL), 2 hipMemcpyPeerAsync. */
int32_t rt_scene_gpu_count(const rt_scene* scene);
int32_t rt_scene_gather_kind(const rt_scene* scene);

/* Scatter gathered shards (concatenated in rank order, DEVICE memory, on `scene`'s device) into a full
 * row-major frame, zero elsewhere.  d_rgb_frame [vsize*hsize*3] device.  d_hit_id_* nullable. */
int rt_assemble(rt_scene* scene, const rt_camera_desc* camera, int32_t n_ranks, const double* d_rgb_shards,
                const int32_t* d_hit_id_shards, double* d_rgb_frame, int32_t* d_hit_id_frame, void* stream);

/* The same for shards that sit `shard_stride` pixels apart instead of back to back (shard r starts at pixel
 * r * shard_stride of d_rgb_shards / d_hit_id_shards) — the layout an RCCL gather of equally sized, padded buffers
 * leaves on the root, so that no repacking is needed between the gather and this call.  shard_stride == 0: back to
 * back, as rt_assemble.  Part of the same replacement of Canvas::write_pixel's loop (src/camera.rs:65-71). */
int rt_assemble_strided(rt_scene* scene, const rt_camera_desc* camera, int32_t n_ranks, const double* d_rgb_shards,
                        const int32_t* d_hit_id_shards, int64_t shard_stride, double* d_rgb_frame,
                        int32_t* d_hit_id_frame, void* stream);

/* Canvas::scale_color (src/canvas.rs:86-98) for n_values f64 channel values in DEVICE memory: clamp to [0, 1], times 255,
 * f64::round (half away from zero), NaN -> 0.  What a one-process-per-GPU driver runs on its shard before the gather:
 * 3 bytes per pixel over xGMI instead of 24.  Ordered on `stream` (NULL: the scene's stream, and the call waits). */
int rt_quantise(rt_scene* scene, const double* d_rgb, int64_t n_values, uint8_t* d_quantised_out, void* stream);

/* rt_assemble_strided for quantised shards (u8 RGB, 3 bytes per pixel; shard r starts at pixel r * shard_stride, 0 = back
 * to back): the scatter of Canvas::write_pixel's loop (src/camera.rs:65-71) for the u8 frame Canvas::canvas_to_ppm reads
 * (src/canvas.rs:33-84).  d_q_frame [vsize*hsize*3] device, zero where no shard writes. */
int rt_assemble_quantised(rt_scene* scene, const rt_camera_desc* camera, int32_t n_ranks, const uint8_t* d_q_shards,
                          int64_t shard_stride, uint8_t* d_q_frame, void* stream);

/* World::color_at (src/world.rs:81-90) for a batch of arbitrary rays, host buffers.
 *   origins, directions [n_rays][3]; rgb_out [n_rays][3]; hit_id_out / flags_out nullable [n_rays]. */
int rt_color_at(rt_scene* scene, int64_t n_rays, const double* origins, const double* directions, int32_t remaining,
                double* rgb_out, int32_t* hit_id_out, uint8_t* flags_out, rt_stats* stats_out);

/* World::intersect_world (src/world.rs:46-55) for a batch of rays: the sorted intersection list.
 *   count_out [n_rays] true list length (may exceed max_hits, then only the first max_hits are stored);
 *   t_out / object_id_out [n_rays][max_hits]. */
int rt_intersect_world(rt_scene* scene, int64_t n_rays, const double* origins, const double* directions,
                       int32_t max_hits, int32_t* count_out, double* t_out, int32_t* object_id_out);

/* World::is_shadowed (src/world.rs:92-107) for a batch of points: out [n_points][n_lights], 1 = shadowed. */
int rt_is_shadowed(rt_scene* scene, int64_t n_points, const double* points, uint8_t* out);

/* ---- the path's inner functions, one entry point each: what the reference's own unit tests call directly
 * (src/intersection.rs:235-410, src/shape.rs:678-730, src/material.rs:161-277, src/pattern.rs:226-426), so that their
 * vectors can be replayed on the device without an oracle in between (tests/test_gpu_kats.py). ---- */

/* columns of rt_prepare_computations' output rows — struct Computations, src/computations.rs:6-18 */
#define RT_COMP_T 0
#define RT_COMP_POINT 1    /* 3 */
#define RT_COMP_EYE 4      /* 3 */
#define RT_COMP_NORMAL 7   /* 3 */
#define RT_COMP_INSIDE 10  /* 0.0 / 1.0 */
#define RT_COMP_OVER 11    /* 3 */
#define RT_COMP_UNDER 14   /* 3 */
#define RT_COMP_REFLECT 17 /* 3 */
#define RT_COMP_N1 20
#define RT_COMP_N2 21
#define RT_COMP_SCHLICK 22 /* Intersection::schlick(comps), src/intersection.rs:97-115 */
#define RT_COMP_U 23       /* the u, v normal_at was given: those of intersections[0] (src/intersection.rs:40) */
#define RT_COMP_V 24
#define RT_COMP_COUNT 25

/* Intersection::prepare_computations (src/intersection.rs:38-95) for `Intersection::hit(world.intersect_world(ray))` of
 * each ray (host buffers): hit_id_out [n_rays] object id of the hit (0: no hit, the row is zero), comps_out
 * [n_rays][RT_COMP_COUNT], flags_out nullable.  The n1/n2 walk runs for every hit, as in the reference. */
int rt_prepare_computations(rt_scene* scene, int64_t n_rays, const double* origins, const double* directions, int32_t* hit_id_out,
                            double* comps_out, uint8_t* flags_out);

/* Shape::normal_at (src/shape.rs:92-108: world_to_object down the parent chain, the kind's local normal, normal_to_world
 * back up) of leaf node `node` (index into rt_scene_desc's node tables) for world points [n][3] -> normals_out [n][3]
 * (host buffers); uv nullable [n][2]: the u, v a SmoothTriangle interpolates with (src/smooth_triangle.rs:98-104). */
int rt_normal_at(rt_scene* scene, int32_t node, int64_t n, const double* points, const double* uv, double* normals_out);

/* Material::lighting (src/material.rs:97-129) with the material of leaf node `node` (its root ancestor's, F7; a pattern is
 * evaluated through Pattern::pattern_at_shape, src/pattern.rs:23-35, i.e. in the leaf's object space) and light `light`:
 * points / eyes / normals [n][3], in_shadow [n] -> rgb_out [n][3] (host buffers). */
int rt_lighting(rt_scene* scene, int32_t node, int32_t light, int64_t n, const double* points, const double* eyes,
                const double* normals, const uint8_t* in_shadow, double* rgb_out);

/* Canvas::canvas_to_ppm (src/canvas.rs:33-84) + scale_color (src/canvas.rs:86-98) on the device, byte for byte:
 * "P3\n{w} {h}\n255\n", then per canvas row the 3*w quantised values, a value that would make a line reach 70
 * characters starts a new line, every row ends its line.  d_rgb_frame: DEVICE pointer, row-major vsize x hsize x 3 f64
 * (what rt_assemble / rt_render_shard leave in HBM).  d_quantised_out: DEVICE pointer or NULL, vsize x hsize x 3 u8
 * (the scale_color values: 8x smaller than the f64 frame for a gather or a D2H copy).  ppm_out: HOST buffer of
 * `cap` bytes or NULL.  Returns the length of the PPM text in bytes (whether or not it fitted; nothing is written
 * past cap), or a negative RT_ERR_* code. */
int64_t rt_canvas_to_ppm(rt_scene* scene, const double* d_rgb_frame, int32_t hsize, int32_t vsize, uint8_t* d_quantised_out,
                         char* ppm_out, int64_t cap);

/* rt_canvas_to_ppm for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code under HIP
 * graph capture, tight serving loops and harnesses that forbid hipMalloc.  It calls no hipMalloc / hipFree / allocating
 * routine: its only scratch (the per-row offsets of the text, vsize + 1 int64) lives in a file-scope device array of
 * the library (128 KiB per GPU, held until the process ends), so vsize is limited to 16384 rows (RT_ERR_UNSUPPORTED
 * beyond).  Every pointer is a DEVICE pointer owned by the caller: d_rgb_frame [vsize*hsize*3] f64 in,
 * d_quantised_out [vsize*hsize*3] u8 out (REQUIRED here), d_ppm_out [cap] the whole PPM text, header included
 * (nothing is written unless all of it fits), d_len_out nullable.  With d_len_out the text length is stored there in
 * stream order, the call returns 0 without waiting and never blocks; without it the call waits for the stream and
 * returns the length.  `stream` NULL = the scene's own stream.  All calls on one GPU share that GPU's scratch array:
 * calls that overlap on one GPU (two streams, two threads) race — serialise them. */
int64_t rt_canvas_to_ppm_noalloc(rt_scene* scene, const double* d_rgb_frame, int32_t hsize, int32_t vsize,
                                 uint8_t* d_quantised_out, char* d_ppm_out, int64_t cap, int64_t* d_len_out, void* stream);

/* Camera::ray_for_pixel (src/camera.rs:37-60) for a batch of pixels, evaluated by the device ray-generation
 * code on device 0: origins_out / directions_out [n][3] (host). */
int rt_ray_for_pixel(const rt_camera_desc* camera, int64_t n, const int32_t* px, const int32_t* py, double* origins_out,
                     double* directions_out);

/* Text of the last error on this thread ("" if none). */
const char* rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
