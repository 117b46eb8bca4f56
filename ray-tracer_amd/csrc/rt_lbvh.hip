// rt_lbvh.hip — the GPU BVH build (rt_scene_opts.device_bvh).
//
// For one triangle mesh, everything between "bounds of the distinct triangles" and "BVH4 nodes the traversal kernels
// read" runs on the device:
//   1. k_morton        63-bit Morton codes of the triangle centroids inside the mesh box
//   2. hipCUB          radix sort of (code, triangle) pairs
//   3. k_radix_tree    binary radix tree over the sorted codes (Karras, "Maximizing Parallelism in the Construction of
//                      BVHs, Octrees, and k-d Trees", HPG 2012) + k_parents
//   4. k_boxes         padded f64 boxes bottom-up: every sorted triangle climbs towards the root, the second arrival
//                      at a node unites its children (one atomic counter per node)
//   5. k_collapse_*    breadth-first collapse into 4-wide nodes, one launch pair per BVH4 level: a node takes its two
//                      binary children's children (a subtree of at most BVH_LEAF_MAX triangles is a leaf; with three
//                      slots the first inner one is opened too — the rule rt_program.cpp's host collapse uses), child
//                      boxes rounded outward to f32, next level's node indices from an exclusive scan
// The host receives the triangle order, the finished nodes in breadth-first (= depth) order and the root box, and only
// renumbers them into the scene-wide node array.  Like the host's SAH tree this is acceleration only: which triangles
// get tested, never what a test returns (renders are bit-identical with either tree, tests/test_gpu_parity.py).
// A radix tree can be deeper than the traversal stack allows (3 pending siblings per BVH4 level, BVH_STACK entries):
// then the build declines and the caller falls back to the host build.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "rt_program.h"

namespace rt {
namespace {

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {  // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// key = 63-bit Morton code of the centroid (21 bits per axis inside the mesh box); ties are broken by position below
__global__ void k_morton(size_t n, const double* __restrict__ lo, const double* __restrict__ hi, double lx, double ly, double lz,
                         double sx, double sy, double sz, unsigned long long* __restrict__ key, int32_t* __restrict__ idx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto q = [](double v, double l, double s) {
        double t = (v - l) * s;  // [0, 1]
        if (!(t > 0.)) t = 0.;
        if (t > 1.) t = 1.;
        return (unsigned long long)(t * 2097151.0);
    };
    const double cx = 0.5 * lo[3 * i] + 0.5 * hi[3 * i], cy = 0.5 * lo[3 * i + 1] + 0.5 * hi[3 * i + 1], cz = 0.5 * lo[3 * i + 2] + 0.5 * hi[3 * i + 2];
    key[i] = (spread21(q(cx, lx, sx)) << 2) | (spread21(q(cy, ly, sy)) << 1) | spread21(q(cz, lz, sz));
    idx[i] = (int32_t)i;
}

// common-prefix length of sorted entries i and j (codes may repeat: the position breaks the tie, as in the paper)
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ key, long long n, long long i, long long j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = key[i], b = key[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clzll((long long)((unsigned long long)i ^ (unsigned long long)j));
}

// internal node i of the binary radix tree over n sorted keys: children (>= 0 internal, < 0: ~position of a leaf) and
// the range of positions it covers
__global__ void k_radix_tree(long long n, const unsigned long long* __restrict__ key, int32_t* __restrict__ left,
                             int32_t* __restrict__ right, int32_t* __restrict__ first, int32_t* __restrict__ last) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = delta(key, n, i, i + 1) - delta(key, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(key, n, i, i - d);
    long long lmax = 2;
    while (delta(key, n, i, i + lmax * d) > dmin) lmax *= 2;
    long long l = 0;
    for (long long t = lmax / 2; t >= 1; t /= 2)
        if (delta(key, n, i, i + (l + t) * d) > dmin) l += t;
    const long long j = i + l * d;
    const int dnode = delta(key, n, i, j);
    long long s = 0;
    for (long long div = 2;; div *= 2) {
        const long long t = (l + div - 1) / div;
        if (delta(key, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const long long gamma = i + s * d + (d < 0 ? -1 : 0);
    const long long lo = i < j ? i : j, hi = i < j ? j : i;
    left[i] = lo == gamma ? ~(int32_t)gamma : (int32_t)gamma;
    right[i] = hi == gamma + 1 ? ~(int32_t)(gamma + 1) : (int32_t)(gamma + 1);
    first[i] = (int32_t)lo;
    last[i] = (int32_t)hi;
}

// parent of every internal node (parent_int) and of every sorted position (parent_leaf); the root keeps -1
__global__ void k_parents(long long n, const int32_t* __restrict__ left, const int32_t* __restrict__ right, int32_t* __restrict__ parent_int,
                          int32_t* __restrict__ parent_leaf) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int32_t l = left[i], r = right[i];
    if (l >= 0) parent_int[l] = (int32_t)i; else parent_leaf[~l] = (int32_t)i;
    if (r >= 0) parent_int[r] = (int32_t)i; else parent_leaf[~r] = (int32_t)i;
}

struct Box6 {
    double v[6];  // lo xyz, hi xyz
};
__device__ __forceinline__ Box6 tri_box(const double* __restrict__ lo, const double* __restrict__ hi, int32_t tri, double pad) {
    Box6 b;
    for (int k = 0; k < 3; ++k) { b.v[k] = lo[3 * (size_t)tri + k] - pad; b.v[3 + k] = hi[3 * (size_t)tri + k] + pad; }
    return b;
}
__device__ __forceinline__ void unite(Box6& a, const Box6& b) {
    for (int k = 0; k < 3; ++k) { a.v[k] = fmin(a.v[k], b.v[k]); a.v[3 + k] = fmax(a.v[3 + k], b.v[3 + k]); }
}
// A box another workgroup wrote before its __threadfence + atomic: read past this compute unit's L1.
__device__ __forceinline__ Box6 load_box_coherent(const double* p) {
    Box6 b;
    for (int k = 0; k < 6; ++k) {
        const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b.v[k] = __longlong_as_double((long long)bits);
    }
    return b;
}

// Bottom-up boxes.  One thread per sorted position; at every node the first arrival stops, the second (its sibling
// subtree is complete and visible: fence before the atomic) unites the two children and climbs on.  Every thread
// ends after at most tree-depth steps, whatever the others do.
__global__ void k_boxes(long long n, const int32_t* __restrict__ order, const double* __restrict__ lo, const double* __restrict__ hi, double pad,
                        const int32_t* __restrict__ left, const int32_t* __restrict__ right, const int32_t* __restrict__ parent_int,
                        const int32_t* __restrict__ parent_leaf, double* box, unsigned* arrived) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int32_t node = parent_leaf[p];
    int32_t from = ~(int32_t)p;  // the child this thread comes from
    Box6 mine = tri_box(lo, hi, order[p], pad);
    while (node >= 0) {
        if (atomicAdd(arrived + node, 1u) == 0u) return;  // the sibling's thread will do this node
        const int32_t l = left[node], r = right[node];
        const int32_t other = l == from ? r : l;
        Box6 sib = other < 0 ? tri_box(lo, hi, order[~other], pad) : load_box_coherent(box + 6 * (size_t)other);
        unite(mine, sib);
        for (int k = 0; k < 6; ++k) box[6 * (size_t)node + k] = mine.v[k];
        __threadfence();
        from = node;
        node = parent_int[node];
    }
}

constexpr int32_t EMPTY = INT32_MIN;
struct Slot {
    int32_t ref;    // >= 0: internal radix-tree node that stays inner (more than BVH_LEAF_MAX triangles); < 0: leaf
    int32_t first;  // leaf: first sorted position
    int32_t count;  // leaf: triangles
};
__device__ __forceinline__ Slot slot_of(int32_t child, const int32_t* __restrict__ first, const int32_t* __restrict__ last) {
    if (child < 0) return {-1, ~child, 1};
    const int32_t f = first[child], c = last[child] - f + 1;
    if (c <= BVH_LEAF_MAX) return {-1, f, c};
    return {child, f, c};
}
// The (at most four) slots of the BVH4 node made from radix-tree node r; returns their number.
__device__ __forceinline__ int slots_of(int32_t r, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                        const int32_t* __restrict__ first, const int32_t* __restrict__ last, Slot out[4]) {
    int ns = 0;
    const Slot a = slot_of(left[r], first, last), b = slot_of(right[r], first, last);
    if (a.ref >= 0) { out[ns++] = slot_of(left[a.ref], first, last); out[ns++] = slot_of(right[a.ref], first, last); } else out[ns++] = a;
    if (b.ref >= 0) { out[ns++] = slot_of(left[b.ref], first, last); out[ns++] = slot_of(right[b.ref], first, last); } else out[ns++] = b;
    if (ns == 3) {  // one of the two was a leaf: a grandchild may be opened too
        for (int s = 0; s < 3; ++s)
            if (out[s].ref >= 0) {
                const int32_t c = out[s].ref;
                for (int m = 2; m > s; --m) out[m + 1] = out[m];
                out[s] = slot_of(left[c], first, last);
                out[s + 1] = slot_of(right[c], first, last);
                ns = 4;
                break;
            }
    }
    return ns;
}

__global__ void k_collapse_count(int32_t n_frontier, const int32_t* __restrict__ frontier, const int32_t* __restrict__ left,
                                 const int32_t* __restrict__ right, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                 int32_t* __restrict__ inner_count) {
    const int32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_frontier) return;
    Slot s[4];
    const int ns = slots_of(frontier[e], left, right, first, last, s);
    int c = 0;
    for (int k = 0; k < ns; ++k) c += s[k].ref >= 0 ? 1 : 0;
    inner_count[e] = c;
}

__device__ __forceinline__ float round_down(double x) {
    float f = __double2float_rd(x);
    return f;
}
__device__ __forceinline__ float round_up(double x) {
    float f = __double2float_ru(x);
    return f;
}

// Writes the BVH4 nodes of one level (node e of the level is out[level_base + e]) and the next level's frontier.
__global__ void k_collapse_emit(int32_t n_frontier, const int32_t* __restrict__ frontier, const int32_t* __restrict__ left,
                                const int32_t* __restrict__ right, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                const int32_t* __restrict__ inner_offset, int32_t level_base, int32_t next_base, int32_t tri_base,
                                const int32_t* __restrict__ order, const double* __restrict__ lo, const double* __restrict__ hi, double pad,
                                const double* __restrict__ box, BvhNode* __restrict__ out, int32_t* __restrict__ next_frontier) {
    const int32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_frontier) return;
    Slot s[4];
    const int ns = slots_of(frontier[e], left, right, first, last, s);
    BvhNode nd;
    int inner = 0;
    for (int c = 0; c < 4; ++c) {
        nd._pad[c] = 0;
        if (c >= ns) {
            for (int k = 0; k < 3; ++k) { nd.lo[k][c] = INFINITY; nd.hi[k][c] = -INFINITY; }
            nd.child[c] = EMPTY;
            continue;
        }
        Box6 b;
        if (s[c].ref >= 0) {
            for (int k = 0; k < 6; ++k) b.v[k] = box[6 * (size_t)s[c].ref + k];
            const int32_t slot = inner_offset[e] + inner++;
            nd.child[c] = next_base + slot;
            next_frontier[slot] = s[c].ref;
        } else {
            b = tri_box(lo, hi, order[s[c].first], pad);
            for (int t = 1; t < s[c].count; ++t) unite(b, tri_box(lo, hi, order[s[c].first + t], pad));
            nd.child[c] = ~(((tri_base + s[c].first) << 2) | (s[c].count - 1));
        }
        for (int k = 0; k < 3; ++k) { nd.lo[k][c] = round_down(b.v[k]); nd.hi[k][c] = round_up(b.v[3 + k]); }
    }
    out[level_base + e] = nd;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
    template <class T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

bool lbvh_build(int device, size_t n, const double* tri_lo, const double* tri_hi, double pad, int32_t tri_base, int max_levels,
                std::vector<int32_t>& order, std::vector<BvhNode>& nodes, std::vector<int32_t>& level_start, double root_box[6]) {
    if (n <= size_t(BVH_LEAF_MAX) + 1 || n > (size_t(1) << 27) || size_t(tri_base) + n >= (size_t(1) << 29)) return false;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return false; }
    // centroid box on the host (one pass over data the host has in hand anyway)
    double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const double c = 0.5 * tri_lo[3 * i + size_t(k)] + 0.5 * tri_hi[3 * i + size_t(k)];
            if (!(c == c) || std::isinf(c)) return false;  // the host build copes with such meshes, the Morton grid does not
            clo[k] = std::min(clo[k], c);
            chi[k] = std::max(chi[k], c);
        }
    DevBuf lo, hi, key_in, key_out, idx_in, idx_out, tmp, l, r, f, la, pi, pl, box, arrived, fr_a, fr_b, cnt, off, out;
    const size_t ni = n - 1;
    if (!lo.alloc(n * 24) || !hi.alloc(n * 24) || !key_in.alloc(n * 8) || !key_out.alloc(n * 8) || !idx_in.alloc(n * 4) || !idx_out.alloc(n * 4) ||
        !l.alloc(ni * 4) || !r.alloc(ni * 4) || !f.alloc(ni * 4) || !la.alloc(ni * 4) || !pi.alloc(ni * 4) || !pl.alloc(n * 4) ||
        !box.alloc(ni * 48) || !arrived.alloc(ni * 4) || !fr_a.alloc(n * 4) || !fr_b.alloc(n * 4) || !cnt.alloc(n * 4) || !off.alloc(n * 4) ||
        !out.alloc(ni * sizeof(BvhNode))) { (void)hipGetLastError(); return false; }
    if (hipMemcpy(lo.p, tri_lo, n * 24, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(hi.p, tri_hi, n * 24, hipMemcpyHostToDevice) != hipSuccess) return false;
    if (hipMemset(arrived.p, 0, ni * 4) != hipSuccess || hipMemset(pi.p, 0xff, ni * 4) != hipSuccess) return false;
    double s[3];
    for (int k = 0; k < 3; ++k) s[k] = chi[k] > clo[k] ? 1. / (chi[k] - clo[k]) : 0.;
    const unsigned blocks = unsigned((n + 255) / 256);
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, 0, n, lo.as<double>(), hi.as<double>(), clo[0], clo[1], clo[2], s[0], s[1], s[2],
                       key_in.as<unsigned long long>(), idx_in.as<int32_t>());
    size_t sort_bytes = 0, scan_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, key_in.as<unsigned long long>(), key_out.as<unsigned long long>(),
                                           idx_in.as<int32_t>(), idx_out.as<int32_t>(), int(n), 0, 63) != hipSuccess) return false;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, cnt.as<int32_t>(), off.as<int32_t>(), int(n)) != hipSuccess) return false;
    if (!tmp.alloc(std::max(sort_bytes, scan_bytes))) return false;
    size_t tmp_bytes = std::max(sort_bytes, scan_bytes);
    if (hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key_in.as<unsigned long long>(), key_out.as<unsigned long long>(),
                                           idx_in.as<int32_t>(), idx_out.as<int32_t>(), int(n), 0, 63) != hipSuccess) return false;
    const int32_t* ord = idx_out.as<int32_t>();
    hipLaunchKernelGGL(k_radix_tree, dim3(blocks), dim3(256), 0, 0, (long long)n, key_out.as<unsigned long long>(), l.as<int32_t>(),
                       r.as<int32_t>(), f.as<int32_t>(), la.as<int32_t>());
    hipLaunchKernelGGL(k_parents, dim3(blocks), dim3(256), 0, 0, (long long)n, l.as<int32_t>(), r.as<int32_t>(), pi.as<int32_t>(), pl.as<int32_t>());
    hipLaunchKernelGGL(k_boxes, dim3(blocks), dim3(256), 0, 0, (long long)n, ord, lo.as<double>(), hi.as<double>(), pad, l.as<int32_t>(),
                       r.as<int32_t>(), pi.as<int32_t>(), pl.as<int32_t>(), box.as<double>(), arrived.as<unsigned>());
    if (hipGetLastError() != hipSuccess) return false;
    // ---- breadth-first collapse, one level per pass; the root (radix-tree node 0) covers more than BVH_LEAF_MAX triangles
    const int32_t zero = 0;
    if (hipMemcpy(fr_a.p, &zero, 4, hipMemcpyHostToDevice) != hipSuccess) return false;
    int32_t n_frontier = 1, level_base = 0;
    level_start.clear();
    int32_t* cur = fr_a.as<int32_t>();
    int32_t* nxt = fr_b.as<int32_t>();
    while (n_frontier > 0) {
        if (int(level_start.size()) >= max_levels) return false;  // deeper than the traversal stack allows: host build instead
        level_start.push_back(level_base);
        const unsigned fb = unsigned((n_frontier + 255) / 256);
        hipLaunchKernelGGL(k_collapse_count, dim3(fb), dim3(256), 0, 0, n_frontier, cur, l.as<int32_t>(), r.as<int32_t>(), f.as<int32_t>(),
                           la.as<int32_t>(), cnt.as<int32_t>());
        if (hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, cnt.as<int32_t>(), off.as<int32_t>(), n_frontier) != hipSuccess) return false;
        int32_t last_off = 0, last_cnt = 0;
        if (hipMemcpy(&last_off, off.as<int32_t>() + (n_frontier - 1), 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&last_cnt, cnt.as<int32_t>() + (n_frontier - 1), 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        const int32_t n_next = last_off + last_cnt, next_base = level_base + n_frontier;
        if (size_t(next_base) + size_t(n_next) > ni) return false;  // cannot happen: every inner BVH4 node is a distinct radix-tree node
        hipLaunchKernelGGL(k_collapse_emit, dim3(fb), dim3(256), 0, 0, n_frontier, cur, l.as<int32_t>(), r.as<int32_t>(), f.as<int32_t>(),
                           la.as<int32_t>(), off.as<int32_t>(), level_base, next_base, tri_base, ord, lo.as<double>(), hi.as<double>(), pad,
                           box.as<double>(), out.as<BvhNode>(), nxt);
        level_base = next_base;
        n_frontier = n_next;
        std::swap(cur, nxt);
    }
    level_start.push_back(level_base);  // end mark: nodes of level k are [level_start[k], level_start[k + 1])
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return false; }
    order.resize(n);
    nodes.resize(size_t(level_base));
    return hipMemcpy(order.data(), idx_out.p, n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(nodes.data(), out.p, size_t(level_base) * sizeof(BvhNode), hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(root_box, box.p, 48, hipMemcpyDeviceToHost) == hipSuccess;
}

}  // namespace rt
