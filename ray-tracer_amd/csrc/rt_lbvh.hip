// rt_lbvh.hip — device side of the optional GPU BVH build.
//
// For one triangle mesh: Morton codes of the triangle centroids, a radix sort, and the binary radix tree over the
// sorted codes (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012) are
// computed on the GPU; rt_program.cpp turns that topology into the BVH4 the traversal kernels use (boxes bottom-up,
// leaves of up to four triangles, 4-wide collapse, depth ordering), exactly as it does for its own SAH tree.  Like the
// SAH tree this is acceleration only: which triangles get tested, never what a test returns.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
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
__global__ void k_morton(size_t n, const double* __restrict__ c, double lx, double ly, double lz, double sx, double sy, double sz,
                         unsigned long long* __restrict__ key, int32_t* __restrict__ idx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto q = [](double v, double lo, double s) {
        double t = (v - lo) * s;  // [0, 1]
        if (!(t > 0.)) t = 0.;
        if (t > 1.) t = 1.;
        return (unsigned long long)(t * 2097151.0);
    };
    key[i] = (spread21(q(c[3 * i], lx, sx)) << 2) | (spread21(q(c[3 * i + 1], ly, sy)) << 1) | spread21(q(c[3 * i + 2], lz, sz));
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

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess; }
    template <class T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

bool lbvh_topology(int device, size_t n, const double* centroids, const double* box_lo, const double* box_hi,
                   std::vector<int32_t>& order, std::vector<int32_t>& left, std::vector<int32_t>& right,
                   std::vector<int32_t>& first, std::vector<int32_t>& last) {
    if (n < 2 || n > (size_t(1) << 30)) return false;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return false; }
    DevBuf c, key_in, key_out, idx_in, idx_out, tmp, l, r, f, la;
    if (!c.alloc(n * 3 * sizeof(double)) || !key_in.alloc(n * 8) || !key_out.alloc(n * 8) || !idx_in.alloc(n * 4) || !idx_out.alloc(n * 4) ||
        !l.alloc(n * 4) || !r.alloc(n * 4) || !f.alloc(n * 4) || !la.alloc(n * 4)) { (void)hipGetLastError(); return false; }
    if (hipMemcpy(c.p, centroids, n * 3 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return false;
    double s[3];
    for (int k = 0; k < 3; ++k) s[k] = box_hi[k] > box_lo[k] ? 1. / (box_hi[k] - box_lo[k]) : 0.;
    const unsigned blocks = unsigned((n + 255) / 256);
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, 0, n, c.as<double>(), box_lo[0], box_lo[1], box_lo[2], s[0], s[1], s[2],
                       key_in.as<unsigned long long>(), idx_in.as<int32_t>());
    size_t tmp_bytes = 0;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key_in.as<unsigned long long>(), key_out.as<unsigned long long>(),
                                           idx_in.as<int32_t>(), idx_out.as<int32_t>(), int(n), 0, 63) != hipSuccess) return false;
    if (!tmp.alloc(tmp_bytes)) return false;
    if (hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, key_in.as<unsigned long long>(), key_out.as<unsigned long long>(),
                                           idx_in.as<int32_t>(), idx_out.as<int32_t>(), int(n), 0, 63) != hipSuccess) return false;
    hipLaunchKernelGGL(k_radix_tree, dim3(blocks), dim3(256), 0, 0, (long long)n, key_out.as<unsigned long long>(), l.as<int32_t>(),
                       r.as<int32_t>(), f.as<int32_t>(), la.as<int32_t>());
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
    order.resize(n);
    left.resize(n - 1); right.resize(n - 1); first.resize(n - 1); last.resize(n - 1);
    return hipMemcpy(order.data(), idx_out.p, n * 4, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(left.data(), l.p, (n - 1) * 4, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(right.data(), r.p, (n - 1) * 4, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(first.data(), f.p, (n - 1) * 4, hipMemcpyDeviceToHost) == hipSuccess &&
           hipMemcpy(last.data(), la.p, (n - 1) * 4, hipMemcpyDeviceToHost) == hipSuccess;
}

}  // namespace rt
