// grid.hip -- Model::create_grid (model.rs:147-201) on the device: the triangle -> cell binning with the
// reference's Box3::include (box3D.rs:31-45, quirk Q9) and saturating cell casts (model.rs:159-167, Q10), the inclusive
// offset_array and the triangle_indices lists in the reference's order (every list ascends by triangle index, model.rs:152,179).
//
// No atomics and no sort: the output order is fixed by construction.  Work is cut into (block of 256 cells) x (segment of
// the triangle range) tiles; a tile counts, per cell, the triangles of its segment whose cell range holds the cell; one scan
// over (cell-major, segment-minor) turns the counts into positions; the same sweep then writes the triangle indices, each
// thread walking its segment in triangle order.  Cost n^3 * n_tris range tests per sweep (1e6 triangles, 32^3 cells: ~10 ms),
// which is why the builder stops at 128 cells per side; the reference fixes 32 (core.rs:62).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>

#include "../../include/rt_mi355x.h"
#include "rtm_internal.h"

#pragma clang fp contract(off)

#define HIP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e__ = (expr);                                                                                   \
        if (e__ != hipSuccess)                                                                                     \
            return rtm_internal_fail(RTM_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));          \
    } while (0)

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 1024;        // triangle ranges staged in LDS at a time
constexpr uint32_t kMaxCells = 128;

struct GridArgs {
    const double* pos[9];  // v0x v0y v0z v1x ... v2z (device), Model::get_triangle evaluated once per triangle
    double bmin[3], cell[3];
    uint32_t n, n_tris, n_seg, seg_len;
    uint64_t n3;
};

// Rust `as usize` on f64 (saturating, NaN -> 0) followed by .min(n - 1)   (model.rs:160-167)
__device__ __forceinline__ uint32_t cell_of(double v, uint32_t n) {
    if (!(v > 0.0)) return 0u;
    if (v >= (double)n) return n - 1u;
    return (uint32_t)v;
}

// one triangle: Box3::from_single_point(v0), include(v1), include(v2) per axis, then the two corner cells (model.rs:152-167)
__global__ __launch_bounds__(kBlock) void grid_range_kernel(GridArgs A, unsigned long long* __restrict__ ranges) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= A.n_tris) return;
    unsigned long long packed = 0ull;
    for (int c = 0; c < 3; ++c) {
        double ctr = A.pos[c][t], h = 0.0;  // box3D.rs:17-22
        for (int k = 1; k < 3; ++k) {
            const double d = A.pos[k * 3 + c][t] - ctr;  // box3D.rs:32
            if (fabs(d) > h) {                            // box3D.rs:33-36 (and :37-44 for y, z)
                h = (fabs(d) + h) / 2.0;
                ctr += (d - h * (__builtin_signbit(d) ? -1.0 : 1.0)) / 2.0;  // the ALREADY updated half extent: quirk Q9
            }
        }
        const uint32_t a = cell_of(((ctr - h) - A.bmin[c]) / A.cell[c], A.n);
        const uint32_t b = cell_of(((ctr + h) - A.bmin[c]) / A.cell[c], A.n);
        packed |= (unsigned long long)a << (10 * c) | (unsigned long long)b << (30 + 10 * c);
    }
    ranges[t] = packed;
}

// FILL = false: counts[cell * n_seg + seg] = triangles of the segment whose range holds the cell.
// FILL = true:  the same walk writes their indices from position start[cell * n_seg + seg] on.
template <bool FILL>
__global__ __launch_bounds__(kBlock) void grid_sweep_kernel(GridArgs A, const unsigned long long* __restrict__ ranges,
                                                            uint32_t* __restrict__ counts, const uint32_t* __restrict__ start,
                                                            uint32_t* __restrict__ tri_idx) {
    __shared__ unsigned long long tile[kTile];
    const uint64_t cell = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t seg = blockIdx.y;
    const bool live = cell < A.n3;
    const uint32_t cx = (uint32_t)(cell % A.n), cy = (uint32_t)((cell / A.n) % A.n), cz = (uint32_t)(cell / ((uint64_t)A.n * A.n));
    const uint32_t t_begin = seg * A.seg_len, t_end = min(A.n_tris, t_begin + A.seg_len);
    uint32_t n_mine = 0;
    const uint32_t at = (FILL && live) ? start[cell * A.n_seg + seg] : 0u;
    for (uint32_t t0 = t_begin; t0 < t_end; t0 += kTile) {
        const uint32_t n_here = min((uint32_t)kTile, t_end - t0);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n_here; i += kBlock) tile[i] = ranges[t0 + i];
        __syncthreads();
        if (!live) continue;
        for (uint32_t i = 0; i < n_here; ++i) {  // every thread reads the same word: an LDS broadcast
            const unsigned long long r = tile[i];
            const uint32_t lo = (uint32_t)r & 0x3FFFFFFFu, hi = (uint32_t)(r >> 30) & 0x3FFFFFFFu;
            const uint32_t x0 = lo & 1023u, y0 = (lo >> 10) & 1023u, z0 = lo >> 20, x1 = hi & 1023u, y1 = (hi >> 10) & 1023u, z1 = hi >> 20;
            // (the reference's loops `for ix in min_ix..=max_ix` are empty when min > max: model.rs:168-170)
            const bool in = cx >= x0 && cx <= x1 && cy >= y0 && cy <= y1 && cz >= z0 && cz <= z1;
            if (in) {
                if (FILL) tri_idx[at + n_mine] = t0 + i;
                ++n_mine;
            }
        }
    }
    if (!FILL && live) counts[cell * A.n_seg + seg] = n_mine;
}

// out[i] = in[0] + ... + in[i - 1] for i = 0..n (n + 1 values), one workgroup walking the array; *overflow is set when the
// total does not fit 32 bits (the C ABI's offsets are u32)
__global__ __launch_bounds__(1024) void grid_scan_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                                         uint32_t* __restrict__ overflow) {
    __shared__ uint32_t part[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long carry = 0ull;
    for (uint64_t base = 0; base < n; base += 1024u) {
        const uint32_t v = base + tid < n ? in[base + tid] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= (uint32_t)off) incl += up;
        }
        __syncthreads();
        if (lane == 63u) part[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 16u; ++w) {
            if (w < wave) before += part[w];
            total += part[w];
        }
        if (base + tid < n) out[base + tid] = (uint32_t)carry + before + incl - v;
        carry += total;
    }
    if (tid == 0) {
        out[n] = (uint32_t)carry;
        *overflow = carry > 0xFFFFFFFFull ? 1u : 0u;
    }
}

__global__ void grid_offsets_kernel(const uint32_t* __restrict__ start, uint32_t n_seg, uint64_t n3, uint32_t* __restrict__ offsets) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n3) offsets[c] = start[(c + 1) * n_seg];  // inclusive prefix (model.rs:185-193)
}

}  // namespace

extern "C" {

int rtm_grid_build_device(const double* const d_pos[9], uint32_t n_tris, const double bbox_center[3], const double bbox_half[3],
                          uint32_t cells_per_side, uint32_t* d_offsets, uint32_t* d_tri_idx, uint64_t tri_idx_capacity,
                          uint64_t* out_n_tri_idx, void* stream_) {
    if (!d_pos || !bbox_center || !bbox_half || !d_offsets || !out_n_tri_idx) return rtm_internal_fail(RTM_ERR_INVALID, "null argument");
    *out_n_tri_idx = 0;
    if (cells_per_side == 0 || cells_per_side > kMaxCells)
        return rtm_internal_fail(RTM_ERR_INVALID, "rtm_grid_build_device: cells_per_side must be 1..128 (the reference uses 32)");
    for (int a = 0; a < 9; ++a)
        if (n_tris && !d_pos[a]) return rtm_internal_fail(RTM_ERR_INVALID, "triangle arrays missing");
    if (rtm_internal_device() < 0) return rtm_internal_fail(RTM_ERR_STATE, "rtm_init has not been called");
    hipStream_t stream = (hipStream_t)stream_;
    GridArgs A{};
    const double n = (double)cells_per_side;
    for (int a = 0; a < 9; ++a) A.pos[a] = d_pos[a];
    for (int c = 0; c < 3; ++c) {  // the same IEEE expressions as the host front end (box3D.rs:55-61, model.rs:151)
        const double bmin = bbox_center[c] - bbox_half[c], bmax = bbox_center[c] + bbox_half[c];
        A.bmin[c] = bmin;
        A.cell[c] = (bmax - bmin) / n;
    }
    A.n = cells_per_side;
    A.n_tris = n_tris;
    A.n3 = (uint64_t)cells_per_side * cells_per_side * cells_per_side;
    const uint32_t cell_blocks = (uint32_t)((A.n3 + kBlock - 1) / kBlock);
    uint32_t n_seg = std::max<uint32_t>(1u, std::min<uint32_t>(64u, (2048u + cell_blocks - 1) / cell_blocks));
    n_seg = std::max<uint32_t>(1u, std::min<uint32_t>(n_seg, (n_tris + kTile - 1) / kTile));
    while (n_seg > 1 && A.n3 * n_seg > (4ull << 20)) --n_seg;
    A.n_seg = n_seg;
    A.seg_len = std::max<uint32_t>(1u, (n_tris + n_seg - 1) / n_seg);
    const uint64_t n_counts = A.n3 * n_seg;
    unsigned long long* d_ranges = nullptr;
    uint32_t *d_counts = nullptr, *d_start = nullptr, *d_flag = nullptr;
    int rc = RTM_OK;
    auto dev_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == RTM_OK) rc = rtm_internal_fail(RTM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
        return rc == RTM_OK;
    };
    dev_ok(hipMalloc((void**)&d_ranges, std::max<size_t>(1, n_tris) * 8), "hipMalloc ranges");
    if (rc == RTM_OK) dev_ok(hipMalloc((void**)&d_counts, n_counts * 4), "hipMalloc counts");
    if (rc == RTM_OK) dev_ok(hipMalloc((void**)&d_start, (n_counts + 1) * 4 + 4), "hipMalloc start");
    d_flag = d_start ? d_start + n_counts + 1 : nullptr;
    uint32_t h_tail[2] = {0, 0};  // total, overflow flag
    if (rc == RTM_OK) {
        if (n_tris) hipLaunchKernelGGL(grid_range_kernel, dim3((n_tris + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, A, d_ranges);
        hipLaunchKernelGGL(grid_sweep_kernel<false>, dim3(cell_blocks, n_seg), dim3(kBlock), 0, stream, A, d_ranges, d_counts, nullptr, nullptr);
        hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, stream, d_counts, d_start, n_counts, d_flag);
        hipLaunchKernelGGL(grid_offsets_kernel, dim3((unsigned)((A.n3 + 255) / 256)), dim3(256), 0, stream, d_start, n_seg, A.n3, d_offsets);
        dev_ok(hipGetLastError(), "grid count launches");
        if (rc == RTM_OK) dev_ok(hipMemcpyAsync(h_tail, d_start + n_counts, 8, hipMemcpyDeviceToHost, stream), "copy total");
        if (rc == RTM_OK) dev_ok(hipStreamSynchronize(stream), "grid count");
    }
    if (rc == RTM_OK) {
        *out_n_tri_idx = h_tail[0];
        if (h_tail[1]) rc = rtm_internal_fail(RTM_ERR_INVALID, "rtm_grid_build_device: more than 2^32 - 1 list entries");
        else if (h_tail[0] > tri_idx_capacity || (h_tail[0] && !d_tri_idx))
            rc = rtm_internal_fail(RTM_ERR_INVALID, "rtm_grid_build_device: tri_idx capacity too small, " + std::to_string(h_tail[0]) + " entries needed (returned in out_n_tri_idx)");
    }
    if (rc == RTM_OK && h_tail[0]) {
        hipLaunchKernelGGL(grid_sweep_kernel<true>, dim3(cell_blocks, n_seg), dim3(kBlock), 0, stream, A, d_ranges, nullptr, d_start, d_tri_idx);
        dev_ok(hipGetLastError(), "grid fill launch");
        if (rc == RTM_OK) dev_ok(hipStreamSynchronize(stream), "grid fill");
    }
    (void)hipFree(d_ranges);
    (void)hipFree(d_counts);
    (void)hipFree(d_start);
    return rc;
}

int rtm_grid_build(const double* const pos[9], uint32_t n_tris, const double bbox_center[3], const double bbox_half[3],
                   uint32_t cells_per_side, uint32_t* out_offsets, uint32_t* out_tri_idx, uint64_t tri_idx_capacity, uint64_t* out_n_tri_idx) {
    if (!pos || !bbox_center || !bbox_half || !out_offsets || !out_n_tri_idx) return rtm_internal_fail(RTM_ERR_INVALID, "null argument");
    *out_n_tri_idx = 0;
    if (cells_per_side == 0 || cells_per_side > kMaxCells) return rtm_internal_fail(RTM_ERR_INVALID, "rtm_grid_build: cells_per_side must be 1..128");
    const uint64_t n3 = (uint64_t)cells_per_side * cells_per_side * cells_per_side;
    double* d_pos_all = nullptr;
    uint32_t *d_off = nullptr, *d_idx = nullptr;
    int rc = RTM_OK;
    auto dev_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == RTM_OK) rc = rtm_internal_fail(RTM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
        return rc == RTM_OK;
    };
    dev_ok(hipMalloc((void**)&d_pos_all, std::max<size_t>(1, (size_t)n_tris) * 9 * 8), "hipMalloc positions");
    if (rc == RTM_OK) dev_ok(hipMalloc((void**)&d_off, n3 * 4), "hipMalloc offsets");
    if (rc == RTM_OK && tri_idx_capacity) dev_ok(hipMalloc((void**)&d_idx, tri_idx_capacity * 4), "hipMalloc tri_idx");
    const double* d_pos[9];
    for (int a = 0; a < 9; ++a) {
        d_pos[a] = d_pos_all ? d_pos_all + (size_t)a * n_tris : nullptr;
        if (rc == RTM_OK && n_tris) {
            if (!pos[a]) rc = rtm_internal_fail(RTM_ERR_INVALID, "triangle arrays missing");
            else dev_ok(hipMemcpy(d_pos_all + (size_t)a * n_tris, pos[a], (size_t)n_tris * 8, hipMemcpyHostToDevice), "copy positions");
        }
    }
    if (rc == RTM_OK)
        rc = rtm_grid_build_device(d_pos, n_tris, bbox_center, bbox_half, cells_per_side, d_off, out_tri_idx ? d_idx : nullptr,
                                   out_tri_idx ? tri_idx_capacity : 0, out_n_tri_idx, nullptr);
    // (too small a capacity still returns the offsets and the number of entries needed)
    const bool have_offsets = rc == RTM_OK || (*out_n_tri_idx > 0 && rc == RTM_ERR_INVALID);
    if (have_offsets && d_off && hipMemcpy(out_offsets, d_off, n3 * 4, hipMemcpyDeviceToHost) != hipSuccess && rc == RTM_OK)
        rc = rtm_internal_fail(RTM_ERR_DEVICE, "copy offsets");
    if (rc == RTM_OK && out_tri_idx && *out_n_tri_idx) dev_ok(hipMemcpy(out_tri_idx, d_idx, *out_n_tri_idx * 4, hipMemcpyDeviceToHost), "copy tri_idx");
    (void)hipFree(d_pos_all);
    (void)hipFree(d_off);
    (void)hipFree(d_idx);
    return rc;
}

}  // extern "C"
