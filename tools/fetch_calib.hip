// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports on gfx950 for the access shapes of the trace kernel: per-lane
// gathers of 8 and 16 bytes at scattered cache lines, against a coalesced streaming read of known size.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// every lane reads 16 B, consecutive lanes consecutive addresses: n16 loads in total
__global__ void k_stream16(const u4* __restrict__ buf, uint64_t n16, uint32_t* out) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const u4 v = buf[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
// every lane reads `bytes` (8 or 16) at byte offset `off` of a pseudo-random 128-byte line; `per_lane` loads per lane; `both` = also the other 64-byte half
template <int BYTES>
__global__ void k_gather(const char* __restrict__ buf, uint64_t n_lines, uint32_t per_lane, uint32_t off, int both, uint64_t salt, uint32_t* out) {
    uint32_t acc = 0;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const uint64_t line = mix(t * 0x9E3779B97F4A7C15ull + k + salt) % n_lines;
        const char* p = buf + line * 128 + off;
        if (BYTES == 16) { const u4 v = *(const u4*)p; acc += v.x ^ v.w; if (both) { const u4 w = *(const u4*)(p + 64); acc += w.y; } }
        else { const u2 v = *(const u2*)p; acc += v.x ^ v.y; if (both) { const u2 w = *(const u2*)(p + 64); acc += w.y; } }
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main() {
    const uint64_t bytes = 2ull << 30;  // 2 GiB: far beyond L2 (4 MB per XCD) and the 256 MB Infinity Cache
    char* buf; uint32_t* out;
    CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(buf, 1, bytes));
    const uint64_t n_lines = bytes / 128;
    const uint32_t blocks = 256 * 16, threads = 256, per_lane = 64;
    const uint64_t lanes = (uint64_t)blocks * threads, loads = lanes * per_lane;
    hipLaunchKernelGGL(k_stream16, dim3(blocks), dim3(threads), 0, 0, (const u4*)buf, (1ull << 30) / 16, out);
    CK(hipDeviceSynchronize());
    std::printf("k_stream16: 1073741824 bytes read (coalesced 16 B per lane)\n");
    hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(threads), 0, 0, buf, n_lines, per_lane, 0u, 0, 1ull, out);
    CK(hipDeviceSynchronize());
    std::printf("k_gather<16> off 0: %llu loads of 16 B = %llu bytes asked, %llu bytes of 128-B lines, %llu of 64-B halves\n", (unsigned long long)loads, (unsigned long long)loads * 16, (unsigned long long)loads * 128, (unsigned long long)loads * 64);
    hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(threads), 0, 0, buf, n_lines, per_lane, 64u, 0, 2ull, out);
    CK(hipDeviceSynchronize());
    std::printf("k_gather<16> off 64: same counts\n");
    hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(threads), 0, 0, buf, n_lines, per_lane, 0u, 1, 3ull, out);
    CK(hipDeviceSynchronize());
    std::printf("k_gather<16> both halves: %llu loads of 16 B on %llu lines\n", (unsigned long long)loads * 2, (unsigned long long)loads);
    hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(threads), 0, 0, buf, n_lines, per_lane, 0u, 0, 4ull, out);
    CK(hipDeviceSynchronize());
    std::printf("k_gather<8> off 0: %llu loads of 8 B = %llu bytes asked\n", (unsigned long long)loads, (unsigned long long)loads * 8);
    std::printf("loads per gather kernel: %llu\n", (unsigned long long)loads);
    CK(hipFree(buf)); CK(hipFree(out));
    return 0;
}
