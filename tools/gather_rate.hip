// gather_rate.hip -- how many scattered 64-byte requests per second the part serves, against the size of the array they
// fall in (inside the 32 MB of L2, inside the 256 MB Infinity Cache, beyond it): the ceiling config 5's trace kernel is
// compared with.  Every lane reads 16 bytes of a pseudo-random 128-byte line, 64 loads per lane,
// 16 waves per CU; the kernel is run once to warm the caches and then timed with HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o tools/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int DEP>
__global__ void k_gather(const char* __restrict__ buf, uint64_t n_lines, uint32_t per_lane, uint64_t salt, uint32_t* out) {
    uint32_t acc = 0;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k) {
        // DEP: the next address depends on the loaded value (a chain of round trips, like cell record -> triangle records)
        const uint64_t line = mix(t * 0x9E3779B97F4A7C15ull + k + salt + (DEP ? acc : 0u)) % n_lines;
        const u4 v = *(const u4*)(buf + line * 128);
        acc += v.x ^ v.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
int main() {
    const uint64_t max_bytes = 2ull << 30;
    char* buf; uint32_t* out;
    CK(hipMalloc((void**)&buf, max_bytes)); CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(buf, 1, max_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t blocks = 256 * 4, threads = 256, per_lane = 256;
    const double loads = (double)blocks * threads * per_lane;
    for (uint64_t mb : {8ull, 24ull, 64ull, 140ull, 200ull, 512ull, 2048ull}) {
        const uint64_t n_lines = (mb << 20) / 128;
        for (int dep = 0; dep < 2; ++dep) {
            float ms = 0.f;
            for (int rep = 0; rep < 3; ++rep) {  // (the first run warms the caches; the last is reported)
                CK(hipEventRecord(e0, 0));
                if (dep) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(threads), 0, 0, buf, n_lines, per_lane, 7ull + rep, out);
                else hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(threads), 0, 0, buf, n_lines, per_lane, 7ull + rep, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::printf("%5llu MB array, %s loads: %.2f ms, %.1f e9 loads/s = %.2f TB/s of 64-byte requests\n", (unsigned long long)mb,
                        dep ? "dependent  " : "independent", ms, loads / ms / 1e6, loads * 64 / ms / 1e9);
        }
    }
    CK(hipFree(buf)); CK(hipFree(out));
    return 0;
}
