// Micro-benchmark: how fast ONE workgroup (8 waves) pulls a 128 KB weight matrix through its compute unit, for the two access patterns a
// 16x16x32 MFMA B operand can be fetched with: (a) row-major W [256][256] bf16 -- a wave instruction touches 16 rows x 64 B; (b) the same data
// pre-tiled in fragment order -- a wave instruction reads 1 KB contiguous.  Reported cold (a 512 MB sweep in between) and warm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(512) void ingest(const u32x4* w, int tiled, int reps, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, kg = lane >> 4;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        const u32x4* base = w + (size_t)r * 8192;            // 128 KB per matrix = 8192 x 16 B
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int nb = wave + 8 * u;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int idx = tiled ? ((nb * 8 + ks) * 64 + lane) : ((nb * 16 + l16) * 32 + ks * 4 + kg);
                const u32x4 v = base[idx];
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *cyc = t1 - t0;
    out[threadIdx.x] = (float)(acc.x ^ acc.y ^ acc.z ^ acc.w);
}
__global__ void sweep(float* p, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.f; }
int main() {
    const int reps = 8;                                       // 8 different matrices = 1 MB, read once each
    u32x4* w; float* out; unsigned long long* cyc; float* big;
    hipMalloc(&w, (size_t)reps * 131072); hipMalloc(&out, 2048); hipMalloc(&cyc, 8); hipMalloc(&big, 512u << 20);
    hipMemset(w, 1, (size_t)reps * 131072); hipMemset(big, 0, 512u << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int tiled = 0; tiled < 2; ++tiled)
        for (int cold = 1; cold >= 0; --cold) {
            float best = 1e9f;
            for (int it = 0; it < 5; ++it) {
                if (cold) sweep<<<2048, 256>>>(big, (512u << 20) / 4);
                hipEventRecord(e0);
                ingest<<<1, 512>>>(w, tiled, reps, out, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s %s: %.1f us for %d x 128 KB  (%.1f GB/s incl. launch; in-kernel %llu cycles)\n", tiled ? "tiled    " : "row-major", cold ? "cold" : "warm",
                   best * 1e3, reps, reps * 131072 / (best * 1e-3) / 1e9, c);
        }
    return 0;
}
