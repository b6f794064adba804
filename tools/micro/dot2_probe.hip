// What v_dot2c_f32_bf16 returns for x . x and x . (1,1) on gfx950, next to the same sums by unpack + fma
// (hipcc --offload-arch=gfx950 -O3 tools/micro/dot2_probe.hip -o /tmp/dot2_probe && /tmp/dot2_probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__device__ __forceinline__ float dot2acc(unsigned a, unsigned b, float c) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
__global__ void k(const unsigned* a, float* o) {
    const unsigned v = a[threadIdx.x];
    const float lo = __builtin_bit_cast(float, v << 16), hi = __builtin_bit_cast(float, v & 0xFFFF0000u);
    o[threadIdx.x * 4 + 0] = dot2acc(v, 0x3F803F80u, 0.f);
    o[threadIdx.x * 4 + 1] = lo + hi;
    o[threadIdx.x * 4 + 2] = dot2acc(v, v, 0.f);
    o[threadIdx.x * 4 + 3] = lo * lo + hi * hi;
}
int main() {
    unsigned h[64];
    float* d_o; unsigned* d_a; float ho[256];
    for (int i = 0; i < 64; ++i) {
        const float x = -7.f + 0.37f * i, y = 3.f - 0.21f * i;
        unsigned ux, uy; memcpy(&ux, &x, 4); memcpy(&uy, &y, 4);
        h[i] = (ux >> 16) | (uy & 0xFFFF0000u);
    }
    hipMalloc(&d_a, sizeof(h)); hipMalloc(&d_o, sizeof(ho));
    hipMemcpy(d_a, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_a, d_o);
    hipMemcpy(ho, d_o, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("lane %d: dot(x,1) %.4f  lo+hi %.4f | dot(x,x) %.4f  lo^2+hi^2 %.4f\n", i, ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3]);
    return 0;
}
