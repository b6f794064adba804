// Probe: is the SGPR offset of a raw buffer load part of the hardware range check on gfx950?  (It decides whether a GEMM may keep its per-lane
// offsets constant and move tile / slab displacement through soffset without losing the out-of-range protection of ragged last tiles.)
// A 64 KiB allocation filled with 1.0f; descriptor extent = the first 4 KiB only.  Four loads per lane, plain and through LDS-DMA:
//   (a) voffset inside, soffset 0            -> 1.0
//   (b) voffset beyond the extent, soffset 0 -> 0.0 (the documented check)
//   (c) voffset inside, soffset pushes the address beyond the extent -> 0.0 if soffset is checked, 1.0 if it is not
//   (d) voffset + soffset inside             -> 1.0
// hipcc --offload-arch=gfx950 -O3 tools/micro/buffer_soffset_range.hip -o tools/micro/_bin/buffer_soffset_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* a, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 4096, 0x00020000);
    const int l = threadIdx.x;
    float r[4];
    r[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l * 4, 0, 0));
    r[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 8192 + l * 4, 0, 0));
    r[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l * 4, 8192, 0));
    r[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, l * 4, 1024, 0));
    for (int k = 0; k < 4; ++k) out[k * 64 + l] = r[k];
    // the same four through LDS-DMA (16 bytes per lane); LDS pre-filled with -7 so that "nothing written" shows
    float* s = (float*)smem;
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 4; ++e) s[k * 256 + l * 4 + e] = -7.f;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem), 16, l * 16, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 1024), 16, 8192 + l * 16, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 2048), 16, l * 16, 8192, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 3072), 16, l * 16, 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < 4; ++k) out[256 + k * 64 + l] = s[k * 256 + l * 4];
}
int main() {
    float *a, *o;
    hipMalloc(&a, 65536);
    hipMalloc(&o, 512 * 4);
    std::vector<float> h(16384, 1.0f), r(512);
    hipMemcpy(a, h.data(), 65536, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, a, o);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    const char* nm[4] = {"(a) inside", "(b) voffset beyond", "(c) soffset beyond", "(d) voffset+soffset inside"};
    for (int k = 0; k < 4; ++k) printf("plain   %-28s lane0 %.1f lane63 %.1f\n", nm[k], r[k * 64], r[k * 64 + 63]);
    for (int k = 0; k < 4; ++k) printf("lds-dma %-28s lane0 %.1f lane63 %.1f\n", nm[k], r[256 + k * 64], r[256 + k * 64 + 63]);
    return 0;
}
