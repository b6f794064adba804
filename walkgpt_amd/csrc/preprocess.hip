// Input pipeline on the GPU -- SURVEY.md §8f row 3.
//
// Replaces, for a batch of uint8 RGB frames already in HBM, the per-image CPU stage of the datasets
// (/root/reference/utils/PAVE_dataset.py:115-121, 217-236):
//     image_np = ResizeLongestSide(S).apply_image(image_rgb)          segment_anything/utils/transforms.py:27-36
//              = np.array(PIL_image.resize((new_w, new_h), BILINEAR))   (torchvision's resize of a PIL image)
//     image    = pad((float(image_np) - pixel_mean) / pixel_std, to S x S)   PAVE_dataset.py:115-121
// Pillow's resize is an antialiasing two-pass convolution in 8-bit fixed point (Resample.c: coefficients scaled by 2^22,
// a rounded uint8 image between the horizontal and the vertical pass).  The two kernels below reproduce it bit for bit:
// the coefficient tables -- a function of (input size, output size) only, computed in float64 exactly as precompute_coeffs /
// normalize_coeffs_8bpc do -- come from the host (walkgpt_amd/preprocess.py caches them), the passes run here, and the second
// pass fuses the normalisation, the zero padding and the NCHW / bf16 conversion.  HBM-bound: reads 3 B, writes 6 B per pixel.
// (x - mean) / std has only 3 x 256 possible values: they come as a table computed by the caller in IEEE fp32 -- this
// library is built with -ffast-math, under which a division here would be a reciprocal multiply, 1 ulp off torch's.
#include "wg_common.h"

#define WG_PIL_PRECISION_BITS 22   // 32 - 8 - 2 (Resample.c)

__device__ __forceinline__ unsigned char wg_clip8(int v) {   // clip8(): (v >> PRECISION_BITS) clamped to [0, 255]
    v >>= WG_PIL_PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: in [B, H, W, 3] -> out [B, H, Wo, 3]; bounds [Wo, 2] = (xmin, count), kk [Wo, ksize]
__global__ __launch_bounds__(256) void wg_pil_resize_h_kernel(const unsigned char* in, unsigned char* out, const int* bounds,
                                                              const int* kk, int ksize, int H, int W, int Wo, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // (b, y, xo)
    if (idx >= total) return;
    const int xo = (int)(idx % Wo);
    const long row = idx / Wo;   // b * H + y
    const int xmin = bounds[2 * xo], cnt = bounds[2 * xo + 1];
    const int* k = kk + (long)xo * ksize;
    const unsigned char* p = in + (row * W + xmin) * 3;
    int s0 = 1 << (WG_PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        const int c = k[x];
        s0 += p[3 * x] * c; s1 += p[3 * x + 1] * c; s2 += p[3 * x + 2] * c;
    }
    unsigned char* o = out + idx * 3;
    o[0] = wg_clip8(s0); o[1] = wg_clip8(s1); o[2] = wg_clip8(s2);
}

// vertical pass + normalise + pad: in [B, H, Wo, 3] -> resized [B, Ho, Wo, 3] (optional) and out [B, 3, S, S] (bf16 or fp32)
template <typename OutT>
__global__ __launch_bounds__(256) void wg_pil_resize_v_norm_kernel(const unsigned char* in, unsigned char* resized, OutT* out,
                                                                   const int* bounds, const int* kk, int ksize, int H, int Wo,
                                                                   int Ho, int S, const float* lut, int do_v, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // (b, yo, xo) over the padded S x S square
    if (idx >= total) return;
    const int xo = (int)(idx % S);
    const int yo = (int)((idx / S) % S);
    const long b = idx / ((long)S * S);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;   // padding is applied after the normalisation: zeros
    if (yo < Ho && xo < Wo) {
        unsigned char r0, r1, r2;
        if (do_v) {
            const int ymin = bounds[2 * yo], cnt = bounds[2 * yo + 1];
            const int* k = kk + (long)yo * ksize;
            int s0 = 1 << (WG_PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
            for (int y = 0; y < cnt; ++y) {
                const unsigned char* p = in + ((b * H + ymin + y) * Wo + xo) * 3;
                const int c = k[y];
                s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
            }
            r0 = wg_clip8(s0); r1 = wg_clip8(s1); r2 = wg_clip8(s2);
        } else {
            const unsigned char* p = in + ((b * H + yo) * Wo + xo) * 3;
            r0 = p[0]; r1 = p[1]; r2 = p[2];
        }
        if (resized) {
            unsigned char* q = resized + ((b * Ho + yo) * Wo + xo) * 3;
            q[0] = r0; q[1] = r1; q[2] = r2;
        }
        v0 = lut[r0]; v1 = lut[256 + r1]; v2 = lut[512 + r2];   // (x - mean_c) / std_c
    }
    const long plane = (long)S * S;
    OutT* o = out + b * 3 * plane + (long)yo * S + xo;
    o[0] = (OutT)v0; o[plane] = (OutT)v1; o[2 * plane] = (OutT)v2;
}

// frames [B, H, W, 3] uint8 -> images [B, 3, S, S] (bf16 if out_bf16 else fp32), optionally the resized uint8 frames
// [B, Ho, Wo, 3].  h_bounds/h_kk (Wo entries) and v_bounds/v_kk (Ho entries) are Pillow's coefficient tables; a null table
// means that pass is the identity (Wo == W / Ho == H).  tmp: B*H*Wo*3 bytes of scratch (unused without a horizontal pass).
// norm_lut: device fp32 [3][256], norm_lut[c][v] = (v - mean_c) / std_c.
extern "C" int wg_preprocess_frames_u8(const void* frames, void* tmp, void* resized, void* out, int out_bf16, const int* h_bounds,
                                       const int* h_kk, int h_ksize, const int* v_bounds, const int* v_kk, int v_ksize, int B, int H,
                                       int W, int Ho, int Wo, int S, const float* norm_lut, void* stream) {
    WG_REQUIRE(frames && out && norm_lut, "preprocess: null operand");
    WG_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Ho <= S && Wo <= S, "preprocess: bad shape");
    WG_REQUIRE((h_bounds && h_kk && tmp) || Wo == W, "preprocess: horizontal tables / scratch missing");
    WG_REQUIRE((v_bounds && v_kk) || Ho == H, "preprocess: vertical tables missing");
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* src = (const unsigned char*)frames;
    if (h_bounds) {
        const long total = (long)B * H * Wo;
        hipLaunchKernelGGL(wg_pil_resize_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, (unsigned char*)tmp,
                           h_bounds, h_kk, h_ksize, H, W, Wo, total);
        src = (const unsigned char*)tmp;
    }
    const long total = (long)B * S * S;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (out_bf16)
        hipLaunchKernelGGL(wg_pil_resize_v_norm_kernel<bf16>, grid, dim3(256), 0, st, src, (unsigned char*)resized, (bf16*)out, v_bounds, v_kk,
                           v_ksize, H, Wo, Ho, S, norm_lut, v_bounds ? 1 : 0, total);
    else
        hipLaunchKernelGGL(wg_pil_resize_v_norm_kernel<float>, grid, dim3(256), 0, st, src, (unsigned char*)resized, (float*)out, v_bounds, v_kk,
                           v_ksize, H, Wo, Ho, S, norm_lut, v_bounds ? 1 : 0, total);
    return wg_check_launch("wg_preprocess_frames_u8");
}
