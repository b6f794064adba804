// LLM-side multimodal splice -- SURVEY.md §8f row 4.
//
// Replaces the per-row Python loop of LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal
// (/root/reference/model/llava_walkgpt/model/llava_arch.py:265-518) for the case WalkGPT produces: every row of input_ids
// holds exactly one IMAGE_TOKEN_INDEX (-200) placeholder (utils/dataset.py collate; the reference asserts equal output
// lengths, :430-431).  Row r of the result is
//     embeds = [ embed_tokens(ids[:s]) , image_features[r] (T tokens) , embed_tokens(ids[s+1:]) ]            (:357-365, :404-408)
//     mask   = [ mask[:s]              , vit_attention_mask[r]        , mask[s+1:] ]                          (:361-366)
//     labels = [ labels[:s]            , IGNORE_INDEX x T             , labels[s+1:] ]                        (:367-378)
// plus WalkGPT's [SEG] read-out mask in spliced coordinates (model/walkgpt.py:293-306): position p is selected when the token
// that FOLLOWS it in the un-spliced row (index p - (T-1) + 1) is a [SEG] id -- the reference's fixed shift by T-1 = 255.
// One launch gathers everything: HBM-bound, (L + T - 1) * H * 2 bytes read and written per row.
#include "wg_common.h"

struct SpliceArgs {
    const long* ids; const bf16* table; const bf16* img; const unsigned char* mask_in; const unsigned char* vit_mask;
    const long* labels_in; const long* seg_ids;
    bf16* embeds; unsigned char* mask_out; long* labels_out; unsigned char* seg_mask;
    int* img_pos; int* img_cnt;
    int rows, L, T, H, V, nseg;
    long image_token, ignore_index;
};

// position and number of image placeholders per row
__global__ __launch_bounds__(256) void wg_splice_scan_kernel(SpliceArgs a) {
    __shared__ int smin[4], scnt[4];
    const int r = blockIdx.x;
    int mn = 0x7fffffff, cnt = 0;
    for (int i = threadIdx.x; i < a.L; i += 256)
        if (a.ids[(long)r * a.L + i] == a.image_token) { mn = min(mn, i); ++cnt; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); cnt += __shfl_xor(cnt, o, 64); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = mn; scnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.img_pos[r] = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
        a.img_cnt[r] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
    }
}

// one workgroup per output token: 16 bytes per lane of its embedding row + the scalar outputs
__global__ __launch_bounds__(256) void wg_splice_gather_kernel(SpliceArgs a) {
    const int Lo = a.L + a.T - 1;
    const int r = blockIdx.x / Lo, j = blockIdx.x % Lo;
    const int s = a.img_cnt[r] == 1 ? a.img_pos[r] : 0;   // (malformed rows are reported by the host from img_cnt)
    const bool is_img = j >= s && j < s + a.T;
    const int i = j < s ? j : j - a.T + 1;                 // source text position when !is_img
    const bf16* src;
    if (is_img) {
        src = a.img + ((long)r * a.T + (j - s)) * a.H;
    } else {
        long id = a.ids[(long)r * a.L + i];
        id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);      // out-of-vocabulary ids cannot fault; the host validates them
        src = a.table + id * a.H;
    }
    bf16* dst = a.embeds + ((long)r * Lo + j) * a.H;
    for (int d = threadIdx.x * 8; d < a.H; d += 2048) *(bf16x8*)(dst + d) = *(const bf16x8*)(src + d);
    if (threadIdx.x == 0) {
        const long o = (long)r * Lo + j;
        if (a.mask_out) a.mask_out[o] = is_img ? (a.vit_mask ? a.vit_mask[(long)r * a.T + (j - s)] : 1) : (a.mask_in ? a.mask_in[(long)r * a.L + i] : 1);
        if (a.labels_out) a.labels_out[o] = is_img ? a.ignore_index : a.labels_in[(long)r * a.L + i];
        if (a.seg_mask) {
            const int nxt = j - (a.T - 1) + 1;             // un-spliced index of the following token under the fixed shift
            bool hit = false;
            if (j >= a.T - 1 && nxt < a.L) {
                const long id = a.ids[(long)r * a.L + nxt];
                for (int k = 0; k < a.nseg; ++k) hit = hit || id == a.seg_ids[k];
            }
            a.seg_mask[o] = hit ? 1 : 0;
        }
    }
}

// ids [rows, L] int64; table [V, H] bf16; image_features [rows, T, H] bf16; masks bool (1 byte); labels int64 (optional);
// seg_ids [nseg] int64 on the device (optional).  Outputs: embeds [rows, L+T-1, H], mask_out / labels_out / seg_mask
// [rows, L+T-1] (each optional), img_pos / img_cnt [rows] int32 (the caller checks img_cnt == 1 for every row).
extern "C" int wg_splice_multimodal_bf16(const long* ids, const void* table, const void* image_features, const void* mask_in,
                                         const void* vit_mask, const long* labels_in, const long* seg_ids, int nseg, void* embeds,
                                         void* mask_out, long* labels_out, void* seg_mask, int* img_pos, int* img_cnt, int rows, int L,
                                         int T, int H, int V, long image_token, long ignore_index, void* stream) {
    WG_REQUIRE(ids && table && image_features && embeds && img_pos && img_cnt, "splice: null operand");
    WG_REQUIRE(rows > 0 && L > 0 && T > 0 && V > 0 && H > 0 && H % 8 == 0, "splice: bad shape (H must be a multiple of 8)");
    WG_REQUIRE(!labels_out || labels_in, "splice: labels_out without labels_in");
    WG_REQUIRE(!seg_mask || (seg_ids && nseg > 0), "splice: seg_mask without seg ids");
    WG_REQUIRE((((uintptr_t)table | (uintptr_t)image_features | (uintptr_t)embeds) & 15) == 0, "splice: misaligned operand");
    SpliceArgs a{};
    a.ids = ids; a.table = (const bf16*)table; a.img = (const bf16*)image_features; a.mask_in = (const unsigned char*)mask_in;
    a.vit_mask = (const unsigned char*)vit_mask; a.labels_in = labels_in; a.seg_ids = seg_ids; a.embeds = (bf16*)embeds;
    a.mask_out = (unsigned char*)mask_out; a.labels_out = labels_out; a.seg_mask = (unsigned char*)seg_mask; a.img_pos = img_pos;
    a.img_cnt = img_cnt; a.rows = rows; a.L = L; a.T = T; a.H = H; a.V = V; a.nseg = nseg; a.image_token = image_token;
    a.ignore_index = ignore_index;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wg_splice_scan_kernel, dim3(rows), dim3(256), 0, st, a);
    hipLaunchKernelGGL(wg_splice_gather_kernel, dim3((unsigned)((long)rows * (L + T - 1))), dim3(256), 0, st, a);
    return wg_check_launch("wg_splice_multimodal_bf16");
}
