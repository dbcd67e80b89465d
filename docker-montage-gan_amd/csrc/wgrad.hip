// wgrad.hip -- weight gradient of the 64-channel, full-resolution 3x3 convolutions on the matrix cores, with the modulation
// and its gradient folded in:
//
//     H_n[co][tap][ci] = sum_{pixels p of sample n} dy[n,p,co] * x[n,p+tap,ci]              (zero outside the image)
//     dW[co][tap][ci]  = sum_n T(s[n,ci]) * H_n[co][tap][ci]                                (s = NULL: 1)
//     ds[n][ci]        = sum_{co,tap} w[co][tap][ci] * H_n[co][tap][ci]                     (optional)
//
// (reference training/networks.py:62-88: the shared-weight convolution of modulated_conv2d reads x * s, so its weight gradient is
// the first line on x * s -- the second line -- and the gradient of the modulation `x * s` w.r.t. s is sum_p dxs * x, which is the
// third line.)  The library computes the first line with a split-K kernel over a zero-filled float32 workspace (395 us at
// [32,64,256,256], HBM bound ~105) and needs x * s materialised; the third line costs another pass over x and dxs.  Here dy and x are
// each read ONCE -- 536 MB at batch 32 -- and both results fall out of the same accumulators.
//
// Both MFMA operands are pixel-major in memory (NHWC: a pixel's 64 channels are contiguous) while the contraction runs over PIXELS:
// A[row = co][k = pixel] and B[k = pixel][col = ci] need 8 consecutive pixels of one channel per lane.  gfx950's transposed LDS
// read delivers exactly that from a pixel-major image: ds_read_b64_tr_b16 hands each lane of a 16-lane group one channel of a
// 4-pixel x 16-channel block (two reads = one 32x32x16 operand fragment).
//   * a workgroup stays inside one sample and walks 4 x 32-pixel tiles; per tile the dy tile (16 KB) and the x halo (6 x 34 pixels,
//     30 KB) arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`; out-of-image halo pixels come back as zeros);
//   * wave (cb, ib) owns output channels cb*32.. and input channels ib*32.. for ALL nine taps: 9 accumulator tiles = 144 VGPRs that
//     live for the whole run of tiles; the dy fragment of a 16-pixel chunk is read once and multiplied with nine shifted x fragments;
//   * LDS image: 1-KiB DMA pieces of 8 pixels, 16-byte slots XOR-swizzled with (pixel pair & 3) << 2 inside the 256-byte bank row of
//     a pixel pair -- the four pixels a 32-lane half reads land in different slot groups; the swizzle is applied on the DMA's
//     per-lane source address; for the reads it reduces to one base per kw and an XOR of 128 for the second 4-pixel block;
//   * at the end the lane's column (ci) is scaled by the rounded style, the 144 values go to the zero-filled float32 dW with one
//     atomic add each, and ds = sum w * H is reduced over the lane pair and added to the zero-filled ds.
// LDS 46 KB, <= 256 VGPRs: two workgroups per CU cover each other's DMA waits.
#include "mg_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_ __attribute__((address_space(3)))

constexpr int WG_C = 64, WG_TH = 4, WG_TW = 32, WG_HR = WG_TH + 2, WG_HROW = 40;
constexpr int WG_XP = WG_HR * WG_HROW / 8;          // 30 DMA pieces: the x halo, rows of 40 pixels (34 used)
constexpr int WG_DP = WG_TH * WG_TW / 8;            // 16 DMA pieces: the dy tile
constexpr int WG_X_BYTES = WG_XP * 1024, WG_D_BYTES = WG_DP * 1024;
constexpr int WG_LDS = WG_X_BYTES + WG_D_BYTES;     // 47,104

struct WgArgs {
    const mg_bf16* dy;      // [N,H,W,64]
    const mg_bf16* x;       // [N,H,W,64]
    const float* s;         // [N,64] or NULL
    const mg_bf16* w;       // [64][9][64] or NULL (needed for ds)
    float* dw;              // [64][9][64] float32, zero-filled
    float* ds;              // [N,64] float32, zero-filled, or NULL
    int n, h, w_;
    int tiles_x, tiles_y, per_wg;
    uint32_t bytes;         // N*H*W*128
};

__device__ __forceinline__ float wg_bf2f(mg_bf16 v) { return to_acc<mg_bf16>(v); }

// 4 pixels x 16 channels, transposed: the lane's channel at the block's 4 pixels
__device__ __forceinline__ s16x4 wg_tr(const LDS_ unsigned char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_ s16x4*)p); }

// (two 4-element reads -> one 8-element operand as ONE vector bit cast: element-wise __builtin_bit_cast(__bf16, a[e]) inserts are
// folded by this compiler into a broadcast of a[0] -- seen in the ISA as v_perm_b32 + v_mov of the first register only)
__device__ __forceinline__ bf16x8 wg_frag(s16x4 a, s16x4 b)
{
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ void __launch_bounds__(256, 2) wgrad3x3_c64_kernel(WgArgs p)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const LDS_ unsigned char* const xh = (const LDS_ unsigned char*)smem;                     // x halo image
    const LDS_ unsigned char* const dyt = (const LDS_ unsigned char*)(smem + WG_X_BYTES);     // dy tile image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave & 1, ib = wave >> 1;
    const int n = blockIdx.y;
    const int tiles_per_n = p.tiles_x * p.tiles_y;
    const int t_begin = blockIdx.x * p.per_wg;
    const int t_end = t_begin + p.per_wg < tiles_per_n ? t_begin + p.per_wg : tiles_per_n;
    if (t_begin >= t_end) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.bytes, 0x00020000);

    // ---- lane constants of the transposed reads: group gi = lane / 16 -> (pixel half h, channel sub-block ch16); lane 4*qq + pp of
    // the group supplies the address of pixel qq, channels 4*pp .. 4*pp+3 of the group's 16
    const uint32_t gi = (uint32_t)lane >> 4, hh = gi >> 1, ch16 = gi & 1u, qq = ((uint32_t)lane >> 2) & 3u, pp = (uint32_t)lane & 3u;
    auto lane_base = [&](uint32_t u, uint32_t cidx) -> uint32_t {       // u = pixel offset of the lane inside a 16-pixel chunk (+ kw)
        const uint32_t v = u >> 1, slot = (((u & 1u) << 3) | cidx) ^ ((v & 3u) << 2);
        return v * 256u + slot * 16u + (pp & 1u) * 8u;
    };
    const uint32_t aA = lane_base(8u * hh + qq, (uint32_t)cb * 4u + ch16 * 2u + (pp >> 1));
    uint32_t aB[3];
    #pragma unroll
    for (int kw = 0; kw < 3; kw++) aB[kw] = lane_base(8u * hh + qq + (uint32_t)kw, (uint32_t)ib * 4u + ch16 * 2u + (pp >> 1));

    f32x16 acc[9];
    #pragma unroll
    for (int t = 0; t < 9; t++)
        #pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    for (int t = t_begin; t < t_end; t++) {
        const int tx = t / p.tiles_y, ty = t - tx * p.tiles_y;       // column-major walk: consecutive tiles share halo rows (L2)
        const int y0 = ty * WG_TH, x0 = tx * WG_TW;
        __syncthreads();                        // every wave is done reading the previous tile's images
        {
            // (an opaque copy of the lane id: the per-lane parts below do not depend on the tile and would otherwise be kept in
            // registers across the whole launch next to the 144 accumulator registers)
            uint32_t lv = (uint32_t)lane;
            asm volatile("" : "+v"(lv));
            const uint32_t l = (lv & 15u) ^ (((lv >> 4) & 3u) << 2);
            const uint32_t sub = (lv >> 4) * 2u + (l >> 3), ch = (l & 7u) * 16u;           // pixel inside the piece's 8, byte offset of the 16-byte chunk
            #pragma unroll
            for (int i = 0; i < (WG_XP + 3) / 4; i++) {
                const int q = wave + 4 * i;
                if (q < WG_XP) {
                    const int hy = q / 5, hx = (q - hy * 5) * 8 + (int)sub;
                    const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                    const int ok = (hx < WG_TW + 2) & ((uint32_t)gy < (uint32_t)p.h) & ((uint32_t)gx < (uint32_t)p.w_);
                    const uint32_t off = ok ? ((uint32_t)((n * p.h + gy) * p.w_ + gx) * 128u + ch) : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_ void*)(smem + q * 1024), 16, (int)off, 0, 0, 0);
                }
            }
            #pragma unroll
            for (int i = 0; i < WG_DP / 4; i++) {
                const int q = wave + 4 * i;
                const int dyy = q / 4, dxx = (q - dyy * 4) * 8 + (int)sub;
                const uint32_t off = (uint32_t)((n * p.h + y0 + dyy) * p.w_ + x0 + dxx) * 128u + ch;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (LDS_ void*)(smem + WG_X_BYTES + q * 1024), 16, (int)off, 0, 0, 0);
            }
        }
        __syncthreads();                        // both images have landed (vmcnt(0) + barrier)

        const uint32_t za = aA;
        #pragma unroll
        for (int y = 0; y < WG_TH; y++)
            #pragma unroll
            for (int j = 0; j < 2; j++) {
                const LDS_ unsigned char* pa = dyt + (y * 16 + 8 * j) * 256;
                const bf16x8 A = wg_frag(wg_tr(pa + za), wg_tr(pa + ((za ^ 128u) + 2 * 256)));
                #pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    const int kh = tap / 3, kw = tap - kh * 3;
                    const LDS_ unsigned char* pb = xh + ((y + kh) * (WG_HROW / 2) + 8 * j) * 256;
                    const bf16x8 B = wg_frag(wg_tr(pb + aB[kw]), wg_tr(pb + ((aB[kw] ^ 128u) + 2 * 256)));
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[tap], 0, 0, 0);
                }
            }
    }

    // ---- results.  acc[tap][reg]: co = cb*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5), ci = ib*32 + (lane&31)
    const int ci = ib * 32 + (lane & 31), half = lane >> 5;
    if (p.ds && p.w) {
        float dsum = 0.f;
        #pragma unroll
        for (int tap = 0; tap < 9; tap++)
            #pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                dsum = __builtin_fmaf(acc[tap][reg], wg_bf2f(p.w[((size_t)co * 9 + tap) * WG_C + ci]), dsum);
            }
        dsum += __shfl_xor(dsum, 32);
        if (half == 0) atomicAdd(p.ds + (size_t)n * WG_C + ci, dsum);
    }
    const float sl = p.s ? wg_bf2f(from_acc<mg_bf16>(p.s[(size_t)n * WG_C + ci])) : 1.f;
    #pragma unroll
    for (int tap = 0; tap < 9; tap++)
        #pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int co = cb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            atomicAdd(p.dw + ((size_t)co * 9 + tap) * WG_C + ci, acc[tap][reg] * sl);
        }
}

} // namespace

extern "C" int mg_wgrad3x3_c64(const void* dy, const void* x, const float* s, const void* w, float* dw, float* ds,
                               int32_t n, int32_t h, int32_t wd, void* stream)
{
    MG_REQUIRE(dy && x && dw, "wgrad3x3_c64: null pointer");
    MG_REQUIRE(!ds || w, "wgrad3x3_c64: ds needs the weights");
    MG_REQUIRE(n >= 0 && h >= WG_TH && wd >= WG_TW && h % WG_TH == 0 && wd % WG_TW == 0,
               "wgrad3x3_c64: height must be a multiple of %d and width of %d", WG_TH, WG_TW);
    MG_REQUIRE((((uintptr_t)dy | (uintptr_t)x) & 15) == 0, "wgrad3x3_c64: activations must be 16-byte aligned");
    MG_REQUIRE((int64_t)n * h * wd * WG_C * 2 < (int64_t)0xFFFFFFF0u, "wgrad3x3_c64: x is too large");
    if (n == 0) return MG_OK;
    WgArgs p;
    p.dy = (const mg_bf16*)dy; p.x = (const mg_bf16*)x; p.s = s; p.w = (const mg_bf16*)w; p.dw = dw; p.ds = ds;
    p.n = n; p.h = h; p.w_ = wd; p.tiles_x = wd / WG_TW; p.tiles_y = h / WG_TH;
    p.bytes = (uint32_t)((int64_t)n * h * wd * WG_C * 2);
    const int tiles_per_n = p.tiles_x * p.tiles_y;
    int per_n = (512 + n - 1) / n;              // two workgroups per CU, each a contiguous run of tiles of one sample
    if (per_n > tiles_per_n) per_n = tiles_per_n;
    p.per_wg = (tiles_per_n + per_n - 1) / per_n;
    per_n = (tiles_per_n + p.per_wg - 1) / p.per_wg;
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_WGRAD, st, 2.0 * n * h * wd * WG_C * 2 + 9.0 * WG_C * WG_C * 4);
    hipLaunchKernelGGL(wgrad3x3_c64_kernel, dim3(per_n, n), dim3(256), WG_LDS, st, p);
    return mg_check_launch(MG_K_WGRAD, "wgrad3x3_c64");
}
