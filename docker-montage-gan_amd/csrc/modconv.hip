// modconv.hip -- style-modulated 3x3 convolution of the 64-channel, full-resolution layers as ONE kernel on the matrix cores:
//
//     y = clamp( act( conv3x3( x * s[n,ci] ) * d[n,co] + noise[n,h,w] * strength + b[co] ) * gain )
//
// (reference training/networks.py:37-98 `modulated_conv2d`, training path, followed by the layer's bias_act, :311-313; with
// s = d = NULL it is the discriminator's conv0 + bias_act, :600-603).  As separate launches that is: the modulation pass
// (read x, write x*s), the convolution (read, write), the epilogue pass (read, write) -- six trips of the largest activation of
// the network ([N,64,256,256] bf16, 268 MB at batch 32) through HBM.  Here: read x once, write y once (and the raw convolution
// output once more when the backward pass will need it).  Bound: HBM (the contraction has 288 FLOP per activation byte; at
// 64 channels the matrix cores are ~3x faster than the memory system can feed them).
//
// Layout: NHWC bf16 activations (torch.channels_last), weights [Cout][kh][kw][Cin] bf16 (channels_last of [O,I,3,3]).
// Decomposition on CDNA4: a persistent grid of one 256-lane workgroup per CU walks 8x32-pixel output tiles.
//   * the 9 x 64 x 64 weights live in LDS for the whole launch (rows padded to 144 B: conflict-free 16-byte fragment reads);
//   * the tile's 10 x 34-pixel halo is staged into LDS with the modulation applied on the way (x * s rounded to bf16, as
//     the reference's `x * styles.to(x.dtype)` rounds it); the NEXT tile's halo is already in flight into registers while this
//     tile is being multiplied (global-load latency hidden behind the MFMAs);
//   * each of the 4 waves owns 2 image rows x 32 pixels x 64 output channels = 2 x 2 accumulators of
//     v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand (rows = output channels) and the pixels as B (columns), so
//     that a lane ends up with 4 consecutive output channels of one pixel per register quad;
//   * epilogue in registers on the float32 accumulators rounded to bf16 first (the reference's convolution returns bf16), then
//     through LDS so that every lane stores full 16-byte pieces of NHWC rows.
#include "mg_common.h"
#include "mg_act.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MC_C = 64;                 // channels in = channels out
constexpr int MC_TH = 8, MC_TW = 32;     // output tile (pixels)
constexpr int MC_HH = MC_TH + 2, MC_HW = MC_TW + 2;      // halo tile
constexpr int MC_ROW = MC_C + 8;         // LDS row of one pixel / one (tap, co) weight row, in bf16 elements: 144 bytes
constexpr int MC_W_ELEMS = 9 * MC_C * MC_ROW;            // 41,472 bf16 = 82,944 B
constexpr int MC_A_ELEMS = MC_HH * MC_HW * MC_ROW;       // 24,480 bf16 = 48,960 B
constexpr int MC_PRE = (MC_HH * MC_HW * 8 + 255) / 256;  // 16-byte halo pieces per lane (11)

struct McArgs {
    const mg_bf16* x;        // [N,H,W,64]
    const mg_bf16* w;        // [64][3][3][64]
    const float* s;          // [N,64] modulation (NULL: none)
    const float* d;          // [N,64] demodulation (NULL: none)
    const float* noise;      // [N,H,W] or [H,W] (NULL: none), multiplied by *strength
    const float* strength;   // device scalar (NULL: 1)
    const mg_bf16* b;        // [64] bias (NULL: none)
    mg_bf16* yconv;          // raw convolution output, bf16 (NULL: not wanted)
    mg_bf16* y;              // epilogue output (NULL: not wanted)
    mg_bf16* xs;             // modulated input x * s, bf16 (NULL: not wanted): what the library's weight gradient reads
    int n, h, w_;            // batch, height, width
    int64_t noise_stride_n;  // 0 = shared noise
    int act;                 // 1 linear, 3 lrelu
    float alpha, gain, clamp;
    int tiles_x, tiles_y, tiles;
    unsigned long long* timing;   // debug: cycles per phase of workgroup 0 (NULL: off)
};

__device__ __forceinline__ float bf2f(mg_bf16 v) { return to_acc<mg_bf16>(v); }

union Pack8 { uint4 u; mg_bf16 h[8]; };

template <int ACT>
__global__ void __launch_bounds__(256, 1) modconv3x3_c64_kernel(McArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    mg_bf16* wl = (mg_bf16*)smem_raw;                   // [9][64][MC_ROW]
    mg_bf16* al = wl + MC_W_ELEMS;                      // [MC_HH][MC_HW][MC_ROW]; reused as the output staging [256][MC_ROW]
    float* dl = (float*)(al + MC_A_ELEMS);              // [64] demodulation of the current sample
    float* bl = dl + MC_C;                              // [64] bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int c8 = tid & 7;                             // this lane's 8-channel piece in every staging step (256 % 8 == 0)

    // ---- weights -> LDS, once per launch: [co][tap][ci] in memory -> [tap][co][ci (+pad)] in LDS
    for (int e = tid; e < 9 * MC_C * 8; e += 256) {
        const int piece = e & 7, rowi = e >> 3;         // rowi = co * 9 + tap
        const int co = rowi / 9, tap = rowi - co * 9;
        const uint4 v = *(const uint4*)(p.w + (size_t)rowi * MC_C + piece * 8);
        *(uint4*)(wl + ((size_t)tap * MC_C + co) * MC_ROW + piece * 8) = v;
    }
    if (tid < MC_C) bl[tid] = p.b ? bf2f(p.b[tid]) : 0.f;

    const float strength = (p.noise && p.strength) ? p.strength[0] : 1.f;
    uint4 pre[MC_PRE];
    float pre_s[8], pre_d = 1.f, pre_nz[2] = {0.f, 0.f};     // the tile's modulation piece, demodulation entry and noise samples
    auto fetch = [&](int tile) {
        const int n = tile / (p.tiles_x * p.tiles_y), rem = tile - n * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int y0 = ty * MC_TH - 1, x0 = tx * MC_TW - 1;
        if (p.s) {
            const float4 a = *(const float4*)(p.s + (size_t)n * MC_C + c8 * 8), b4 = *(const float4*)(p.s + (size_t)n * MC_C + c8 * 8 + 4);
            pre_s[0] = a.x; pre_s[1] = a.y; pre_s[2] = a.z; pre_s[3] = a.w; pre_s[4] = b4.x; pre_s[5] = b4.y; pre_s[6] = b4.z; pre_s[7] = b4.w;
        }
        if (p.d && tid < MC_C) pre_d = p.d[(size_t)n * MC_C + tid];
        if (p.noise && p.y) {
            #pragma unroll
            for (int mi = 0; mi < 2; mi++)
                pre_nz[mi] = p.noise[(size_t)n * p.noise_stride_n + (size_t)(ty * MC_TH + wave * 2 + mi) * p.w_ + tx * MC_TW + r];
        }
        #pragma unroll
        for (int i = 0; i < MC_PRE; i++) {
            const int chunk = tid + i * 256, pix = chunk >> 3;
            const int py = pix / MC_HW, px = pix - py * MC_HW;
            const int gy = y0 + py, gx = x0 + px;
            uint4 v = {0u, 0u, 0u, 0u};
            if (pix < MC_HH * MC_HW && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_)
                v = *(const uint4*)(p.x + (((size_t)n * p.h + gy) * p.w_ + gx) * MC_C + c8 * 8);
            pre[i] = v;
        }
    };

    unsigned long long t_[6] = {0, 0, 0, 0, 0, 0}, tc = 0;
    #define MC_MARK(i) do { if (p.timing) { unsigned long long now = __builtin_readcyclecounter(); t_[i] += now - tc; tc = now; } } while (0)
    int tile = blockIdx.x;
    if (tile < p.tiles) fetch(tile);
    if (p.timing) tc = __builtin_readcyclecounter();
    for (; tile < p.tiles; tile += gridDim.x) {
        const int n = tile / (p.tiles_x * p.tiles_y), rem = tile - n * (p.tiles_x * p.tiles_y);
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int oy0 = ty * MC_TH, ox0 = tx * MC_TW;

        // ---- halo (already in registers) -> LDS, modulated
        float sv[8];
        #pragma unroll
        for (int j = 0; j < 8; j++) sv[j] = p.s ? bf2f(from_acc<mg_bf16>(pre_s[j])) : 1.f;
        const float nzv[2] = {pre_nz[0], pre_nz[1]};
        #pragma unroll
        for (int i = 0; i < MC_PRE; i++) {
            const int chunk = tid + i * 256, pix = chunk >> 3;
            if (pix < MC_HH * MC_HW) {
                Pack8 v; v.u = pre[i];
                if (p.s) {
                    #pragma unroll
                    for (int j = 0; j < 8; j++) v.h[j] = from_acc<mg_bf16>(bf2f(v.h[j]) * sv[j]);
                }
                *(uint4*)(al + (size_t)pix * MC_ROW + c8 * 8) = v.u;
                if (p.xs) {         // the tile's own pixels (not the halo ring) of the modulated input, for the backward pass
                    const int py = pix / MC_HW, px = pix - py * MC_HW;
                    if (py >= 1 && py <= MC_TH && px >= 1 && px <= MC_TW)
                        *(uint4*)(p.xs + (((size_t)n * p.h + oy0 + py - 1) * p.w_ + ox0 + px - 1) * MC_C + c8 * 8) = v.u;
                }
            }
        }
        if (tid < MC_C) dl[tid] = p.d ? pre_d : 1.f;
        __syncthreads();
        MC_MARK(0);

        // ---- next tile's halo goes into flight now; it lands while the matrix cores work
        if (tile + (int)gridDim.x < p.tiles) fetch(tile + gridDim.x);

        MC_MARK(1);
        // ---- 9 taps x 4 k-steps x (2 channel halves x 2 rows) MFMAs
        f32x16 acc[2][2];
        #pragma unroll
        for (int a = 0; a < 2; a++)
            #pragma unroll
            for (int b2 = 0; b2 < 2; b2++)
                #pragma unroll
                for (int e = 0; e < 16; e++) acc[a][b2][e] = 0.f;
        #pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int kh = tap / 3, kw = tap - kh * 3;
            const mg_bf16* wrow = wl + ((size_t)tap * MC_C + r) * MC_ROW + hh * 8;                 // + ni * 32 * MC_ROW + kc * 16
            const mg_bf16* arow = al + ((size_t)(wave * 2 + kh) * MC_HW + r + kw) * MC_ROW + hh * 8;  // + mi * MC_HW * MC_ROW + kc * 16
            #pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                bf16x8 wf[2], xf[2];
                #pragma unroll
                for (int ni = 0; ni < 2; ni++) wf[ni] = *(const bf16x8*)(wrow + (size_t)ni * 32 * MC_ROW + kc * 16);
                #pragma unroll
                for (int mi = 0; mi < 2; mi++) xf[mi] = *(const bf16x8*)(arow + (size_t)mi * MC_HW * MC_ROW + kc * 16);
                #pragma unroll
                for (int ni = 0; ni < 2; ni++)
                    #pragma unroll
                    for (int mi = 0; mi < 2; mi++)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
            }
        }
        __syncthreads();            // every wave is done reading the halo tile: its LDS becomes the output staging area
        MC_MARK(2);

        // ---- outputs.  acc[ni][mi][reg]: output channel ni*32 + (reg&3) + 8*(reg>>2) + 4*hh of pixel (row wave*2+mi, column r)
        const BaConst<float> k = {p.alpha, p.gain, p.clamp};
        // this lane's 32 output channels: demodulation (rounded to the activation dtype, the reference's `dcoefs.to(x.dtype)`)
        // and bias, fetched from LDS once per tile as 16-byte reads -- not once per element inside the loops below, where
        // every read would be a full LDS round trip in front of its arithmetic at one wave per SIMD
        float dq[2][4][4], bq[2][4][4];
        #pragma unroll
        for (int ni = 0; ni < 2; ni++)
            #pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = ni * 32 + 8 * g + 4 * hh;
                const float4 dv = *(const float4*)(dl + co), bv = *(const float4*)(bl + co);
                dq[ni][g][0] = bf2f(from_acc<mg_bf16>(dv.x)); dq[ni][g][1] = bf2f(from_acc<mg_bf16>(dv.y));
                dq[ni][g][2] = bf2f(from_acc<mg_bf16>(dv.z)); dq[ni][g][3] = bf2f(from_acc<mg_bf16>(dv.w));
                bq[ni][g][0] = bv.x; bq[ni][g][1] = bv.y; bq[ni][g][2] = bv.z; bq[ni][g][3] = bv.w;
            }
        for (int which = 0; which < 2; which++) {
            mg_bf16* out = which == 0 ? p.yconv : p.y;
            if (!out) continue;
            #pragma unroll
            for (int mi = 0; mi < 2; mi++) {
                const int prow = wave * 2 + mi;
                const float nz = nzv[mi];
                #pragma unroll
                for (int ni = 0; ni < 2; ni++)
                    #pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int co = ni * 32 + 8 * g + 4 * hh;
                        union { uint2 u; mg_bf16 h[4]; } q;
                        #pragma unroll
                        for (int e = 0; e < 4; e++) {
                            float v = bf2f(from_acc<mg_bf16>(acc[ni][mi][g * 4 + e]));      // the convolution's own bf16 result
                            if (which == 1) {
                                // the arithmetic of fma_bias_act_nhwc_fwd_kernel (csrc/fused.hip): d rounded to the activation
                                // dtype, one fma for the noise, the sum rounded to the activation dtype, then bias + act + gain + clamp
                                if (p.d) v *= dq[ni][g][e];
                                if (p.noise) v = __builtin_fmaf(nz, strength, v);
                                v = bf2f(from_acc<mg_bf16>(v));
                                v = ba_eval<float, ACT>(0, v, bq[ni][g][e], 0.f, 0.f, 1.f, k);
                            }
                            q.h[e] = from_acc<mg_bf16>(v);
                        }
                        *(uint2*)(al + ((size_t)prow * MC_TW + r) * MC_ROW + co) = q.u;
                    }
            }
            __syncthreads();
            MC_MARK(3);
            // 256 pixels x 8 pieces of 16 bytes, consecutive lanes on consecutive pieces of an NHWC row
            #pragma unroll
            for (int i = 0; i < 8; i++) {
                const int chunk = tid + i * 256, pix = chunk >> 3;
                const int prow = pix >> 5, pcol = pix & 31;
                const uint4 v = *(const uint4*)(al + (size_t)pix * MC_ROW + c8 * 8);
                *(uint4*)(out + (((size_t)n * p.h + oy0 + prow) * p.w_ + ox0 + pcol) * MC_C + c8 * 8) = v;
            }
            __syncthreads();
            MC_MARK(4);
        }
    }
    if (p.timing && blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 5; i++) p.timing[i] = t_[i];
}

} // namespace

static unsigned long long* g_mc_timing = nullptr;
extern "C" void mg_modconv_debug_timing(void* buf) { g_mc_timing = (unsigned long long*)buf; }

extern "C" int mg_modconv3x3_c64(const void* x, const void* w, const float* s, const float* d, const float* noise, const float* strength,
                                 const void* b, void* yconv, void* y, void* xs, int32_t n, int32_t h, int32_t wd, int64_t noise_stride_n,
                                 int act, float alpha, float gain, float clamp, void* stream)
{
    MG_REQUIRE(x && w && (y || yconv), "modconv3x3_c64: null pointer");
    MG_REQUIRE(n >= 0 && h >= MC_TH && wd >= MC_TW && h % MC_TH == 0 && wd % MC_TW == 0,
               "modconv3x3_c64: height must be a multiple of %d and width of %d", MC_TH, MC_TW);
    MG_REQUIRE(act == 1 || act == 3, "modconv3x3_c64: activation must be linear (1) or lrelu (3)");
    MG_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)yconv) & 15) == 0, "modconv3x3_c64: pointers must be 16-byte aligned");
    MG_REQUIRE((int64_t)n * h * wd * MC_C <= INT32_MAX, "x is too large");
    if (n == 0) return MG_OK;
    McArgs p;
    p.x = (const mg_bf16*)x; p.w = (const mg_bf16*)w; p.s = s; p.d = d; p.noise = noise; p.strength = strength; p.b = (const mg_bf16*)b;
    p.yconv = (mg_bf16*)yconv; p.y = (mg_bf16*)y; p.xs = (mg_bf16*)xs; p.n = n; p.h = h; p.w_ = wd; p.noise_stride_n = noise_stride_n;
    p.act = act; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.tiles_x = wd / MC_TW; p.tiles_y = h / MC_TH; p.tiles = n * p.tiles_x * p.tiles_y;
    p.timing = g_mc_timing;
    const size_t lds = (size_t)(MC_W_ELEMS + MC_A_ELEMS) * 2 + 2 * MC_C * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_c64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)modconv3x3_c64_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    const int grid = p.tiles < 256 ? p.tiles : 256;
    const double px = (double)n * h * wd * MC_C * 2;
    MgLaunchScope scope(MG_K_MODCONV, st, px * (1 + (y ? 1 : 0) + (yconv ? 1 : 0) + (xs ? 1 : 0)) + 9.0 * MC_C * MC_C * 2);
    if (act == 3) hipLaunchKernelGGL(modconv3x3_c64_kernel<3>, dim3(grid), dim3(256), lds, st, p);
    else          hipLaunchKernelGGL(modconv3x3_c64_kernel<1>, dim3(grid), dim3(256), lds, st, p);
    return mg_check_launch(MG_K_MODCONV, "modconv3x3_c64");
}
