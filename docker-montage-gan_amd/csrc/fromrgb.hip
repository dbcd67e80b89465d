// fromrgb.hip -- the discriminator's FromRGB layer on channels-last reduced-precision tensors as ONE pass each way.
//
// The reference's DiscriminatorBlock (training/networks.py:589-594: `y = self.fromrgb(img)`, a Conv2dLayer with kernel_size 1, :138-186)
// runs, per forward,
//
//     yc = conv2d(img, weight.to(x.dtype))                 1x1, I = img_channels (<= 4) -> C = 64..512: reads 8 B, writes C*2 B per pixel
//     y  = bias_act(yc, bias.to(x.dtype), lrelu, gain, clamp)             reads and writes the C-channel tensor again
//
// and per backward the bias_act gradient (reads dy and y, writes dpre), the convolution's weight gradient (reads dpre and img), the
// bias gradient (reads dpre again) and -- when the images carry gradients (the generator phases) -- its data gradient (dpre once more).
// 2*I = 8 FLOP per output byte: an HBM problem.  Here the C-channel tensor is written once in the forward pass (acc -> T -> + bias ->
// activation -> gain -> clamp -> T, the rounding points of the two launches) and read once in the backward pass: a lane owns 8 output
// channels of a pixel, keeps its 8 x I weights in registers, and
//     dpre = T(dy * f'(y) * gain  where |y| < clamp)             (the tensor autograd would have materialised)
//     dW[c][i] += dpre[c] * img[i],   db[c] += dpre[c]           32 + 8 float32 running sums per lane -> wave shuffles -> LDS -> one atomic
//                                                                per value and workgroup into the zero-filled accumulators
//     dimg[i]  = T(sum_c w[c][i] * dpre[c])                      butterfly over the C/8 lanes of the pixel (only when wanted)
// Algorithmic bytes: forward (I + C) * sizeof(T) per pixel, backward (2 C + I [+ I]) * sizeof(T).
#include "mg_common.h"
#include "mg_act.h"

namespace {

constexpr int FR_I = 4;         // most input channels (RGBA); fewer are zero-padded in registers
constexpr int FR_U = 4;         // pixels in flight per lane

struct FrArgs {
    const void* x;      // [N*HW, I] T (channels-last image)
    const void* w;      // [C, I] T
    const void* b;      // [C] T or NULL
    void* y;            // [N*HW, C] T forward output / saved output in the backward pass
    const void* dy;     // [N*HW, C] T
    void* dx;           // [N*HW, I] T or NULL
    float* gw;          // [C, I] float32, zero-filled
    float* gb;          // [C] float32, zero-filled, or NULL
    int c, i;
    uint32_t pixels, pix_per_wg;
    float alpha, gain, clamp;
};

template <class T> struct alignas(16) FrVec { T v[8]; };

// the image's I <= 4 channels of one pixel as floats (I = 4: one 8-byte load; else element loads)
template <class T, bool I4>
__device__ __forceinline__ void fr_load_pixel(const T* x, uint32_t pix, int ni, float (&v)[FR_I])
{
    if (I4) {
        struct alignas(8) Q { T v[4]; };
        const Q q = *(const Q*)(x + (size_t)pix * 4);
        #pragma unroll
        for (int i = 0; i < 4; i++) v[i] = to_acc<T>(q.v[i]);
    } else {
        T raw[FR_I];
        #pragma unroll
        for (int i = 0; i < FR_I; i++) raw[i] = x[(size_t)pix * ni + (i < ni ? i : 0)];
        #pragma unroll
        for (int i = 0; i < FR_I; i++) v[i] = i < ni ? (float)to_acc<T>(raw[i]) : 0.f;
    }
}

// the lane's 8 x I weights (and 8 biases) as floats.  I = 4: the 8 rows are 64 contiguous bytes -> four unconditional 16-byte loads (element by
// element behind `i < p.i`, every load sat behind its own test and `s_waitcnt vmcnt(0)`: 32 serialised round trips per workgroup, which made the
// first version of this kernel slower than the two launches it replaces); I < 4: element loads from clamped indices, dropped by selects.
template <class T>
__device__ __forceinline__ void fr_lane_weights(const FrArgs& p, int chunk, float (&wv)[8][FR_I])
{
    const T* w = (const T*)p.w;
    if (p.i == 4) {
        FrVec<T> q[4];
        #pragma unroll
        for (int r = 0; r < 4; r++) q[r] = ((const FrVec<T>*)(w + (size_t)chunk * 32))[r];
        #pragma unroll
        for (int e = 0; e < 8; e++)
            #pragma unroll
            for (int i = 0; i < 4; i++) wv[e][i] = to_acc<T>(q[(e * 4 + i) / 8].v[(e * 4 + i) % 8]);
    } else {
        T raw[8][FR_I];
        #pragma unroll
        for (int e = 0; e < 8; e++)
            #pragma unroll
            for (int i = 0; i < FR_I; i++) raw[e][i] = w[(size_t)(chunk * 8 + e) * p.i + (i < p.i ? i : 0)];
        #pragma unroll
        for (int e = 0; e < 8; e++)
            #pragma unroll
            for (int i = 0; i < FR_I; i++) wv[e][i] = i < p.i ? (float)to_acc<T>(raw[e][i]) : 0.f;
    }
}

template <class T, int A, bool I4>
__global__ void __launch_bounds__(256) fromrgb_fwd_kernel(FrArgs p)
{
    const int lpp = p.c / 8;                               // lanes per pixel
    const int chunk = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppi = 256 / lpp;
    const BaConst<float> k = {p.alpha, p.gain, p.clamp};
    float wv[8][FR_I], bv[8];
    fr_lane_weights<T>(p, chunk, wv);
    #pragma unroll
    for (int e = 0; e < 8; e++) bv[e] = 0.f;
    if (p.b) {
        const FrVec<T> q = ((const FrVec<T>*)p.b)[chunk];
        #pragma unroll
        for (int e = 0; e < 8; e++) bv[e] = to_acc<T>(q.v[e]);
    }
    const uint32_t lo = blockIdx.x * p.pix_per_wg, hi = min(p.pixels, lo + p.pix_per_wg);
    for (uint32_t pix0 = lo + pl; pix0 < hi; pix0 += ppi * FR_U) {
        float xv[FR_U][FR_I];
        #pragma unroll
        for (int u = 0; u < FR_U; u++) {
            const uint32_t pix = pix0 + u * ppi;
            fr_load_pixel<T, I4>((const T*)p.x, pix < hi ? pix : lo, p.i, xv[u]);
        }
        #pragma unroll
        for (int u = 0; u < FR_U; u++) {
            const uint32_t pix = pix0 + u * ppi;
            if (pix >= hi) continue;
            FrVec<T> out;
            #pragma unroll
            for (int e = 0; e < 8; e++) {
                float acc = 0.f;
                #pragma unroll
                for (int i = 0; i < FR_I; i++) acc = __builtin_fmaf(wv[e][i], xv[u][i], acc);
                asm volatile("" : "+v"(acc));                       // a float32 value of its own, then the convolution's rounding to T
                const float yc = to_acc<T>(from_acc<T>(acc));
                out.v[e] = from_acc<T>(ba_eval<float, A>(0, yc, bv[e], 0.f, 0.f, 1.f, k));
            }
            *(FrVec<T>*)((T*)p.y + (size_t)pix * p.c + chunk * 8) = out;
        }
    }
}

template <class T, int A, bool WANT_DX, bool I4>
__global__ void __launch_bounds__(256) fromrgb_bwd_kernel(FrArgs p)
{
    const int lpp = p.c / 8;
    const int chunk = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppi = 256 / lpp;
    const BaConst<float> k = {p.alpha, p.gain, p.clamp};
    float wv[8][FR_I];
    fr_lane_weights<T>(p, chunk, wv);
    float gw[8][FR_I], gb[8];
    #pragma unroll
    for (int e = 0; e < 8; e++) {
        gb[e] = 0.f;
        #pragma unroll
        for (int i = 0; i < FR_I; i++) gw[e][i] = 0.f;
    }
    const uint32_t lo = blockIdx.x * p.pix_per_wg, hi = min(p.pixels, lo + p.pix_per_wg);
    for (uint32_t pix0 = lo + pl; pix0 < hi; pix0 += ppi * FR_U) {
        FrVec<T> vdy[FR_U], vy[FR_U];
        float xv[FR_U][FR_I];
        #pragma unroll
        for (int u = 0; u < FR_U; u++) {
            const uint32_t pix = pix0 + u * ppi, pc = pix < hi ? pix : lo;          // (a clamped, valid address: no branch around the loads)
            vdy[u] = *(const FrVec<T>*)((const T*)p.dy + (size_t)pc * p.c + chunk * 8);
            if (A != 1) vy[u] = *(const FrVec<T>*)((const T*)p.y + (size_t)pc * p.c + chunk * 8);
            fr_load_pixel<T, I4>((const T*)p.x, pc, p.i, xv[u]);
        }
        #pragma unroll
        for (int u = 0; u < FR_U; u++) {
            const uint32_t pix = pix0 + u * ppi;
            const bool live = pix < hi;
            float part[FR_I] = {0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int e = 0; e < 8; e++) {
                // (linear: the API saves no output -- reference bias_act.py:26 `ref=''` -- so nothing is masked where the clamp saturated)
                float g = ba_eval<float, A>(1, to_acc<T>(vdy[u].v[e]), 0.f, 0.f, A == 1 ? 0.f : (float)to_acc<T>(vy[u].v[e]), 1.f, k);
                g = live ? (float)to_acc<T>(from_acc<T>(g)) : 0.f;            // autograd materialises dpre in T
                gb[e] += g;
                #pragma unroll
                for (int i = 0; i < FR_I; i++) {
                    gw[e][i] = __builtin_fmaf(g, xv[u][i], gw[e][i]);
                    if (WANT_DX) part[i] = __builtin_fmaf(wv[e][i], g, part[i]);
                }
            }
            if (WANT_DX) {
                // the C/8 lanes of the pixel are consecutive lanes: butterfly within the group (lpp is a power of two <= 64)
                for (int o = 1; o < lpp; o <<= 1)
                    #pragma unroll
                    for (int i = 0; i < FR_I; i++) part[i] += __shfl_xor(part[i], o, 64);
                if (chunk == 0 && live) {
                    T* dst = (T*)p.dx + (size_t)pix * p.i;
                    if (I4) {
                        struct alignas(8) Q { T v[4]; } q;
                        #pragma unroll
                        for (int i = 0; i < 4; i++) q.v[i] = from_acc<T>(part[i]);
                        *(Q*)dst = q;
                    } else {
                        #pragma unroll
                        for (int i = 0; i < FR_I; i++) if (i < p.i) dst[i] = from_acc<T>(part[i]);
                    }
                }
            }
        }
    }
    // ---- the lane's 40 sums: lanes with the same `chunk` sit lpp apart -> wave shuffles over the strides >= lpp, then the 4 waves meet in LDS
    #pragma unroll
    for (int e = 0; e < 8; e++) {
        for (int o = lpp; o < 64; o <<= 1) {
            gb[e] += __shfl_xor(gb[e], o, 64);
            #pragma unroll
            for (int i = 0; i < FR_I; i++) gw[e][i] += __shfl_xor(gw[e][i], o, 64);
        }
    }
    __shared__ float red[4][64][8 * (FR_I + 1) + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lpp <= 64 && lane < lpp) {
        #pragma unroll
        for (int e = 0; e < 8; e++) {
            red[wave][lane][e * (FR_I + 1) + FR_I] = gb[e];
            #pragma unroll
            for (int i = 0; i < FR_I; i++) red[wave][lane][e * (FR_I + 1) + i] = gw[e][i];
        }
    }
    __syncthreads();
    // waves cover different pixels of the same channel chunks when lpp <= 64 (one wave holds whole pixels); value v of chunk ch
    const int vals = 8 * (FR_I + 1);
    for (int idx = threadIdx.x; idx < lpp * vals; idx += 256) {
        const int ch = idx / vals, v = idx - ch * vals, e = v / (FR_I + 1), i = v - e * (FR_I + 1);
        const float t = red[0][ch][v] + red[1][ch][v] + red[2][ch][v] + red[3][ch][v];
        if (i == FR_I) { if (p.gb) atomicAdd(p.gb + ch * 8 + e, t); }
        else if (i < p.i) atomicAdd(p.gw + (size_t)(ch * 8 + e) * p.i + i, t);
    }
}

template <class T>
__global__ void fromrgb_finish_kernel(const float* gw32, const float* gb32, T* gw, T* gb, int nw, int nb)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < nw) gw[t] = from_acc<T>(gw32[t]);
    if (gb32 && gb && t < nb) gb[t] = from_acc<T>(gb32[t]);
}

int check_fromrgb(const FrArgs& p, int dtype, int act)
{
    MG_REQUIRE(dtype == MG_F16 || dtype == MG_BF16, "fromrgb: float16 or bfloat16 tensors only");
    MG_REQUIRE(act == 1 || act == 3, "fromrgb: activation must be linear (1) or lrelu (3)");
    MG_REQUIRE(p.c == 64 || p.c == 128 || p.c == 256 || p.c == 512, "fromrgb: 64, 128, 256 or 512 output channels");
    MG_REQUIRE(p.i >= 1 && p.i <= FR_I, "fromrgb: 1..%d input channels", FR_I);
    MG_REQUIRE((((uintptr_t)p.x | (uintptr_t)p.dx) & 7) == 0 && (((uintptr_t)p.y | (uintptr_t)p.dy | (uintptr_t)p.w | (uintptr_t)p.b) & 15) == 0,
               "fromrgb: pointers must be aligned (16 bytes; the image 8)");
    MG_REQUIRE((int64_t)p.pixels * p.c * 2 < (int64_t)1 << 40, "fromrgb: too large");
    return MG_OK;
}

void fr_grid(FrArgs& p, dim3& grid)
{
    // ~4 workgroups per CU, each a contiguous run of pixels: the backward kernel ends in one atomic per accumulator value and workgroup, all
    // of them on the same 5 C addresses -- 4096 small workgroups would queue 4096 adds per address
    const uint32_t ppi = 256u / (uint32_t)(p.c / 8), unit = ppi * FR_U;
    uint32_t wgs = 1024u;
    uint32_t per = ((p.pixels + wgs - 1) / wgs + unit - 1) / unit * unit;
    if (per < unit * 4u) per = unit * 4u;
    wgs = (p.pixels + per - 1) / per;
    p.pix_per_wg = per;
    grid = dim3(wgs);
}

} // namespace

extern "C" int mg_fromrgb_nhwc_fwd(const void* x, const void* w, const void* b, void* y, int dtype, int act, float alpha, float gain, float clamp,
                                   int64_t pixels, int32_t c, int32_t i, void* stream)
{
    MG_REQUIRE(x && w && y, "fromrgb_fwd: null pointer");
    MG_REQUIRE(pixels >= 0 && pixels < (int64_t)UINT32_MAX / 2, "fromrgb: too many pixels");
    FrArgs p = {};
    p.x = x; p.w = w; p.b = b; p.y = y; p.c = c; p.i = i; p.pixels = (uint32_t)pixels; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    if (int rc = check_fromrgb(p, dtype, act)) return rc;
    if (pixels == 0) return MG_OK;
    dim3 grid; fr_grid(p, grid);
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_FROMRGB, st, (double)pixels * (c + i) * 2.0);
#define FR_FWD(T, A) do { if (i == 4) hipLaunchKernelGGL((fromrgb_fwd_kernel<T, A, true>), grid, dim3(256), 0, st, p); \
                          else hipLaunchKernelGGL((fromrgb_fwd_kernel<T, A, false>), grid, dim3(256), 0, st, p); } while (0)
    if (dtype == MG_BF16) { if (act == 3) FR_FWD(mg_bf16, 3); else FR_FWD(mg_bf16, 1); }
    else                  { if (act == 3) FR_FWD(mg_half, 3); else FR_FWD(mg_half, 1); }
#undef FR_FWD
    return mg_check_launch(MG_K_FROMRGB, "fromrgb_fwd");
}

extern "C" int mg_fromrgb_nhwc_bwd(const void* x, const void* w, const void* y, const void* dy, void* dx, float* gw32, float* gb32, void* gw, void* gb,
                                   int dtype, int act, float alpha, float gain, float clamp, int64_t pixels, int32_t c, int32_t i, void* stream)
{
    MG_REQUIRE(x && w && y && dy && gw32 && gw, "fromrgb_bwd: null pointer");
    MG_REQUIRE(!gb || gb32, "fromrgb_bwd: the bias gradient needs its float32 accumulator");
    MG_REQUIRE(pixels >= 0 && pixels < (int64_t)UINT32_MAX / 2, "fromrgb: too many pixels");
    FrArgs p = {};
    p.x = x; p.w = w; p.y = const_cast<void*>(y); p.dy = dy; p.dx = dx; p.gw = gw32; p.gb = gb32; p.c = c; p.i = i; p.pixels = (uint32_t)pixels;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    if (int rc = check_fromrgb(p, dtype, act)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (pixels > 0) {
        dim3 grid; fr_grid(p, grid);
        MgLaunchScope scope(MG_K_FROMRGB, st, (double)pixels * (2.0 * c + i + (dx ? i : 0)) * 2.0);
#define FR_BWD(T, A) do { if (dx) { if (i == 4) hipLaunchKernelGGL((fromrgb_bwd_kernel<T, A, true, true>), grid, dim3(256), 0, st, p); \
                                    else hipLaunchKernelGGL((fromrgb_bwd_kernel<T, A, true, false>), grid, dim3(256), 0, st, p); } \
                          else    { if (i == 4) hipLaunchKernelGGL((fromrgb_bwd_kernel<T, A, false, true>), grid, dim3(256), 0, st, p); \
                                    else hipLaunchKernelGGL((fromrgb_bwd_kernel<T, A, false, false>), grid, dim3(256), 0, st, p); } } while (0)
        if (dtype == MG_BF16) { if (act == 3) FR_BWD(mg_bf16, 3); else FR_BWD(mg_bf16, 1); }
        else                  { if (act == 3) FR_BWD(mg_half, 3); else FR_BWD(mg_half, 1); }
#undef FR_BWD
        if (int rc = mg_check_launch(MG_K_FROMRGB, "fromrgb_bwd")) return rc;
    }
    const int nw = c * i, blocks = (nw + 255) / 256;
    if (dtype == MG_BF16) hipLaunchKernelGGL(fromrgb_finish_kernel<mg_bf16>, dim3(blocks), dim3(256), 0, st, gw32, gb32, (mg_bf16*)gw, (mg_bf16*)gb, nw, c);
    else                  hipLaunchKernelGGL(fromrgb_finish_kernel<mg_half>, dim3(blocks), dim3(256), 0, st, gw32, gb32, (mg_half*)gw, (mg_half*)gb, nw, c);
    return mg_check_launch(-1, "fromrgb_finish");
}
