// torgb.hip -- the generator's ToRGB layer on channels-last reduced-precision activations as ONE pass each way.
//
// The reference's ToRGBLayer (training/networks.py:343-359) followed by the block's image-path cast (:460-464) is, on the
// training path (modulated_conv2d with fused_modconv=False, :62-88):
//
//     xs  = x * styles.to(x.dtype)[N,C,1,1]                 read x, write xs            (the modulation)
//     yc  = conv2d(xs, weight.to(x.dtype))                  read xs, write [N,O,H,W]    (1x1, C -> O = img_channels <= 4)
//     out = bias_act(yc, bias.to(x.dtype), clamp=conv_clamp)
//     y   = out.to(float32, contiguous)                     the image path stays float32 NCHW
//
// The contraction has 2*O = 8 FLOP per input byte: nothing for the matrix cores, everything for HBM -- the modulation pass
// alone moves the block's largest activation twice.  Here x is read ONCE (16-byte vectors of 8 channels, fully coalesced),
// each lane multiplies its 8 channels by the rounded styles (the same rounding to x.dtype as the three launches), takes the
// O partial dot products, the C/8 lanes of a pixel add them up by a butterfly of wave shuffles, and lanes 0..O-1 of the pixel
// write the float32 planes.  Backward (plain; the differentiable form stays with the composition in Python): x and dy are read
// once and dx written once; the three parameter gradients come out of ONE float32 quantity
//     G[n,o,c] = sum_px dyc[n,px,o] * x[n,px,c]        ->   dW[o,c] = sum_n s[n,c] G[n,o,c],   ds[n,c] = sum_o W[o,c] G[n,o,c]
// kept in registers per lane (the lane's 8 channels x O), reduced over the workgroup (wave shuffles, then LDS) and added to the
// zero-filled [N,O,C] accumulator once per workgroup; a second, tiny launch turns G into dW, ds, db.  dyc = dy rounded to T: for the
// linear activation the API saves no output, so bias_act's backward masks nothing even when a clamp is set (reference
// bias_act.py:26 `ref=''`, bias_act.cu:139-141 with yref = 0) -- ToRGB passes gradients through saturated pixels, and so does this.
// Algorithmic bytes: forward (C*2 + O*4) per pixel, backward (2*C*2 + O*4) per pixel.
#include "mg_common.h"
#include "mg_act.h"

namespace {

constexpr int TG_O = 4;         // most output channels (RGBA)
constexpr int TG_U = 4;         // 16-byte vectors in flight per lane

struct TgArgs {
    const void* x;      // [N, HW, C] T (channels-last)
    const void* w;      // [O, C] T
    const float* s;     // [N, C] float32 styles (rounded to T inside, as styles.to(x.dtype))
    const void* b;      // [O] T or NULL
    float* y;           // [N, O, HW] float32 forward output
    const float* gy;    // [N, O, HW] float32 incoming gradient
    void* gx;           // [N, HW, C] T
    float* G;           // [N, O, C] float32, zero-filled by the caller (one atomic add per value and workgroup)
    float* gb;          // [O] float32, zero-filled by the caller (NULL: not wanted)
    int n, c, hw, o;
    float clamp;
};

template <class T> struct alignas(16) TgVec { T v[8]; };

// The lane's 8 styles (rounded to T) and its 8 columns of the O weight rows: unconditional 16-byte loads (a load behind a
// run-time condition makes hipcc branch around it and drain the queue per element: 40 dependent round trips per workgroup).
template <class T>
__device__ __forceinline__ void tg_lane_constants(const TgArgs& p, int n, int chunk, int lpp, float (&sv)[8], float (&wv)[TG_O][8])
{
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const f4_t* sp = (const f4_t*)(p.s + (size_t)n * p.c + chunk * 8);
    const f4_t s0 = sp[0], s1 = sp[1];
    TgVec<T> wr[TG_O];
    #pragma unroll
    for (int o = 0; o < TG_O; o++) wr[o] = ((const TgVec<T>*)p.w)[(size_t)(o < p.o ? o : 0) * lpp + chunk];
    const float sf[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    #pragma unroll
    for (int j = 0; j < 8; j++) sv[j] = to_acc<T>(from_acc<T>(sf[j]));
    #pragma unroll
    for (int o = 0; o < TG_O; o++)
        #pragma unroll
        for (int j = 0; j < 8; j++) wv[o][j] = o < p.o ? to_acc<T>(wr[o].v[j]) : 0.f;
}

template <class T, int LPP_LOG2>
__global__ void __launch_bounds__(256) torgb_fwd_kernel(TgArgs p)
{
    constexpr int LPP = 1 << LPP_LOG2;              // lanes per pixel = C / 8
    const int n = blockIdx.y, tid = threadIdx.x, chunk = tid & (LPP - 1);
    const uint32_t vecs = (uint32_t)p.hw << LPP_LOG2;
    float sv[8], wv[TG_O][8];
    tg_lane_constants<T>(p, n, chunk, LPP, sv, wv);
    const bool b0 = chunk & 1, b1 = chunk & 2;
    const int mine_o = ((chunk & 1) << 1) | ((chunk >> 1) & 1);         // the output channel this lane ends up with (lanes 0..3 of the pixel store)
    float bias = 0.f;
    if (p.b) bias = to_acc<T>(((const T*)p.b)[mine_o < p.o ? mine_o : 0]);
    const BaConst<float> k = {0.f, 1.f, p.clamp};
    const TgVec<T>* xn = (const TgVec<T>*)p.x + (size_t)n * vecs;
    float* yn = p.y + (size_t)n * p.o * p.hw;
    for (uint32_t base = blockIdx.x * (256u * TG_U); base < vecs; base += gridDim.x * (256u * TG_U)) {
        TgVec<T> xin[TG_U];
        #pragma unroll
        for (int u = 0; u < TG_U; u++) {
            const uint32_t vi = base + u * 256u + tid;
            if (vi < vecs) xin[u] = xn[vi];
        }
        #pragma unroll
        for (int u = 0; u < TG_U; u++) {
            const uint32_t vi = base + u * 256u + tid;
            if (vi >= vecs) continue;               // (a pixel's lanes leave together: vecs and the bases are multiples of LPP)
            float acc[TG_O] = {0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int j = 0; j < 8; j++) {
                const float xs = to_acc<T>(from_acc<T>(to_acc<T>(xin[u].v[j]) * sv[j]));      // x * styles, rounded as the modulation pass stores it
                #pragma unroll
                for (int o = 0; o < TG_O; o++) acc[o] = __builtin_fmaf(xs, wv[o][j], acc[o]);
            }
            // the pixel's C/8 lanes add up: two exchange steps in which a lane gives away the half it does not keep (4 -> 2 -> 1
            // values: lane bit 0 picks the pair {0,1} / {2,3}, bit 1 the member), then log2(C/8) - 2 one-value butterfly steps:
            // 4 + ... shuffles per vector instead of 4 per step
            const float s0 = b0 ? acc[0] : acc[2], s1 = b0 ? acc[1] : acc[3];
            const float r0 = (b0 ? acc[2] : acc[0]) + __shfl_xor(s0, 1), r1 = (b0 ? acc[3] : acc[1]) + __shfl_xor(s1, 1);
            float v = (b1 ? r1 : r0) + __shfl_xor(b1 ? r0 : r1, 2);
            #pragma unroll
            for (int m = 4; m < LPP; m <<= 1) v += __shfl_xor(v, m);
            if (chunk < 4 && mine_o < p.o) {        // lanes 0..3 of the pixel hold output channels 0, 2, 1, 3
                const float yc = to_acc<T>(from_acc<T>(v));                                 // the convolution returns T
                const float out = to_acc<T>(from_acc<T>(ba_eval<float, 1>(0, yc, bias, 0.f, 0.f, 1.f, k)));
                yn[(size_t)mine_o * p.hw + (vi >> LPP_LOG2)] = out;
            }
        }
    }
}

template <class T, int LPP_LOG2>
__global__ void __launch_bounds__(256) torgb_bwd_kernel(TgArgs p)
{
    constexpr int LPP = 1 << LPP_LOG2;
    __shared__ float red[4][LPP < 64 ? LPP : 64][TG_O * 8 + TG_O];
    const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, chunk = tid & (LPP - 1);
    const uint32_t vecs = (uint32_t)p.hw << LPP_LOG2;
    float sv[8], wv[TG_O][8];
    tg_lane_constants<T>(p, n, chunk, LPP, sv, wv);
    float Gp[TG_O][8], gbp[TG_O];
    #pragma unroll
    for (int o = 0; o < TG_O; o++) {
        gbp[o] = 0.f;
        #pragma unroll
        for (int j = 0; j < 8; j++) Gp[o][j] = 0.f;
    }
    const TgVec<T>* xn = (const TgVec<T>*)p.x + (size_t)n * vecs;
    TgVec<T>* gxn = (TgVec<T>*)p.gx + (size_t)n * vecs;
    const float* gyn = p.gy + (size_t)n * p.o * p.hw;
    const int group0 = lane & ~(LPP - 1);           // first lane of this pixel's group inside the wave
    for (uint32_t base = blockIdx.x * (256u * TG_U); base < vecs; base += gridDim.x * (256u * TG_U)) {
        TgVec<T> xin[TG_U];
        float gq[TG_U];
        #pragma unroll
        for (int u = 0; u < TG_U; u++) {
            const uint32_t vi = base + u * 256u + tid;
            gq[u] = 0.f;
            if (vi < vecs) {
                xin[u] = xn[vi];
                if (chunk < p.o) gq[u] = gyn[(size_t)chunk * p.hw + (vi >> LPP_LOG2)];       // lane `chunk` of the pixel fetches the gradient of channel `chunk`
            }
        }
        #pragma unroll
        for (int u = 0; u < TG_U; u++) {
            const uint32_t vi = base + u * 256u + tid;
            const float mine = to_acc<T>(from_acc<T>(gq[u]));       // the cast's backward rounds the gradient to T; linear bias_act passes it on
            float g[TG_O];
            #pragma unroll
            for (int o = 0; o < TG_O; o++) g[o] = __shfl(mine, group0 + o);         // (every lane of the wave takes part)
            if (vi >= vecs) continue;
            TgVec<T> out;
            #pragma unroll
            for (int j = 0; j < 8; j++) {
                const float xj = to_acc<T>(xin[u].v[j]);
                float a = 0.f;
                #pragma unroll
                for (int o = 0; o < TG_O; o++) {
                    a = __builtin_fmaf(g[o], wv[o][j], a);
                    Gp[o][j] = __builtin_fmaf(g[o], xj, Gp[o][j]);
                }
                out.v[j] = from_acc<T>(to_acc<T>(from_acc<T>(a)) * sv[j]);          // the data gradient of the convolution is a T tensor, then * styles
            }
            gxn[vi] = out;
            #pragma unroll
            for (int o = 0; o < TG_O; o++) gbp[o] += g[o];
        }
    }
    // lanes of one wave that own the same channel chunk (stride LPP), then the four waves through LDS, then one atomic add per value
    #pragma unroll
    for (int m = LPP; m < 64; m <<= 1) {
        #pragma unroll
        for (int o = 0; o < TG_O; o++) {
            gbp[o] += __shfl_xor(gbp[o], m);
            #pragma unroll
            for (int j = 0; j < 8; j++) Gp[o][j] += __shfl_xor(Gp[o][j], m);
        }
    }
    if (lane < LPP) {
        #pragma unroll
        for (int o = 0; o < TG_O; o++) {
            red[wave][lane][TG_O * 8 + o] = gbp[o];
            #pragma unroll
            for (int j = 0; j < 8; j++) red[wave][lane][o * 8 + j] = Gp[o][j];
        }
    }
    __syncthreads();
    // 256 % LPP == 0 puts chunk == lane % LPP in every wave, so entry [w][l] of every wave w belongs to channel chunk l
    for (int e = tid; e < LPP * (TG_O * 8 + TG_O); e += 256) {
        const int l = e / (TG_O * 8 + TG_O), q = e - l * (TG_O * 8 + TG_O);
        const float v = red[0][l][q] + red[1][l][q] + red[2][l][q] + red[3][l][q];
        if (q < TG_O * 8) {
            const int o = q >> 3, j = q & 7;
            if (o < p.o) atomicAdd(p.G + ((size_t)n * p.o + o) * p.c + l * 8 + j, v);
        } else if (l == 0 && p.gb && q - TG_O * 8 < p.o) {
            atomicAdd(p.gb + (q - TG_O * 8), v);
        }
    }
}

// The parameter gradients out of G, one small launch: a workgroup owns 64 channels (lane x = channel, 4 lanes y walk the samples),
// writes ds[n,c] = sum_o W[o,c] G[n,o,c] and leaves dW[o,c] = sum_n T(s[n,c]) G[n,o,c] and db, rounded to T (the parameters the
// layer multiplies with are T tensors).
template <class T>
__global__ void __launch_bounds__(256) torgb_bwd_finish_kernel(const float* G, const float* gb32, const float* s, const T* w,
                                                               T* gw, float* gs, T* gb, int n, int c, int o)
{
    __shared__ float red[4][TG_O][64];
    const int cx = threadIdx.x & 63, ny = threadIdx.x >> 6, ch = blockIdx.x * 64 + cx;
    float wv[TG_O], acc[TG_O] = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
    for (int q = 0; q < TG_O; q++) wv[q] = to_acc<T>(w[(size_t)(q < o ? q : 0) * c + ch]);
    for (int i = ny; i < n; i += 4) {
        float g[TG_O];
        #pragma unroll
        for (int q = 0; q < TG_O; q++) g[q] = G[((size_t)i * o + (q < o ? q : 0)) * c + ch];
        const float sn = to_acc<T>(from_acc<T>(s[(size_t)i * c + ch]));
        float ds = 0.f;
        #pragma unroll
        for (int q = 0; q < TG_O; q++)
            if (q < o) { ds = __builtin_fmaf(wv[q], g[q], ds); acc[q] = __builtin_fmaf(sn, g[q], acc[q]); }
        gs[(size_t)i * c + ch] = ds;
    }
    #pragma unroll
    for (int q = 0; q < TG_O; q++) red[ny][q][cx] = acc[q];
    __syncthreads();
    if (ny == 0) {
        #pragma unroll
        for (int q = 0; q < TG_O; q++)
            if (q < o) gw[(size_t)q * c + ch] = from_acc<T>(red[0][q][cx] + red[1][q][cx] + red[2][q][cx] + red[3][q][cx]);
    }
    if (gb && blockIdx.x == 0 && threadIdx.x < o) gb[threadIdx.x] = from_acc<T>(gb32[threadIdx.x]);
}

// workgroups per sample of the backward pass: at least 2 grid-stride iterations each, the whole launch ~1024 workgroups when the
// tensor is large enough
uint32_t torgb_parts(int n, int c, int hw)
{
    const uint32_t vecs = (uint32_t)hw * (uint32_t)(c / 8);
    uint32_t by_work = vecs / (256u * TG_U * 2u), by_fill = (1024u + (uint32_t)n - 1) / (uint32_t)(n > 0 ? n : 1);
    uint32_t parts = by_work < by_fill ? by_work : by_fill;
    return parts < 1 ? 1 : parts > 256 ? 256 : parts;
}

template <class T>
int launch_torgb(const TgArgs& p, bool backward, uint32_t parts, hipStream_t st)
{
    const uint32_t lpp = p.c / 8, vecs = (uint32_t)p.hw * lpp;
    uint32_t bx = parts;
    if (!backward) {
        bx = (vecs + 256u * TG_U - 1) / (256u * TG_U);
        const uint32_t cap = (4096u + (uint32_t)p.n - 1) / (uint32_t)p.n;        // ~16 workgroups per CU over the whole launch
        if (bx > cap) bx = cap;
    }
    const dim3 grid(bx, p.n), block(256);
#define TG_LAUNCH(L)                                                                                          \
    do {                                                                                                      \
        if (backward) hipLaunchKernelGGL((torgb_bwd_kernel<T, L>), grid, block, 0, st, p);                    \
        else          hipLaunchKernelGGL((torgb_fwd_kernel<T, L>), grid, block, 0, st, p);                    \
    } while (0)
    switch (lpp) {
    case 8:  TG_LAUNCH(3); break;
    case 16: TG_LAUNCH(4); break;
    case 32: TG_LAUNCH(5); break;
    case 64: TG_LAUNCH(6); break;
    default: mg_set_error("torgb: the channel count must be 64, 128, 256 or 512 (got %d)", p.c); return MG_E_ARG;
    }
#undef TG_LAUNCH
    return MG_OK;
}

int check_torgb(const TgArgs& p, int dtype)
{
    MG_REQUIRE(dtype == MG_F16 || dtype == MG_BF16, "torgb: float16 or bfloat16 activations only");
    MG_REQUIRE(p.n >= 0 && p.hw >= 0 && p.o >= 1 && p.o <= TG_O, "torgb: 1..%d output channels", TG_O);
    MG_REQUIRE(p.c == 64 || p.c == 128 || p.c == 256 || p.c == 512, "torgb: the channel count must be 64, 128, 256 or 512 (got %d)", p.c);
    MG_REQUIRE((int64_t)p.n * p.hw * p.c <= INT32_MAX && (int64_t)p.hw * (p.c / 8) <= INT32_MAX - 256 * TG_U * 64, "torgb: x is too large");
    MG_REQUIRE((((uintptr_t)p.x | (uintptr_t)p.gx | (uintptr_t)p.w | (uintptr_t)p.s) & 15) == 0, "torgb: x, gx, w and s must be 16-byte aligned");
    return MG_OK;
}

} // namespace

extern "C" int mg_torgb_nhwc_fwd(const void* x, const void* w, const float* s, const void* b, float* y, int dtype,
                                 int32_t n, int32_t c, int32_t hw, int32_t o, float clamp, void* stream)
{
    MG_REQUIRE(x && w && s && y, "torgb_fwd: null pointer");
    TgArgs p = {};
    p.x = x; p.w = w; p.s = s; p.b = b; p.y = y; p.n = n; p.c = c; p.hw = hw; p.o = o; p.clamp = clamp;
    if (int rc = check_torgb(p, dtype)) return rc;
    if (n == 0 || hw == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_TORGB, st, (double)n * hw * (c * 2.0 + o * 4.0));
    int rc = dtype == MG_BF16 ? launch_torgb<mg_bf16>(p, false, 0, st) : launch_torgb<mg_half>(p, false, 0, st);
    if (rc) return rc;
    return mg_check_launch(MG_K_TORGB, "torgb_fwd");
}

extern "C" int mg_torgb_bwd_finish(const float* G, const float* gb32, const float* s, const void* w, void* gw, float* gs, void* gb,
                                   int dtype, int32_t n, int32_t c, int32_t o, void* stream)
{
    MG_REQUIRE(G && s && w && gw && gs && (gb32 || !gb), "torgb_bwd_finish: null pointer");
    MG_REQUIRE(dtype == MG_F16 || dtype == MG_BF16, "torgb: float16 or bfloat16 activations only");
    MG_REQUIRE(n >= 0 && c > 0 && c % 64 == 0 && o >= 1 && o <= TG_O, "torgb_bwd_finish: bad sizes");
    if (n == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MG_BF16) hipLaunchKernelGGL(torgb_bwd_finish_kernel<mg_bf16>, dim3(c / 64), dim3(256), 0, st, G, gb32, s, (const mg_bf16*)w, (mg_bf16*)gw, gs, (mg_bf16*)gb, n, c, o);
    else                  hipLaunchKernelGGL(torgb_bwd_finish_kernel<mg_half>, dim3(c / 64), dim3(256), 0, st, G, gb32, s, (const mg_half*)w, (mg_half*)gw, gs, (mg_half*)gb, n, c, o);
    return mg_check_launch(-1, "torgb_bwd_finish");
}

extern "C" int mg_torgb_nhwc_bwd(const void* x, const void* w, const float* s, const float* gy, void* gx, float* G, float* gb,
                                 int dtype, int32_t n, int32_t c, int32_t hw, int32_t o, void* stream)
{
    const int32_t parts = (int32_t)torgb_parts(n, c, hw);
    MG_REQUIRE(x && w && s && gy && gx && G, "torgb_bwd: null pointer");
    TgArgs p = {};
    p.x = x; p.w = w; p.s = s; p.gy = gy; p.gx = gx; p.G = G; p.gb = gb; p.n = n; p.c = c; p.hw = hw; p.o = o; p.clamp = -1.f;
    if (int rc = check_torgb(p, dtype)) return rc;
    if (n == 0 || hw == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_TORGB, st, (double)n * hw * (2.0 * c * 2.0 + o * 4.0));
    int rc = dtype == MG_BF16 ? launch_torgb<mg_bf16>(p, true, (uint32_t)parts, st) : launch_torgb<mg_half>(p, true, (uint32_t)parts, st);
    if (rc) return rc;
    return mg_check_launch(MG_K_TORGB, "torgb_bwd");
}
