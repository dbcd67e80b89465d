// residual.hip -- the tail of the discriminator's residual block in one pass each way.
//
// The reference's DiscriminatorBlock (training/networks.py:596-600, architecture 'resnet') ends with
//
//     y = self.skip(x, gain=sqrt(0.5))        ->  ... conv ... ; y = bias_act(y_raw, None, 'linear', gain = sqrt(0.5))        read, write
//     x = self.conv1(x, gain=sqrt(0.5))       ->  ... conv ... ; x = bias_act(x_raw, b, 'lrelu', gain, clamp)                 read, write
//     x = y.add_(x)                                                                                                          2 reads, write
//
// seven passes over the block's output tensor where three do: out = T(T(bias_act(x_raw, b)) + T(y_raw * g)) reads the two raw convolution
// outputs once and writes once (the rounding points of the three launches: each bias_act result and the sum are T tensors).  Backward: the
// activation's output is not kept -- it is recomputed from x_raw and b (a handful of VALU operations), which gives the same sign / clamp mask
// bias_act's backward reads from its saved output -- so one pass reads dout and x_raw and writes
//     dx = T(dout * f'(a) * gain  where |a| < clamp),   dy = T(dout * g),   db += dx        (float32 sums -> wave shuffles -> LDS -> atomics)
// instead of the two bias_act gradient launches and the bias reduction.  Channels-last float16 / bfloat16, C in {64,128,256,512}.
#include "mg_common.h"
#include "mg_act.h"

namespace {

constexpr int RS_U = 4;         // 16-byte vectors in flight per lane and stream

struct RsArgs {
    const void* x;      // [pixels, C] raw output of conv1
    const void* b;      // [C] or NULL
    const void* r;      // [pixels, C] raw output of the skip convolution
    void* out;          // [pixels, C]
    const void* dout;   // backward: incoming gradient
    void* dx;           // backward: gradient w.r.t. x_raw
    void* dr;           // backward: gradient w.r.t. y_raw (or NULL)
    float* gb;          // backward: [C] float32, zero-filled (or NULL)
    int c;
    uint32_t pixels, pix_per_wg;
    float alpha, gain, clamp, rgain;
};

template <class T> struct alignas(16) RsVec { T v[8]; };

template <class T>
__device__ __forceinline__ void rs_bias(const RsArgs& p, int chunk, float (&bv)[8])
{
    #pragma unroll
    for (int e = 0; e < 8; e++) bv[e] = 0.f;
    if (p.b) {
        const RsVec<T> q = ((const RsVec<T>*)p.b)[chunk];
        #pragma unroll
        for (int e = 0; e < 8; e++) bv[e] = to_acc<T>(q.v[e]);
    }
}

template <class T, int A>
__global__ void __launch_bounds__(256) residual_fwd_kernel(RsArgs p)
{
    const int lpp = p.c / 8;
    const int chunk = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppi = 256 / lpp;
    const BaConst<float> k = {p.alpha, p.gain, p.clamp}, kr = {0.f, p.rgain, -1.f};
    float bv[8];
    rs_bias<T>(p, chunk, bv);
    const uint32_t lo = blockIdx.x * p.pix_per_wg, hi = min(p.pixels, lo + p.pix_per_wg);
    for (uint32_t pix0 = lo + pl; pix0 < hi; pix0 += ppi * RS_U) {
        RsVec<T> vx[RS_U], vr[RS_U];
        #pragma unroll
        for (int u = 0; u < RS_U; u++) {
            const uint32_t pix = pix0 + u * ppi, pc = pix < hi ? pix : lo;          // (a clamped, valid address: no branch around the loads)
            vx[u] = *(const RsVec<T>*)((const T*)p.x + (size_t)pc * p.c + chunk * 8);
            vr[u] = *(const RsVec<T>*)((const T*)p.r + (size_t)pc * p.c + chunk * 8);
        }
        #pragma unroll
        for (int u = 0; u < RS_U; u++) {
            const uint32_t pix = pix0 + u * ppi;
            if (pix >= hi) continue;
            RsVec<T> o;
            #pragma unroll
            for (int e = 0; e < 8; e++) {
                const float a = to_acc<T>(from_acc<T>(ba_eval<float, A>(0, to_acc<T>(vx[u].v[e]), bv[e], 0.f, 0.f, 1.f, k)));
                const float s = to_acc<T>(from_acc<T>(ba_eval<float, 1>(0, to_acc<T>(vr[u].v[e]), 0.f, 0.f, 0.f, 1.f, kr)));
                float t = a + s;
                asm volatile("" : "+v"(t));                 // the sum of two T tensors, rounded on its own
                o.v[e] = from_acc<T>(t);
            }
            *(RsVec<T>*)((T*)p.out + (size_t)pix * p.c + chunk * 8) = o;
        }
    }
}

template <class T, int A, bool WANT_DR>
__global__ void __launch_bounds__(256) residual_bwd_kernel(RsArgs p)
{
    const int lpp = p.c / 8;
    const int chunk = threadIdx.x % lpp, pl = threadIdx.x / lpp, ppi = 256 / lpp;
    const BaConst<float> k = {p.alpha, p.gain, p.clamp}, kr = {0.f, p.rgain, -1.f};
    float bv[8], gb[8];
    rs_bias<T>(p, chunk, bv);
    #pragma unroll
    for (int e = 0; e < 8; e++) gb[e] = 0.f;
    const uint32_t lo = blockIdx.x * p.pix_per_wg, hi = min(p.pixels, lo + p.pix_per_wg);
    for (uint32_t pix0 = lo + pl; pix0 < hi; pix0 += ppi * RS_U) {
        RsVec<T> vd[RS_U], vx[RS_U];
        #pragma unroll
        for (int u = 0; u < RS_U; u++) {
            const uint32_t pix = pix0 + u * ppi, pc = pix < hi ? pix : lo;
            vd[u] = *(const RsVec<T>*)((const T*)p.dout + (size_t)pc * p.c + chunk * 8);
            vx[u] = *(const RsVec<T>*)((const T*)p.x + (size_t)pc * p.c + chunk * 8);
        }
        #pragma unroll
        for (int u = 0; u < RS_U; u++) {
            const uint32_t pix = pix0 + u * ppi;
            const bool live = pix < hi;
            RsVec<T> ox, orr;
            #pragma unroll
            for (int e = 0; e < 8; e++) {
                const float d = to_acc<T>(vd[u].v[e]);
                // the activation's output as the forward pass rounded it: what bias_act's backward would have read back
                const float a = to_acc<T>(from_acc<T>(ba_eval<float, A>(0, to_acc<T>(vx[u].v[e]), bv[e], 0.f, 0.f, 1.f, k)));
                // (linear: no saved output in the API, nothing masked -- reference bias_act.py:26 `ref=''`)
                const float g = ba_eval<float, A>(1, d, 0.f, 0.f, A == 1 ? 0.f : a, 1.f, k);
                ox.v[e] = from_acc<T>(g);
                gb[e] += live ? (float)to_acc<T>(ox.v[e]) : 0.f;
                if (WANT_DR) orr.v[e] = from_acc<T>(ba_eval<float, 1>(1, d, 0.f, 0.f, 0.f, 1.f, kr));
            }
            if (live) {
                *(RsVec<T>*)((T*)p.dx + (size_t)pix * p.c + chunk * 8) = ox;
                if (WANT_DR) *(RsVec<T>*)((T*)p.dr + (size_t)pix * p.c + chunk * 8) = orr;
            }
        }
    }
    if (!p.gb) return;
    #pragma unroll
    for (int e = 0; e < 8; e++)
        for (int o = lpp; o < 64; o <<= 1) gb[e] += __shfl_xor(gb[e], o, 64);
    __shared__ float red[4][64][9];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < lpp) {
        #pragma unroll
        for (int e = 0; e < 8; e++) red[wave][lane][e] = gb[e];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < lpp * 8; idx += 256) {
        const int ch = idx / 8, e = idx - ch * 8;
        atomicAdd(p.gb + ch * 8 + e, red[0][ch][e] + red[1][ch][e] + red[2][ch][e] + red[3][ch][e]);
    }
}

int check_residual(const RsArgs& p, int dtype, int act)
{
    MG_REQUIRE(dtype == MG_F16 || dtype == MG_BF16, "residual_bias_act: float16 or bfloat16 tensors only");
    MG_REQUIRE(act == 1 || act == 3, "residual_bias_act: activation must be linear (1) or lrelu (3)");
    MG_REQUIRE(p.c == 64 || p.c == 128 || p.c == 256 || p.c == 512, "residual_bias_act: 64, 128, 256 or 512 channels");
    MG_REQUIRE((((uintptr_t)p.x | (uintptr_t)p.b | (uintptr_t)p.r | (uintptr_t)p.out | (uintptr_t)p.dout | (uintptr_t)p.dx | (uintptr_t)p.dr) & 15) == 0,
               "residual_bias_act: pointers must be 16-byte aligned");
    return MG_OK;
}

void rs_grid(RsArgs& p, dim3& grid)
{
    const uint32_t ppi = 256u / (uint32_t)(p.c / 8), unit = ppi * RS_U;
    uint32_t wgs = 2048u;
    uint32_t per = ((p.pixels + wgs - 1) / wgs + unit - 1) / unit * unit;
    if (per < unit * 4u) per = unit * 4u;
    wgs = (p.pixels + per - 1) / per;
    p.pix_per_wg = per;
    grid = dim3(wgs);
}

} // namespace

extern "C" int mg_residual_bias_act_fwd(const void* x, const void* b, const void* r, void* out, int dtype, int act, float alpha, float gain, float clamp,
                                        float r_gain, int64_t pixels, int32_t c, void* stream)
{
    MG_REQUIRE(x && r && out, "residual_bias_act_fwd: null pointer");
    MG_REQUIRE(pixels >= 0 && pixels * c < (int64_t)1 << 40 && pixels < (int64_t)UINT32_MAX / 2, "residual_bias_act: too large");
    RsArgs p = {};
    p.x = x; p.b = b; p.r = r; p.out = out; p.c = c; p.pixels = (uint32_t)pixels; p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.rgain = r_gain;
    if (int rc = check_residual(p, dtype, act)) return rc;
    if (pixels == 0) return MG_OK;
    dim3 grid; rs_grid(p, grid);
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_RESIDUAL, st, (double)pixels * c * 2.0 * 3.0);
    if (dtype == MG_BF16) { if (act == 3) hipLaunchKernelGGL((residual_fwd_kernel<mg_bf16, 3>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((residual_fwd_kernel<mg_bf16, 1>), grid, dim3(256), 0, st, p); }
    else                  { if (act == 3) hipLaunchKernelGGL((residual_fwd_kernel<mg_half, 3>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((residual_fwd_kernel<mg_half, 1>), grid, dim3(256), 0, st, p); }
    return mg_check_launch(MG_K_RESIDUAL, "residual_bias_act_fwd");
}

extern "C" int mg_residual_bias_act_bwd(const void* dout, const void* x, const void* b, void* dx, void* dr, float* gb32, int dtype, int act, float alpha,
                                        float gain, float clamp, float r_gain, int64_t pixels, int32_t c, void* stream)
{
    MG_REQUIRE(dout && x && dx, "residual_bias_act_bwd: null pointer");
    MG_REQUIRE(pixels >= 0 && pixels * c < (int64_t)1 << 40 && pixels < (int64_t)UINT32_MAX / 2, "residual_bias_act: too large");
    RsArgs p = {};
    p.dout = dout; p.x = x; p.b = b; p.dx = dx; p.dr = dr; p.gb = gb32; p.c = c; p.pixels = (uint32_t)pixels;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.rgain = r_gain;
    if (int rc = check_residual(p, dtype, act)) return rc;
    if (pixels == 0) return MG_OK;
    dim3 grid; rs_grid(p, grid);
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_RESIDUAL, st, (double)pixels * c * 2.0 * (dr ? 4.0 : 3.0));
#define RS_BWD(T, A) do { if (dr) hipLaunchKernelGGL((residual_bwd_kernel<T, A, true>), grid, dim3(256), 0, st, p); \
                          else hipLaunchKernelGGL((residual_bwd_kernel<T, A, false>), grid, dim3(256), 0, st, p); } while (0)
    if (dtype == MG_BF16) { if (act == 3) RS_BWD(mg_bf16, 3); else RS_BWD(mg_bf16, 1); }
    else                  { if (act == 3) RS_BWD(mg_half, 3); else RS_BWD(mg_half, 1); }
#undef RS_BWD
    return mg_check_launch(MG_K_RESIDUAL, "residual_bias_act_bwd");
}
