// fused.hip -- fusions the plugin API permits without changing what callers compute.
//
//  * mg_fma_bias_act (backward: mg_bwd_fused below): the modulated-conv epilogue of SynthesisLayer,
//      x = fma(x, dcoefs[N,C,1,1], noise[1,1,H,W]) ; x = bias_act(x, b, act, gain, clamp)
//    (reference training/networks.py:86-88 + 311-313 -> fma.py:16, bias_act.py:71)
//    as ONE read and ONE write of the activation instead of three round trips.
//  * mg_filtered_lrelu: upsample-FIR -> bias+lrelu(+clamp) -> downsample-FIR with the
//    up-sampled intermediate living in LDS only.  Absent upstream; defined as
//    the composition of the reference's own ops (upfirdn2d -> bias_act -> upfirdn2d).
#include "mg_common.h"
#include "mg_act.h"

namespace {

template <class T> struct FbArgs {
    const T* x; const float* d; const float* noise; const T* b; T* y; T* y2;
    float alpha, gain, clamp, noise_gain;
    const float* noise_gain_dev;
    int64_t noise_stride_n;
    uint32_t c, hw, total_vec;
    FastDiv hwv_div, c_div;     // hw/VEC and C
};

// forward.  One lane = one 16-byte vector inside one (n,c) plane (hw % VEC == 0) or one element.
// U vectors (and their noise) are requested before any is consumed; grid-stride over chunks.
constexpr int FB_U = 4;

template <class T, int A, int VEC>
__global__ void __launch_bounds__(256) fma_bias_act_fwd_kernel(FbArgs<T> p)
{
    struct alignas(sizeof(T) * VEC) V { T v[VEC]; };
    typedef float f4_t __attribute__((ext_vector_type(4)));
    struct NV { float v[VEC]; };
    const BaConst<float> k = {p.alpha, p.gain, p.clamp};
    const uint32_t hwv = p.hw / VEC;
    const float ngain = p.noise_gain * ((p.noise && p.noise_gain_dev) ? *p.noise_gain_dev : 1.f);
    const bool nz_vec = (VEC == 1) || ((((uintptr_t)p.noise) & 15) == 0 && ((p.noise_stride_n * 4) & 15) == 0);
    for (uint32_t base = blockIdx.x * (256u * FB_U); base < p.total_vec; base += gridDim.x * (256u * FB_U)) {
        V xin[FB_U];
        NV nzv[FB_U];
        uint32_t plane[FB_U];
        #pragma unroll
        for (int u = 0; u < FB_U; u++) {
            const uint32_t vi = base + u * 256u + threadIdx.x;
            if (vi >= p.total_vec) continue;
            plane[u] = fd_div(vi, p.hwv_div);
            xin[u] = ((const V*)p.x)[vi];
            if (p.noise) {
                const uint32_t pv = vi - plane[u] * hwv, n = fd_div(plane[u], p.c_div);
                const float* nz = p.noise + (int64_t)n * p.noise_stride_n + (size_t)pv * VEC;
                if (VEC >= 4 && nz_vec) {
                    #pragma unroll
                    for (int q = 0; q < VEC / 4; q++) {             // explicit 16-byte loads
                        const f4_t t = ((const f4_t*)nz)[q];
                        nzv[u].v[4 * q] = t.x; nzv[u].v[4 * q + 1] = t.y; nzv[u].v[4 * q + 2] = t.z; nzv[u].v[4 * q + 3] = t.w;
                    }
                } else {
                    #pragma unroll
                    for (int e = 0; e < VEC; e++) nzv[u].v[e] = nz[e];
                }
            }
        }
        #pragma unroll
        for (int u = 0; u < FB_U; u++) {
            const uint32_t vi = base + u * 256u + threadIdx.x;
            if (vi >= p.total_vec) continue;
            const float dc = p.d ? (float)to_acc<T>(from_acc<T>(p.d[plane[u]])) : 1.f;      // dcoefs.to(x.dtype), as the reference multiplies
            float bb = 0.f;
            if (p.b) { const uint32_t n = fd_div(plane[u], p.c_div); bb = to_acc<T>(p.b[plane[u] - n * p.c]); }
            V out;
            #pragma unroll
            for (int e = 0; e < VEC; e++) {
                float v = to_acc<T>(xin[u].v[e]) * dc;
                if (p.noise) v = __builtin_fmaf(nzv[u].v[e], ngain, v);        // one rounding, in both layouts
                // the reference rounds the fma result to T before bias_act reads it back
                v = to_acc<T>(from_acc<T>(v));
                out.v[e] = from_acc<T>(ba_eval<float, A>(0, v, bb, 0.f, 0.f, 1.f, k));
            }
            ((V*)p.y)[vi] = out;
        }
    }
}

template <class T, int A>
void launch_fb(const FbArgs<T>& p0, bool wide, hipStream_t s)
{
    constexpr int VEC = 16 / sizeof(T);
    FbArgs<T> p = p0;
    int vec = wide ? VEC : 1;
    p.hwv_div = make_fastdiv(p.hw / vec);
    p.total_vec = p.total_vec / vec;
    uint32_t blocks = (p.total_vec + 256u * FB_U - 1) / (256u * FB_U);
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    if (blocks < 1) blocks = 1;
    if (wide) hipLaunchKernelGGL((fma_bias_act_fwd_kernel<T, A, VEC>), dim3(blocks), dim3(256), 0, s, p);
    else      hipLaunchKernelGGL((fma_bias_act_fwd_kernel<T, A, 1>), dim3(blocks), dim3(256), 0, s, p);
}

template <class T>
int dispatch_fb(const FbArgs<T>& p, int act, bool wide, hipStream_t s)
{
    switch (act) {
    case 1: launch_fb<T, 1>(p, wide, s); return MG_OK;
    case 2: launch_fb<T, 2>(p, wide, s); return MG_OK;
    case 3: launch_fb<T, 3>(p, wide, s); return MG_OK;
    case 4: launch_fb<T, 4>(p, wide, s); return MG_OK;
    case 5: launch_fb<T, 5>(p, wide, s); return MG_OK;
    case 6: launch_fb<T, 6>(p, wide, s); return MG_OK;
    case 7: launch_fb<T, 7>(p, wide, s); return MG_OK;
    case 8: launch_fb<T, 8>(p, wide, s); return MG_OK;
    }
    mg_set_error("fma_bias_act: activation %d needs the pre-activation input and is not fusable", act);
    return MG_E_ARG;
}

// ---------------------------------------------------------------------------
// filtered_lrelu: one workgroup = one (plane, output tile).  Runtime filter sizes
// (<= FL_MAXF taps per axis), up/down in {1,2}.  Three LDS stages:
//   s_in  : input footprint (+bias), fp32, zero outside the image
//   s_mid : up-sampled, filtered, activated intermediate (the tensor that never hits HBM)
//   out   : decimating FIR of s_mid, straight to global
constexpr int FL_TW = 32, FL_TH = 32, FL_MAXF = 12;
constexpr int FL_MIDW = FL_TW * 2 + FL_MAXF - 1, FL_MIDH = FL_TH * 2 + FL_MAXF - 1;   // worst case down=2


struct FlArgs {
    const void* x; const void* b; const float* fu; const float* fd; void* y;
    int n, c, in_h, in_w, out_h, out_w, fu_h, fu_w, fd_h, fd_w, up, down, padx0, pady0, flip;
    float alpha, gain, clamp;
    int tiles_x, tiles_y;
};

template <class T>
__global__ void __launch_bounds__(256) filtered_lrelu_kernel(FlArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_fu = smem;                              // [fu_h*fu_w] flipped
    float* s_fd = s_fu + FL_MAXF * FL_MAXF;          // [fd_h*fd_w] flipped
    float* s_in = s_fd + FL_MAXF * FL_MAXF;
    const int tile_x = blockIdx.x % p.tiles_x, tile_y = (blockIdx.x / p.tiles_x) % p.tiles_y;
    const int plane = blockIdx.x / (p.tiles_x * p.tiles_y);
    const int ch = plane % p.c;
    const int up = p.up, down = p.down;
    // intermediate region needed by this output tile
    const int mid_x0 = tile_x * FL_TW * down, mid_y0 = tile_y * FL_TH * down;
    const int tw = min(FL_TW, p.out_w - tile_x * FL_TW), th = min(FL_TH, p.out_h - tile_y * FL_TH);
    const int mid_w = (tw - 1) * down + p.fd_w, mid_h = (th - 1) * down + p.fd_h;
    // input region needed by that intermediate region (same index path as upfirdn2d)
    const int in_x0 = mg_floor_div(mid_x0 + up - 1 - p.padx0, up), in_y0 = mg_floor_div(mid_y0 + up - 1 - p.pady0, up);
    const int in_x1 = mg_floor_div(mid_x0 + mid_w - 1 + up - 1 - p.padx0 + p.fu_w - 1, up);
    const int in_y1 = mg_floor_div(mid_y0 + mid_h - 1 + up - 1 - p.pady0 + p.fu_h - 1, up);
    const int in_w = in_x1 - in_x0 + 1, in_h = in_y1 - in_y0 + 1;
    float* s_mid = s_in + in_w * in_h;

    for (int i = threadIdx.x; i < p.fu_h * p.fu_w; i += 256) {
        int ky = i / p.fu_w, kx = i - ky * p.fu_w;
        s_fu[i] = p.fu[(p.flip ? ky : p.fu_h - 1 - ky) * p.fu_w + (p.flip ? kx : p.fu_w - 1 - kx)];
    }
    for (int i = threadIdx.x; i < p.fd_h * p.fd_w; i += 256) {
        int ky = i / p.fd_w, kx = i - ky * p.fd_w;
        s_fd[i] = p.fd[(p.flip ? ky : p.fd_h - 1 - ky) * p.fd_w + (p.flip ? kx : p.fd_w - 1 - kx)];
    }
    const float bias = p.b ? (float)to_acc<T>(((const T*)p.b)[ch]) : 0.f;
    const T* xp = (const T*)p.x + (size_t)plane * p.in_h * p.in_w;
    for (int i = threadIdx.x; i < in_w * in_h; i += 256) {
        int ry = i / in_w, rx = i - ry * in_w, ix = in_x0 + rx, iy = in_y0 + ry;
        float v = 0.f;
        if (ix >= 0 && iy >= 0 && ix < p.in_w && iy < p.in_h) v = (float)to_acc<T>(xp[(size_t)iy * p.in_w + ix]) + bias;
        s_in[i] = v;
    }
    __syncthreads();
    const float up_gain = (float)(up * up);
    for (int i = threadIdx.x; i < mid_w * mid_h; i += 256) {
        int my = i / mid_w, mx = i - my * mid_w;
        int midx = mid_x0 + mx + up - 1 - p.padx0, midy = mid_y0 + my + up - 1 - p.pady0;
        int ix = mg_floor_div(midx, up), iy = mg_floor_div(midy, up);
        int fx0 = (ix + 1) * up - midx - 1, fy0 = (iy + 1) * up - midy - 1;
        float v = 0.f;
        for (int ky = fy0, ry = iy - in_y0; ky < p.fu_h; ky += up, ry++)
            for (int kx = fx0, rx = ix - in_x0; kx < p.fu_w; kx += up, rx++)
                v += s_in[ry * in_w + rx] * s_fu[ky * p.fu_w + kx];
        v *= up_gain;
        // the composition rounds the up-sampled tensor to T, then bias_act (bias already added) rounds again
        v = (float)to_acc<T>(from_acc<T>(v));
        v = (v > 0.f ? v : v * p.alpha) * p.gain;
        if (p.clamp >= 0.f) v = fminf(fmaxf(v, -p.clamp), p.clamp);
        s_mid[i] = (float)to_acc<T>(from_acc<T>(v));
    }
    __syncthreads();
    T* yp = (T*)p.y + (size_t)plane * p.out_h * p.out_w;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        int oy = i / tw, ox = i - oy * tw;
        float v = 0.f;
        for (int ky = 0; ky < p.fd_h; ky++)
            for (int kx = 0; kx < p.fd_w; kx++)
                v += s_mid[(oy * down + ky) * mid_w + (ox * down + kx)] * s_fd[ky * p.fd_w + kx];
        yp[(size_t)(tile_y * FL_TH + oy) * p.out_w + (tile_x * FL_TW + ox)] = from_acc<T>(v);
    }
}


// ---------------------------------------------------------------------------
// Backward with fused reductions (mg_bwd_fused).  One workgroup = one slice of one (n,c) plane, so the
// three sums it owns are a wave-shuffle + LDS reduction followed by ONE float atomic each.
template <class T> struct BrArgs {
    const T* dy; const T* yref; const T* x; const float* d; const float* noise; T* dx;
    float* gd; float* gb; float* gs;
    float alpha, gain, clamp;
    int64_t noise_stride_n;
    uint32_t c, hw, slices, slice_len;      // slice_len: elements per slice, a multiple of 256*VEC
};

template <class T, int A, int VEC>
__global__ void __launch_bounds__(256) bwd_fused_kernel(BrArgs<T> p)
{
    struct alignas(sizeof(T) * VEC) V { T v[VEC]; };
    const BaConst<float> k = {p.alpha, p.gain, p.clamp};
    const uint32_t plane = blockIdx.x / p.slices, slice = blockIdx.x - plane * p.slices;
    const uint32_t n = plane / p.c, ch = plane - n * p.c;
    const size_t base = (size_t)plane * p.hw;
    const uint32_t lo = slice * p.slice_len, hi = min(p.hw, lo + p.slice_len);
    const float dc = p.d ? (float)to_acc<T>(from_acc<T>(p.d[plane])) : 1.f;        // dcoefs.to(x.dtype)
    const float* nz = p.noise ? p.noise + (int64_t)n * p.noise_stride_n : nullptr;
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const bool nz_vec = (VEC == 1) || ((((uintptr_t)p.noise) & 15) == 0 && ((p.noise_stride_n * 4) & 15) == 0);
    float s_gd = 0.f, s_gb = 0.f, s_gs = 0.f;
    constexpr int BU = 4;                                           // vectors in flight per lane
    for (uint32_t i0 = lo + threadIdx.x * VEC; i0 < hi; i0 += 256u * VEC * BU) {
        // hw % VEC == 0 on the vector path, so a vector never crosses the end of the plane
        V vdy[BU], vy[BU], vx[BU];
        float nzv[BU][VEC];
        #pragma unroll
        for (int u = 0; u < BU; u++) {
            const uint32_t i = i0 + u * 256u * VEC;
            if (i >= hi) continue;
            vdy[u] = *(const V*)(p.dy + base + i);
            if (p.yref) vy[u] = *(const V*)(p.yref + base + i);
            if (p.x) vx[u] = *(const V*)(p.x + base + i);
            if (nz) {
                if (VEC >= 4 && nz_vec) {
                    #pragma unroll
                    for (int q = 0; q < VEC / 4; q++) {
                        const f4_t t = ((const f4_t*)(nz + i))[q];
                        nzv[u][4 * q] = t.x; nzv[u][4 * q + 1] = t.y; nzv[u][4 * q + 2] = t.z; nzv[u][4 * q + 3] = t.w;
                    }
                } else {
                    #pragma unroll
                    for (int e = 0; e < VEC; e++) nzv[u][e] = nz[i + e];
                }
            }
        }
        #pragma unroll
        for (int u = 0; u < BU; u++) {
            const uint32_t i = i0 + u * 256u * VEC;
            if (i >= hi) continue;
            V out;
            #pragma unroll
            for (int e = 0; e < VEC; e++) {
                float g = to_acc<T>(vdy[u].v[e]);
                if (p.yref) g = ba_eval<float, A>(1, g, 0.f, 0.f, to_acc<T>(vy[u].v[e]), 1.f, k);
                const float gq = to_acc<T>(from_acc<T>(g));        // autograd materialises dpre in T
                out.v[e] = from_acc<T>(gq * dc);
                s_gb += gq;
                if (p.x) s_gd += gq * to_acc<T>(vx[u].v[e]);
                if (nz) s_gs += gq * nzv[u][e];
            }
            *(V*)(p.dx + base + i) = out;
        }
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s_gd += __shfl_down(s_gd, o, 64); s_gb += __shfl_down(s_gb, o, 64); s_gs += __shfl_down(s_gs, o, 64);
    }
    __shared__ float red[3][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = s_gd; red[1][wv] = s_gb; red[2][wv] = s_gs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (p.gd) atomicAdd(p.gd + plane, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        if (p.gb) atomicAdd(p.gb + ch, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        if (p.gs) atomicAdd(p.gs + (blockIdx.x % MG_GS_SLOTS), red[2][0] + red[2][1] + red[2][2] + red[2][3]);   // spread: one address would serialise
    }
}

// Small planes (the 16x16 ... 4x4 float32 blocks: 16 k planes of 256 / 64 / 16 elements): one WAVE per plane, four
// planes per workgroup, wave-shuffle reductions only -- no LDS, no barrier, and four times fewer workgroups than
// the plane-per-workgroup kernel above, which leaves 3/4 (or more) of its lanes idle on such planes.
template <class T, int A, int VEC>
__global__ void __launch_bounds__(256) bwd_fused_small_kernel(BrArgs<T> p, uint32_t planes)
{
    struct alignas(sizeof(T) * VEC) V { T v[VEC]; };
    const BaConst<float> k = {p.alpha, p.gain, p.clamp};
    const uint32_t lane = threadIdx.x & 63, plane = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (plane >= planes) return;                                    // wave-uniform
    const uint32_t n = plane / p.c, ch = plane - n * p.c;
    const size_t base = (size_t)plane * p.hw;
    const float dc = p.d ? (float)to_acc<T>(from_acc<T>(p.d[plane])) : 1.f;
    const float* nz = p.noise ? p.noise + (int64_t)n * p.noise_stride_n : nullptr;
    float s_gd = 0.f, s_gb = 0.f, s_gs = 0.f;
    for (uint32_t i = lane * VEC; i < p.hw; i += 64u * VEC) {
        const V vdy = *(const V*)(p.dy + base + i);
        V vy, vx, out;
        if (p.yref) vy = *(const V*)(p.yref + base + i);
        if (p.x) vx = *(const V*)(p.x + base + i);
        #pragma unroll
        for (int e = 0; e < VEC; e++) {
            float g = to_acc<T>(vdy.v[e]);
            if (p.yref) g = ba_eval<float, A>(1, g, 0.f, 0.f, to_acc<T>(vy.v[e]), 1.f, k);
            const float gq = to_acc<T>(from_acc<T>(g));
            out.v[e] = from_acc<T>(gq * dc);
            s_gb += gq;
            if (p.x) s_gd += gq * to_acc<T>(vx.v[e]);
            if (nz) s_gs += gq * nz[i + e];
        }
        *(V*)(p.dx + base + i) = out;
    }
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s_gd += __shfl_down(s_gd, o, 64); s_gb += __shfl_down(s_gb, o, 64); s_gs += __shfl_down(s_gs, o, 64);
    }
    if (lane == 0) {
        if (p.gd) atomicAdd(p.gd + plane, s_gd);
        if (p.gb) atomicAdd(p.gb + ch, s_gb);
        if (p.gs) atomicAdd(p.gs + (blockIdx.x % MG_GS_SLOTS), s_gs);
    }
}

template <class T, int A>
void launch_br(BrArgs<T> p, bool wide, uint32_t planes, hipStream_t s)
{
    constexpr int VEC = 16 / sizeof(T);
    const uint32_t vec = wide ? VEC : 1;
    if (p.hw <= 64u * vec * 4u && planes >= 1024u) {                // small planes, many of them: one wave per plane
        const uint32_t blocks = (planes + 3u) / 4u;
        if (wide) hipLaunchKernelGGL((bwd_fused_small_kernel<T, A, VEC>), dim3(blocks), dim3(256), 0, s, p, planes);
        else      hipLaunchKernelGGL((bwd_fused_small_kernel<T, A, 1>), dim3(blocks), dim3(256), 0, s, p, planes);
        return;
    }
    const uint32_t unit = 256u * vec;
    uint32_t slices = (p.hw + unit * 8 - 1) / (unit * 8);          // about 8 vectors per lane per slice (2 rounds of 4 in flight)
    if (slices < 1) slices = 1;
    if (slices > 64) slices = 64;
    uint32_t len = (p.hw + slices - 1) / slices;
    len = (len + unit - 1) / unit * unit;
    slices = (p.hw + len - 1) / len;
    p.slices = slices; p.slice_len = len;
    if (wide) hipLaunchKernelGGL((bwd_fused_kernel<T, A, VEC>), dim3(planes * slices), dim3(256), 0, s, p);
    else      hipLaunchKernelGGL((bwd_fused_kernel<T, A, 1>), dim3(planes * slices), dim3(256), 0, s, p);
}

template <class T>
int dispatch_br(const BrArgs<T>& p, int act, bool wide, uint32_t planes, hipStream_t s)
{
    switch (act) {
    case 1: launch_br<T, 1>(p, wide, planes, s); return MG_OK;
    case 2: launch_br<T, 2>(p, wide, planes, s); return MG_OK;
    case 3: launch_br<T, 3>(p, wide, planes, s); return MG_OK;
    }
    mg_set_error("bwd_fused: only linear / relu / lrelu are fused (activation %d)", act);
    return MG_E_ARG;
}

// ---------------------------------------------------------------------------
// Channels-last (NHWC) variants of the epilogue and of its fused backward: the same arithmetic on the layout the
// implicit-GEMM convolutions produce, so a bf16 block never has to be transposed back to NCHW.  One lane = one
// 16-byte vector of VEC consecutive channels of one pixel; dcoefs and the bias arrive as small aligned vectors
// (L1/L2 resident), the noise as one scalar per pixel.
template <class T> struct FnArgs {
    const T* x; const float* d; const float* noise; const T* b; T* y;
    float alpha, gain, clamp, noise_gain;
    const float* noise_gain_dev;
    int64_t noise_stride_n;
    uint32_t c, hw, cv, total_vec;
    FastDiv cv_div, hw_div;
};

// HD / HN / HB: dcoefs / noise / bias present -- compile-time, and FULL chunks carry no bounds test, so the FB_U activation
// vectors of a chunk and their per-pixel noise samples are requested back to back (tested at run time, every load sat behind
// its own branch and `s_waitcnt vmcnt(0)`).
template <class T, int A, bool HD, bool HN, bool HB, bool FULL>
__device__ __forceinline__ void fn_chunk(const FnArgs<T>& p, const BaConst<float>& k, float ngain, uint32_t base)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    struct alignas(16) V { T v[VEC]; };
    typedef float f4_t __attribute__((ext_vector_type(4)));
    V xin[FB_U], bv[FB_U];
    f4_t dq[FB_U][VEC / 4];
    float nz[FB_U];
    #pragma unroll
    for (int u = 0; u < FB_U; u++) {
        const uint32_t vi = base + u * 256u + threadIdx.x;
        if (!FULL && vi >= p.total_vec) continue;
        xin[u] = ((const V*)p.x)[vi];
        const uint32_t pixn = fd_div(vi, p.cv_div), cvi = vi - pixn * p.cv;
        const uint32_t n = fd_div(pixn, p.hw_div), pix = pixn - n * p.hw;
        nz[u] = HN ? p.noise[(int64_t)n * p.noise_stride_n + pix] : 0.f;
        if (HB) bv[u] = *(const V*)(p.b + cvi * VEC);
        if (HD) {
            const f4_t* dp = (const f4_t*)(p.d + (size_t)n * p.c + cvi * VEC);
            #pragma unroll
            for (int q = 0; q < VEC / 4; q++) dq[u][q] = dp[q];
        }
    }
    #pragma unroll
    for (int u = 0; u < FB_U; u++) {
        const uint32_t vi = base + u * 256u + threadIdx.x;
        if (!FULL && vi >= p.total_vec) continue;
        V out;
        #pragma unroll
        for (int e = 0; e < VEC; e++) {
            const float de = HD ? (float)to_acc<T>(from_acc<T>(dq[u][e / 4][e % 4])) : 1.f;       // dcoefs.to(x.dtype)
            float v = to_acc<T>(xin[u].v[e]) * de;
            if (HN) v = __builtin_fmaf(nz[u], ngain, v);
            // (a float32 value of its own, as in the NCHW kernel, whose run-time noise test keeps the pieces apart: with HN known
            // the compiler would contract a float32 product with the bias add inside ba_eval into one fma, and fuse the float16
            // rounding below into the fma -- v_fma_mixlo_f16, ONE rounding where the NCHW kernel and the reference round twice)
            asm volatile("" : "+v"(v));
            v = to_acc<T>(from_acc<T>(v));                      // the reference rounds the fma result to T
            out.v[e] = from_acc<T>(ba_eval<float, A>(0, v, HB ? (float)to_acc<T>(bv[u].v[e]) : 0.f, 0.f, 0.f, 1.f, k));
        }
        ((V*)p.y)[vi] = out;
    }
}

template <class T, int A, bool HD, bool HN, bool HB>
__global__ void __launch_bounds__(256) fma_bias_act_nhwc_fwd_kernel(FnArgs<T> p)
{
    const BaConst<float> k = {p.alpha, p.gain, p.clamp};
    const float ngain = p.noise_gain * ((HN && p.noise_gain_dev) ? *p.noise_gain_dev : 1.f);
    for (uint32_t base = blockIdx.x * (256u * FB_U); base < p.total_vec; base += gridDim.x * (256u * FB_U)) {
        if (base + 256u * FB_U <= p.total_vec) fn_chunk<T, A, HD, HN, HB, true>(p, k, ngain, base);
        else fn_chunk<T, A, HD, HN, HB, false>(p, k, ngain, base);
    }
}

template <class T, int A>
void launch_fn(const FnArgs<T>& f, uint32_t blocks, hipStream_t s)
{
    const int key = (f.d ? 4 : 0) | (f.noise ? 2 : 0) | (f.b ? 1 : 0);
    switch (key) {
    case 0: hipLaunchKernelGGL((fma_bias_act_nhwc_fwd_kernel<T, A, false, false, false>), dim3(blocks), dim3(256), 0, s, f); break;
    case 1: hipLaunchKernelGGL((fma_bias_act_nhwc_fwd_kernel<T, A, false, false, true>), dim3(blocks), dim3(256), 0, s, f); break;
    case 2: hipLaunchKernelGGL((fma_bias_act_nhwc_fwd_kernel<T, A, false, true, false>), dim3(blocks), dim3(256), 0, s, f); break;
    case 3: hipLaunchKernelGGL((fma_bias_act_nhwc_fwd_kernel<T, A, false, true, true>), dim3(blocks), dim3(256), 0, s, f); break;
    case 4: hipLaunchKernelGGL((fma_bias_act_nhwc_fwd_kernel<T, A, true, false, false>), dim3(blocks), dim3(256), 0, s, f); break;
    case 5: hipLaunchKernelGGL((fma_bias_act_nhwc_fwd_kernel<T, A, true, false, true>), dim3(blocks), dim3(256), 0, s, f); break;
    case 6: hipLaunchKernelGGL((fma_bias_act_nhwc_fwd_kernel<T, A, true, true, false>), dim3(blocks), dim3(256), 0, s, f); break;
    default: hipLaunchKernelGGL((fma_bias_act_nhwc_fwd_kernel<T, A, true, true, true>), dim3(blocks), dim3(256), 0, s, f); break;
    }
}

template <class T> struct BnArgs {
    const T* dy; const T* yref; const T* x; const float* d; const float* noise; T* dx;
    float* gd; float* gb; float* gs;       // gd [N*C]; gb [N*C] (per-sample partial sums); gs [1]
    float alpha, gain, clamp;
    int64_t noise_stride_n;
    uint32_t c, hw, cv, slices, slice_pix;
    // `recon`: the epilogue's input x is not read but rebuilt from its saved output y (what gd sums over):
    //     pre = y / gain  (/ alpha where negative, lrelu),   x = (pre - noise * strength - b) / d
    // exact up to the rounding of y to T; where the clamp saturated the gradient is zero and x does not matter.
    const T* bias; const float* strength; int recon;
};

// One workgroup = one slice of the pixels of one sample.  256 % cv == 0, so a lane keeps the same channel vector
// for the whole slice and its VEC running sums per reduction stay in registers; they meet in LDS at the end.
//
// What is read is a compile-time property (YREF: the saved output; XM: 0 nothing / 1 the epilogue's input x / 2 x rebuilt from
// the saved output; NZ: the noise plane), and a slice that lies inside the sample runs rounds of BU pixels per lane with no
// bounds test: tested at run time (`if (p.yref) ... = load`), every load of the round sat in its own basic block behind a
// `s_waitcnt vmcnt(0)` -- 180 branches and one load in flight per lane (round 3: 3.7 TB/s of algorithmic bytes; the loads of a
// round now issue back to back).  FULL = false is the same body with the bounds tests, for the last slice of a ragged sample.
template <class T, int A, int YREF, int XM, int NZ, bool FULL, int BU>
__device__ __forceinline__ void bn_round(const BnArgs<T>& p, const BaConst<float>& k, size_t base, uint32_t pix0, uint32_t ppi, uint32_t hi,
                                         const float* nzp, const float (&dv)[16 / sizeof(T)], const float (&bv)[16 / sizeof(T)],
                                         const float (&rdv)[16 / sizeof(T)], float rstr, float rgain, float ralpha,
                                         float (&s_gd)[16 / sizeof(T)], float (&s_gb)[16 / sizeof(T)], float& s_gs)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    struct alignas(16) V { T v[VEC]; };
    V vdy[BU], vy[BU], vx[BU];
    float nz[BU];
    #pragma unroll
    for (int u = 0; u < BU; u++) {
        const uint32_t pix = pix0 + u * ppi;
        if (!FULL && pix >= hi) continue;
        const size_t off = base + (size_t)pix * p.c;
        vdy[u] = *(const V*)(p.dy + off);
        if (YREF) vy[u] = *(const V*)(p.yref + off);
        if (XM == 1) vx[u] = *(const V*)(p.x + off);
        nz[u] = NZ ? nzp[pix] : 0.f;
    }
    #pragma unroll
    for (int u = 0; u < BU; u++) {
        const uint32_t pix = pix0 + u * ppi;
        if (!FULL && pix >= hi) continue;
        V out;
        float srow = 0.f;
        #pragma unroll
        for (int e = 0; e < VEC; e++) {
            float g = to_acc<T>(vdy[u].v[e]);
            if (YREF) g = ba_eval<float, A>(1, g, 0.f, 0.f, to_acc<T>(vy[u].v[e]), 1.f, k);
            const float gq = to_acc<T>(from_acc<T>(g));        // autograd materialises dpre in T
            out.v[e] = from_acc<T>(gq * dv[e]);
            s_gb[e] += gq;
            if (XM == 1) s_gd[e] += gq * to_acc<T>(vx[u].v[e]);
            else if (XM == 2) {
                float pre = to_acc<T>(vy[u].v[e]) * rgain;
                if (A == 3 && pre < 0.f) pre *= ralpha;
                s_gd[e] += gq * ((pre - nz[u] * rstr - bv[e]) * rdv[e]);
            }
            srow += gq;
        }
        if (NZ) s_gs += srow * nz[u];
        *(V*)(p.dx + base + (size_t)pix * p.c) = out;
    }
}

template <class T, int A, int YREF, int XM, int NZ>
__global__ void __launch_bounds__(256) bwd_fused_nhwc_kernel(BnArgs<T> p)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    struct alignas(16) V { T v[VEC]; };
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const BaConst<float> k = {p.alpha, p.gain, p.clamp};
    const uint32_t n = blockIdx.x / p.slices, slice = blockIdx.x - n * p.slices;
    const uint32_t pl = threadIdx.x / p.cv, cvi = threadIdx.x - pl * p.cv, ppi = 256u / p.cv;
    const uint32_t lo = slice * p.slice_pix, hi = min(p.hw, lo + p.slice_pix);
    const size_t base = (size_t)n * p.hw * p.c + cvi * VEC;
    const float* nzp = NZ ? p.noise + (int64_t)n * p.noise_stride_n : nullptr;
    float dv[VEC];
    #pragma unroll
    for (int e = 0; e < VEC; e++) dv[e] = 1.f;
    if (p.d) {
        const f4_t* dp = (const f4_t*)(p.d + (size_t)n * p.c + cvi * VEC);
        #pragma unroll
        for (int q = 0; q < VEC / 4; q++) { const f4_t t = dp[q]; dv[4 * q] = t.x; dv[4 * q + 1] = t.y; dv[4 * q + 2] = t.z; dv[4 * q + 3] = t.w; }
        #pragma unroll
        for (int e = 0; e < VEC; e++) dv[e] = to_acc<T>(from_acc<T>(dv[e]));                      // dcoefs.to(x.dtype)
    }
    float s_gd[VEC], s_gb[VEC], s_gs = 0.f;
    #pragma unroll
    for (int e = 0; e < VEC; e++) { s_gd[e] = 0.f; s_gb[e] = 0.f; }
    float bv[VEC], rdv[VEC];                // XM == 2: bias and 1 / d of the lane's channels
    #pragma unroll
    for (int e = 0; e < VEC; e++) { bv[e] = 0.f; rdv[e] = 1.f; }
    const float rstr = (XM == 2 && p.strength) ? *p.strength : 1.f;
    const float rgain = 1.f / p.gain, ralpha = (A == 3 && p.alpha != 0.f) ? 1.f / p.alpha : 0.f;
    if (XM == 2) {
        if (p.bias) {
            const V b16 = *(const V*)(p.bias + cvi * VEC);
            #pragma unroll
            for (int e = 0; e < VEC; e++) bv[e] = to_acc<T>(b16.v[e]);
        }
        #pragma unroll
        for (int e = 0; e < VEC; e++) rdv[e] = 1.f / dv[e];
    }
    constexpr int BU = 4;
    uint32_t pix0 = lo + pl;
    // (slice_pix is a multiple of ppi * BU: inside a whole slice every lane of every round has a pixel)
    if (lo + p.slice_pix <= p.hw)
        for (; pix0 < hi; pix0 += ppi * BU)
            bn_round<T, A, YREF, XM, NZ, true, BU>(p, k, base, pix0, ppi, hi, nzp, dv, bv, rdv, rstr, rgain, ralpha, s_gd, s_gb, s_gs);
    else
        for (; pix0 < hi; pix0 += ppi * BU)
            bn_round<T, A, YREF, XM, NZ, false, BU>(p, k, base, pix0, ppi, hi, nzp, dv, bv, rdv, rstr, rgain, ralpha, s_gd, s_gb, s_gs);
    __shared__ float red[256][VEC + 1];
    const uint32_t C = p.c;
    if (p.gd) {
        #pragma unroll
        for (int e = 0; e < VEC; e++) red[threadIdx.x][e] = s_gd[e];
        __syncthreads();
        for (uint32_t ch = threadIdx.x; ch < C; ch += 256u) {
            const uint32_t cq = ch / VEC, e = ch - cq * VEC;
            float t = 0.f;
            for (uint32_t q = 0; q < ppi; q++) t += red[q * p.cv + cq][e];
            atomicAdd(p.gd + (size_t)n * C + ch, t);
        }
        __syncthreads();
    }
    if (p.gb) {
        #pragma unroll
        for (int e = 0; e < VEC; e++) red[threadIdx.x][e] = s_gb[e];
        __syncthreads();
        for (uint32_t ch = threadIdx.x; ch < C; ch += 256u) {
            const uint32_t cq = ch / VEC, e = ch - cq * VEC;
            float t = 0.f;
            for (uint32_t q = 0; q < ppi; q++) t += red[q * p.cv + cq][e];
            atomicAdd(p.gb + (size_t)n * C + ch, t);
        }
    }
    if (NZ && p.gs) {
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) s_gs += __shfl_down(s_gs, o, 64);
        __shared__ float red_s[4];
        if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = s_gs;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(p.gs + (blockIdx.x % MG_GS_SLOTS), red_s[0] + red_s[1] + red_s[2] + red_s[3]);
    }
}

template <class T, int A, int YREF, int XM>
void launch_bn(const BnArgs<T>& b, uint32_t n, hipStream_t s)
{
    if (b.noise) hipLaunchKernelGGL((bwd_fused_nhwc_kernel<T, A, YREF, XM, 1>), dim3(n * b.slices), dim3(256), 0, s, b);
    else hipLaunchKernelGGL((bwd_fused_nhwc_kernel<T, A, YREF, XM, 0>), dim3(n * b.slices), dim3(256), 0, s, b);
}

template <class T, int A>
void launch_nhwc(const FnArgs<T>* f, const BnArgs<T>* b, uint32_t n, hipStream_t s)
{
    if (f) {
        uint32_t blocks = (f->total_vec + 256u * FB_U - 1) / (256u * FB_U);
        if (blocks > 256u * 8u) blocks = 256u * 8u;
        launch_fn<T, A>(*f, blocks, s);
    } else {
        // what the launch reads, as template arguments (the entry point checked: recon implies yref and no x)
        const int xm = b->recon ? 2 : (b->x ? 1 : 0);
        if (b->yref) {
            if (xm == 2) launch_bn<T, A, 1, 2>(*b, n, s);
            else if (xm == 1) launch_bn<T, A, 1, 1>(*b, n, s);
            else launch_bn<T, A, 1, 0>(*b, n, s);
        } else {
            if (xm == 1) launch_bn<T, A, 0, 1>(*b, n, s);
            else launch_bn<T, A, 0, 0>(*b, n, s);
        }
    }
}

template <class T>
int dispatch_nhwc(const FnArgs<T>* f, const BnArgs<T>* b, int act, uint32_t n, hipStream_t s)
{
    switch (act) {
    case 1: launch_nhwc<T, 1>(f, b, n, s); return MG_OK;
    case 2: launch_nhwc<T, 2>(f, b, n, s); return MG_OK;
    case 3: launch_nhwc<T, 3>(f, b, n, s); return MG_OK;
    }
    mg_set_error("channels-last epilogue: only linear / relu / lrelu are fused (activation %d)", act);
    return MG_E_ARG;
}

} // namespace

extern "C" int mg_fma_bias_act(const void* x, const float* d, const float* noise, const void* b, void* y, int dtype,
                               int act, float alpha, float gain, float clamp, float noise_gain, const float* noise_gain_dev,
                               int64_t noise_stride_n, int32_t n, int32_t c, int32_t hw, void* stream)
{
    MG_REQUIRE(x && y, "x must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "fma_bias_act: unsupported dtype %d", dtype);
    MG_REQUIRE(n >= 0 && c >= 1 && hw >= 1, "fma_bias_act: bad shape");
    int64_t total = (int64_t)n * c * hw;
    MG_REQUIRE(total <= INT32_MAX, "x is too large");
    if (total == 0) return MG_OK;
    int es = mg_dtype_size(dtype), vec = 16 / es;
    bool wide = (hw % vec == 0) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_FMA_BIAS_ACT, s, 2.0 * total * es);
    int rc = MG_OK;
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        FbArgs<T> p;
        p.x = (const T*)x; p.d = d; p.noise = noise; p.b = (const T*)b; p.y = (T*)y; p.y2 = nullptr;
        p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.noise_gain = noise_gain; p.noise_gain_dev = noise_gain_dev; p.noise_stride_n = noise_stride_n;
        p.c = (uint32_t)c; p.hw = (uint32_t)hw; p.total_vec = (uint32_t)total; p.c_div = make_fastdiv((uint32_t)c);
        rc = dispatch_fb<T>(p, act, wide, s);
    });
    if (rc) return rc;
    return mg_check_launch(MG_K_FMA_BIAS_ACT, "fma_bias_act");
}

extern "C" int mg_filtered_lrelu(const void* x, const void* b, const float* fu, const float* fd, void* y, int dtype,
                                 int32_t n, int32_t c, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                                 int32_t fu_h, int32_t fu_w, int32_t fd_h, int32_t fd_w, int up, int down, int padx0,
                                 int pady0, float alpha, float gain, float clamp, int flip, void* stream)
{
    MG_REQUIRE(x && fu && fd && y, "x must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "filtered_lrelu: unsupported dtype %d", dtype);
    MG_REQUIRE((up == 1 || up == 2) && (down == 1 || down == 2), "filtered_lrelu: up and down must be 1 or 2");
    MG_REQUIRE(fu_h >= 1 && fu_w >= 1 && fd_h >= 1 && fd_w >= 1 && fu_h <= FL_MAXF && fu_w <= FL_MAXF &&
               fd_h <= FL_MAXF && fd_w <= FL_MAXF, "filtered_lrelu: filters must be 1..%d taps per axis", FL_MAXF);
    MG_REQUIRE(n >= 0 && c >= 1 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, "output must be at least 1x1");
    MG_REQUIRE((int64_t)n * c * in_h * in_w <= INT32_MAX && (int64_t)n * c * out_h * out_w <= INT32_MAX, "x is too large");
    if (n == 0) return MG_OK;
    FlArgs p;
    p.x = x; p.b = b; p.fu = fu; p.fd = fd; p.y = y;
    p.n = n; p.c = c; p.in_h = in_h; p.in_w = in_w; p.out_h = out_h; p.out_w = out_w;
    p.fu_h = fu_h; p.fu_w = fu_w; p.fd_h = fd_h; p.fd_w = fd_w; p.up = up; p.down = down;
    p.padx0 = padx0; p.pady0 = pady0; p.flip = flip ? 1 : 0; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.tiles_x = (out_w + FL_TW - 1) / FL_TW; p.tiles_y = (out_h + FL_TH - 1) / FL_TH;
    // LDS: worst-case footprint of one tile
    int mid_w = (FL_TW - 1) * down + fd_w, mid_h = (FL_TH - 1) * down + fd_h;
    int iw = (mid_w + fu_w - 1) / up + 2, ih = (mid_h + fu_h - 1) / up + 2;
    size_t lds = sizeof(float) * (2 * FL_MAXF * FL_MAXF + (size_t)iw * ih + (size_t)mid_w * mid_h);
    MG_REQUIRE(lds <= 160 * 1024, "filtered_lrelu: tile does not fit LDS");
    hipStream_t s = (hipStream_t)stream;
    int es = mg_dtype_size(dtype);
    MgLaunchScope scope(MG_K_FILTERED_LRELU, s, ((double)n * c * in_h * in_w + (double)n * c * out_h * out_w) * es);
    int64_t blocks = (int64_t)p.tiles_x * p.tiles_y * n * c;
    MG_REQUIRE(blocks <= INT32_MAX, "x is too large");
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)filtered_lrelu_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((filtered_lrelu_kernel<T>), dim3((uint32_t)blocks), dim3(256), lds, s, p);
    });
    return mg_check_launch(MG_K_FILTERED_LRELU, "filtered_lrelu");
}

extern "C" int mg_bwd_fused(const void* dy, const void* yref, const void* x, const float* d, const float* noise,
                            void* dx, float* gd, float* gb, float* gs, int dtype, int act, float alpha, float gain,
                            float clamp, int64_t noise_stride_n, int32_t n, int32_t c, int32_t hw, void* stream)
{
    MG_REQUIRE(dy && dx, "dy must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "bwd_fused: unsupported dtype %d", dtype);
    MG_REQUIRE(n >= 0 && c >= 1 && hw >= 1, "bwd_fused: bad shape");
    MG_REQUIRE(!gd || x, "bwd_fused: gd needs x");
    MG_REQUIRE(!gs || noise, "bwd_fused: gs needs noise");
    int64_t total = (int64_t)n * c * hw;
    MG_REQUIRE(total <= INT32_MAX, "x is too large");
    if (total == 0) return MG_OK;
    int es = mg_dtype_size(dtype), vec = 16 / es;
    bool wide = (hw % vec == 0) && ((((uintptr_t)dy | (uintptr_t)yref | (uintptr_t)x | (uintptr_t)dx) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    int passes = 2 + (yref != nullptr) + (x != nullptr);
    MgLaunchScope scope(MG_K_BWD_FUSED, s, (double)passes * total * es);
    int rc = MG_OK;
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        BrArgs<T> p;
        p.dy = (const T*)dy; p.yref = (const T*)yref; p.x = (const T*)x; p.d = d; p.noise = gs ? noise : nullptr; p.dx = (T*)dx;
        p.gd = gd; p.gb = gb; p.gs = gs; p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.noise_stride_n = noise_stride_n;
        p.c = (uint32_t)c; p.hw = (uint32_t)hw; p.slices = 1; p.slice_len = (uint32_t)hw;
        rc = dispatch_br<T>(p, act, wide, (uint32_t)(n * c), s);
    });
    if (rc) return rc;
    return mg_check_launch(MG_K_BWD_FUSED, "bwd_fused");
}

extern "C" int mg_fma_bias_act_nhwc(const void* x, const float* d, const float* noise, const void* b, void* y, int dtype,
                                    int act, float alpha, float gain, float clamp, float noise_gain, const float* noise_gain_dev,
                                    int64_t noise_stride_n, int32_t n, int32_t c, int32_t hw, void* stream)
{
    MG_REQUIRE(x && y, "x must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "fma_bias_act_nhwc: unsupported dtype %d", dtype);
    MG_REQUIRE(n >= 0 && c >= 1 && hw >= 1, "fma_bias_act_nhwc: bad shape");
    int64_t total = (int64_t)n * c * hw;
    MG_REQUIRE(total <= INT32_MAX, "x is too large");
    if (total == 0) return MG_OK;
    int es = mg_dtype_size(dtype), vec = 16 / es;
    MG_REQUIRE(c % vec == 0, "fma_bias_act_nhwc: channels must be a multiple of %d", vec);
    MG_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)d | (uintptr_t)b) & 15) == 0, "fma_bias_act_nhwc: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_FMA_BIAS_ACT, s, 2.0 * total * es);
    int rc = MG_OK;
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        FnArgs<T> p;
        p.x = (const T*)x; p.d = d; p.noise = noise; p.b = (const T*)b; p.y = (T*)y;
        p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.noise_gain = noise_gain; p.noise_gain_dev = noise_gain_dev; p.noise_stride_n = noise_stride_n;
        p.c = (uint32_t)c; p.hw = (uint32_t)hw; p.cv = (uint32_t)(c / vec); p.total_vec = (uint32_t)(total / vec);
        p.cv_div = make_fastdiv(p.cv); p.hw_div = make_fastdiv(p.hw);
        rc = dispatch_nhwc<T>(&p, nullptr, act, (uint32_t)n, s);
    });
    if (rc) return rc;
    return mg_check_launch(MG_K_FMA_BIAS_ACT, "fma_bias_act_nhwc");
}

extern "C" int mg_bwd_fused_nhwc(const void* dy, const void* yref, const void* x, const float* d, const float* noise,
                                 void* dx, float* gd, float* gb_partial, float* gs, int dtype, int act, float alpha, float gain,
                                 float clamp, int64_t noise_stride_n, int32_t n, int32_t c, int32_t hw, void* stream)
{
    MG_REQUIRE(dy && dx, "dy must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "bwd_fused_nhwc: unsupported dtype %d", dtype);
    MG_REQUIRE(n >= 0 && c >= 1 && hw >= 1, "bwd_fused_nhwc: bad shape");
    return mg_bwd_fused_nhwc_y(dy, yref, x, d, noise, nullptr, nullptr, 0, dx, gd, gb_partial, gs, dtype, act, alpha, gain, clamp,
                               noise_stride_n, n, c, hw, stream);
}

extern "C" int mg_bwd_fused_nhwc_y(const void* dy, const void* yref, const void* x, const float* d, const float* noise,
                                   const float* strength, const void* bias, int recon, void* dx, float* gd, float* gb_partial, float* gs,
                                   int dtype, int act, float alpha, float gain, float clamp, int64_t noise_stride_n, int32_t n, int32_t c,
                                   int32_t hw, void* stream)
{
    MG_REQUIRE(dy && dx, "dy must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "bwd_fused_nhwc: unsupported dtype %d", dtype);
    MG_REQUIRE(n >= 0 && c >= 1 && hw >= 1, "bwd_fused_nhwc: bad shape");
    MG_REQUIRE(!gd || x || recon, "bwd_fused_nhwc: gd needs x (or recon)");
    MG_REQUIRE(!recon || (yref && !x && gain != 0.f && (act == 1 || act == 3)), "bwd_fused_nhwc: recon needs the saved output, a non-zero gain and a linear / lrelu activation");
    MG_REQUIRE(!gs || noise, "bwd_fused_nhwc: gs needs noise");
    int64_t total = (int64_t)n * c * hw;
    MG_REQUIRE(total <= INT32_MAX, "x is too large");
    if (total == 0) return MG_OK;
    int es = mg_dtype_size(dtype), vec = 16 / es;
    MG_REQUIRE(c % vec == 0 && c / vec <= 256 && 256 % (c / vec) == 0, "bwd_fused_nhwc: channels/%d must divide 256", vec);
    MG_REQUIRE((((uintptr_t)dy | (uintptr_t)yref | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)d | (uintptr_t)bias) & 15) == 0, "bwd_fused_nhwc: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int passes = 2 + (yref != nullptr) + (x != nullptr);
    MgLaunchScope scope(MG_K_BWD_FUSED, s, (double)passes * total * es);
    int rc = MG_OK;
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        BnArgs<T> p;
        p.dy = (const T*)dy; p.yref = (const T*)yref; p.x = (const T*)x; p.d = d; p.noise = (gs || (recon && gd)) ? noise : nullptr; p.dx = (T*)dx;
        p.gd = gd; p.gb = gb_partial; p.gs = gs; p.alpha = alpha; p.gain = gain; p.clamp = clamp; p.noise_stride_n = noise_stride_n;
        p.bias = (const T*)bias; p.strength = strength; p.recon = (recon && gd) ? 1 : 0;
        p.c = (uint32_t)c; p.hw = (uint32_t)hw; p.cv = (uint32_t)(c / vec);
        const uint32_t ppi = 256u / p.cv, unit = ppi * 4u;            // pixels per round of 4 vectors in flight
        uint32_t per = unit * 4u;                                     // about 16 vectors per lane per slice
        uint32_t slices = (p.hw + per - 1) / per;
        if (slices > 64) slices = 64;
        if (slices < 1) slices = 1;
        per = ((p.hw + slices - 1) / slices + unit - 1) / unit * unit;
        slices = (p.hw + per - 1) / per;
        p.slices = slices; p.slice_pix = per;
        rc = dispatch_nhwc<T>(nullptr, &p, act, (uint32_t)n, s);
    });
    if (rc) return rc;
    return mg_check_launch(MG_K_BWD_FUSED, "bwd_fused_nhwc");
}
