// demod.hip -- demodulation coefficients of the (non-fused-weights) modulated convolution and their gradient.
//
//   d[n,o] = rsqrt( sum_{i,kh,kw} (w[o,i,kh,kw] * s[n,i])^2 + 1e-8 )
//          = rsqrt( sum_i s[n,i]^2 * wsq[o,i] + 1e-8 ),   wsq[o,i] = sum_{kh,kw} w[o,i,kh,kw]^2
// (reference training/networks.py:76-77 on the per-sample weights; as the [N,I] x [I,O] product here).  As separate
// aten ops this is pow, sum, pow, mm, add, rsqrt forward and nine more kernels backward, 4-10 us each, for every
// SynthesisLayer call -- about 4 k launches per training step.  Three kernels here: forward (one workgroup per output
// channel: w[o] is squared and summed on the fly, wsq[o,:] is kept for the backward), gradient w.r.t. the weight
// (workgroup per output channel) and w.r.t. the styles (workgroup per sample).  All float32.
#include "mg_common.h"

namespace {

constexpr int DM_MAXN = 256;        // samples per launch (one value per lane slice in LDS)

constexpr int DM_MAXKK = 9;         // taps per (o, i) pair the coalesced paths stage (1x1 and 3x3 kernels); larger kernels read strided

// Weight layouts: CL = 0: [O][I][kk] (contiguous), CL = 1: [O][kk][I] (torch.channels_last of [O,I,kh,kw]).  Either way one
// output channel's weights are ONE contiguous run of I*kk floats, which the kernels below read / write with consecutive
// lanes on consecutive addresses.

// block = one output channel o.  Phase 1: wsq[o,i] = sum_k w^2 (coalesced), phase 2: 8 lanes per sample over the input channels.
template <int CL>
__global__ void __launch_bounds__(256) demod_fwd_kernel(const float* __restrict__ w, const float* __restrict__ s, float* __restrict__ d,
                                                       float* __restrict__ wsq, int n, int o_total, int ic, int kk)
{
    extern __shared__ float sm[];                   // [ic] squared sums, then (CL = 0) a staging tile of 256 * kk floats
    float* s_wsq = sm;
    const int o = blockIdx.x;
    const float* wp = w + (size_t)o * ic * kk;
    if (CL == 1) {
        // [kk][I]: lane t owns channels t, t+256, ...; for a fixed tap consecutive lanes read consecutive floats
        for (int i = threadIdx.x; i < ic; i += 256) {
            float a = 0.f;
            if (kk == 9) {          // the 3x3 layers: nine loads in flight instead of nine round trips (the run-time loop is not unrolled)
                float v[9];
                #pragma unroll
                for (int k = 0; k < 9; k++) v[k] = wp[(size_t)k * ic + i];
                #pragma unroll
                for (int k = 0; k < 9; k++) a = __builtin_fmaf(v[k], v[k], a);
            } else {
                for (int k = 0; k < kk; k++) { const float v = wp[(size_t)k * ic + i]; a = __builtin_fmaf(v, v, a); }
            }
            s_wsq[i] = a;
            wsq[(size_t)o * ic + i] = a;
        }
    } else if (kk <= DM_MAXKK) {
        // [I][kk]: stage 256 channels x kk taps (one contiguous run) into LDS with coalesced loads, then lane t sums its kk
        // neighbours (stride kk = 1 or 9 words: conflict-free)
        float* tile = sm + ic;
        for (int i0 = 0; i0 < ic; i0 += 256) {
            const int cnt = min(256, ic - i0) * kk;
            const float* src = wp + (size_t)i0 * kk;
            float stage[DM_MAXKK];          // cnt <= 256 * DM_MAXKK: every load of the run goes out before the first LDS write
            #pragma unroll
            for (int u = 0; u < DM_MAXKK; u++) { const int e = threadIdx.x + u * 256; stage[u] = e < cnt ? src[e] : 0.f; }
            #pragma unroll
            for (int u = 0; u < DM_MAXKK; u++) { const int e = threadIdx.x + u * 256; if (e < cnt) tile[e] = stage[u]; }
            __syncthreads();
            const int i = i0 + threadIdx.x;
            if (i < ic) {
                float a = 0.f;
                for (int k = 0; k < kk; k++) { const float v = tile[threadIdx.x * kk + k]; a = __builtin_fmaf(v, v, a); }
                s_wsq[i] = a;
                wsq[(size_t)o * ic + i] = a;
            }
            __syncthreads();
        }
    } else {
        for (int i = threadIdx.x; i < ic; i += 256) {
            float a = 0.f;
            for (int k = 0; k < kk; k++) { const float v = wp[(size_t)i * kk + k]; a = __builtin_fmaf(v, v, a); }
            s_wsq[i] = a;
            wsq[(size_t)o * ic + i] = a;
        }
    }
    __syncthreads();
    // each group of 8 lanes handles one sample: lane q of the group sums i = q, q+8, ...
    const int q = threadIdx.x & 7;
    for (int nn = threadIdx.x >> 3; nn < n; nn += 32) {
        const float* sp = s + (size_t)nn * ic;
        float a = 0.f;
        int i = q;
        for (; i + 56 < ic; i += 64) {          // 8 styles per lane in flight; same summation order as the plain loop
            float v[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) v[u] = sp[i + 8 * u];
            #pragma unroll
            for (int u = 0; u < 8; u++) a = __builtin_fmaf(v[u] * v[u], s_wsq[i + 8 * u], a);
        }
        for (; i < ic; i += 8) { const float v = sp[i]; a = __builtin_fmaf(v * v, s_wsq[i], a); }
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
        if (q == 0) d[(size_t)nn * o_total + o] = 1.f / sqrtf(a + 1e-8f);
    }
}

// dw[o,i,k] = 2 w[o,i,k] * sum_n t[n,o] s[n,i]^2,  t = -0.5 g d^3.   block = one output channel; dw has w's layout
template <int CL>
__global__ void __launch_bounds__(256) demod_bwd_w_kernel(const float* __restrict__ w, const float* __restrict__ s, const float* __restrict__ d,
                                                         const float* __restrict__ g, float* __restrict__ dw, int n, int o_total, int ic, int kk)
{
    extern __shared__ float sm[];                   // [DM_MAXN] t, then [ic] per-channel factors
    float* s_t = sm;
    float* s_a = sm + DM_MAXN;
    const int o = blockIdx.x;
    for (int nn = threadIdx.x; nn < n; nn += 256) {
        const float dv = d[(size_t)nn * o_total + o];
        s_t[nn] = -0.5f * g[(size_t)nn * o_total + o] * dv * dv * dv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ic; i += 256) {
        float a = 0.f;
        int nn = 0;
        for (; nn + 8 <= n; nn += 8) {          // 8 samples in flight, summed in the plain loop's order
            float v[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) v[u] = s[(size_t)(nn + u) * ic + i];
            #pragma unroll
            for (int u = 0; u < 8; u++) a = __builtin_fmaf(s_t[nn + u], v[u] * v[u], a);
        }
        for (; nn < n; nn++) { const float v = s[(size_t)nn * ic + i]; a = __builtin_fmaf(s_t[nn], v * v, a); }
        s_a[i] = a * 2.f;
    }
    __syncthreads();
    // one contiguous run of ic * kk floats, consecutive lanes on consecutive addresses; the channel of flat element e is
    // e / kk ([I][kk]) or e % ic ([kk][I])
    const size_t base = (size_t)o * ic * kk;
    const int total = ic * kk;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int i = CL ? (e % ic) : (e / kk);
        dw[base + e] = w[base + e] * s_a[i];
    }
}

// ds[n,i] = 2 s[n,i] * sum_o t[n,o] wsq[o,i].   block = one sample x 64 input channels; 4 lane groups split the sum over o
__global__ void __launch_bounds__(256) demod_bwd_s_kernel(const float* __restrict__ s, const float* __restrict__ d, const float* __restrict__ g,
                                                         const float* __restrict__ wsq, float* __restrict__ ds, int o_total, int ic, int tiles_i)
{
    extern __shared__ float s_to[];                 // [o_total] then [4][64] partial sums
    float* s_red = s_to + o_total;
    const int nn = blockIdx.x / tiles_i, i = (blockIdx.x % tiles_i) * 64 + (threadIdx.x & 63), os = threadIdx.x >> 6;
    for (int o = threadIdx.x; o < o_total; o += 256) {
        const float dv = d[(size_t)nn * o_total + o];
        s_to[o] = -0.5f * g[(size_t)nn * o_total + o] * dv * dv * dv;
    }
    __syncthreads();
    float a = 0.f;
    if (i < ic) {
        // 8 independent loads in flight per lane: the serial form of this loop (one dependent global load per step) was the
        // whole cost of the kernel (32 us per launch at 512 x 512)
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int o = os;
        for (; o + 28 < o_total; o += 32) {
            float v[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) v[u] = wsq[(size_t)(o + 4 * u) * ic + i];
            #pragma unroll
            for (int u = 0; u < 8; u++) part[u] = __builtin_fmaf(s_to[o + 4 * u], v[u], part[u]);
        }
        for (; o < o_total; o += 4) part[0] = __builtin_fmaf(s_to[o], wsq[(size_t)o * ic + i], part[0]);
        a = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    s_red[os * 64 + (threadIdx.x & 63)] = a;
    __syncthreads();
    if (os == 0 && i < ic) {
        const int il = threadIdx.x & 63;
        a = (s_red[il] + s_red[64 + il]) + (s_red[128 + il] + s_red[192 + il]);
        ds[(size_t)nn * ic + i] = 2.f * s[(size_t)nn * ic + i] * a;
    }
}

} // namespace

extern "C" int mg_demod_fwd(const float* w, const float* s, float* d, float* wsq, int32_t n, int32_t oc, int32_t ic, int32_t kk,
                            int32_t channels_last, void* stream)
{
    MG_REQUIRE(w && s && d && wsq, "w must reside on the device");
    MG_REQUIRE(n >= 0 && oc >= 1 && ic >= 1 && kk >= 1 && ic <= 8192, "demod: bad shape");
    if (n == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_DEMOD, st, 4.0 * ((double)oc * ic * kk + (double)n * ic + (double)n * oc));
    const size_t lds = sizeof(float) * ((size_t)ic + (channels_last ? 0 : 256 * DM_MAXKK));
    if (channels_last) hipLaunchKernelGGL(demod_fwd_kernel<1>, dim3((uint32_t)oc), dim3(256), lds, st, w, s, d, wsq, n, oc, ic, kk);
    else               hipLaunchKernelGGL(demod_fwd_kernel<0>, dim3((uint32_t)oc), dim3(256), lds, st, w, s, d, wsq, n, oc, ic, kk);
    return mg_check_launch(MG_K_DEMOD, "demod_fwd");
}

extern "C" int mg_demod_bwd(const float* w, const float* s, const float* d, const float* wsq, const float* g, float* ds, float* dw,
                            int32_t n, int32_t oc, int32_t ic, int32_t kk, int32_t channels_last, void* stream)
{
    MG_REQUIRE(w && s && d && wsq && g, "w must reside on the device");
    MG_REQUIRE(n >= 0 && n <= DM_MAXN && oc >= 1 && oc <= 16384 && ic >= 1 && ic <= 8192 && kk >= 1, "demod: bad shape (at most %d samples)", DM_MAXN);
    if (n == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_DEMOD, st, 4.0 * (2.0 * oc * ic * kk + 2.0 * n * ic + 2.0 * n * oc + (double)oc * ic));
    if (dw) {
        const size_t lds = sizeof(float) * ((size_t)DM_MAXN + ic);
        if (channels_last) hipLaunchKernelGGL(demod_bwd_w_kernel<1>, dim3((uint32_t)oc), dim3(256), lds, st, w, s, d, g, dw, n, oc, ic, kk);
        else               hipLaunchKernelGGL(demod_bwd_w_kernel<0>, dim3((uint32_t)oc), dim3(256), lds, st, w, s, d, g, dw, n, oc, ic, kk);
    }
    if (ds) {
        const int tiles_i = (ic + 63) / 64;
        hipLaunchKernelGGL(demod_bwd_s_kernel, dim3((uint32_t)(n * tiles_i)), dim3(256), sizeof(float) * ((size_t)oc + 256), st, s, d, g, wsq, ds, oc, ic, tiles_i);
    }
    return mg_check_launch(MG_K_DEMOD, "demod_bwd");
}
