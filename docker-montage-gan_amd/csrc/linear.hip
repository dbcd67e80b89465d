// linear.hip -- fully connected layers on a SMALL batch: y[M,O] = x[M,I] @ w[O,I]^T (+ b), M <= 64, float32.
//
// The mapping networks (8 layers), the per-layer style affines (21 per generator pass) and the discriminator heads
// are [32 x 512] x [512 x 512]-sized products: 17 MFLOP and 1 MB of weights each -- 1 us of work that the BLAS
// library's general tiles turn into 9-10 us kernels, ~3 k of them per training step.  Here one WAVE owns one output
// feature (forward, weight gradient) or four samples of one 512-wide slice of the input features (input gradient):
// lanes run along the contiguous input-feature axis (32-byte vector loads of both operands), the other operand is
// re-read from L1/L2, sums are finished with wave shuffles.  (reference training/networks.py:104-133, torch.addmm)
#include "mg_common.h"

namespace {

constexpr int LN_MAXM = 64;         // samples
constexpr int LN_K = 8;             // input features per lane per chunk -> 512 per wave and chunk
typedef float f4_t __attribute__((ext_vector_type(4)));

// loads LN_K consecutive floats starting at p[i0] (zero beyond `limit`); `vec` = 16-byte aligned rows, full chunk
__device__ __forceinline__ void ln_load(const float* __restrict__ p, int i0, int limit, bool vec, float (&v)[LN_K])
{
    if (vec && i0 + LN_K <= limit) {
        const f4_t a = *(const f4_t*)(p + i0), b = *(const f4_t*)(p + i0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        #pragma unroll
        for (int k = 0; k < LN_K; k++) v[k] = (i0 + k < limit) ? p[i0 + k] : 0.f;
    }
}

// forward: wave = one output feature o; samples in tiles of 8 (accumulators stay in registers), lanes along i
__global__ void __launch_bounds__(256) linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ y, int m_total, int ic, int oc, int vec)
{
    constexpr int MT = 8;
    const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= oc) return;                                            // wave-uniform
    const float bias = b ? b[o] : 0.f;
    const float* wrow = w + (size_t)o * ic;
    for (int m0 = 0; m0 < m_total; m0 += MT) {
        float acc[MT];
        #pragma unroll
        for (int q = 0; q < MT; q++) acc[q] = 0.f;
        for (int i0 = lane * LN_K; i0 < ic; i0 += 64 * LN_K) {
            float wv[LN_K];
            ln_load(wrow, i0, ic, vec, wv);
            #pragma unroll
            for (int q = 0; q < MT; q++) {
                if (m0 + q >= m_total) break;                       // wave-uniform
                float xv[LN_K];
                ln_load(x + (size_t)(m0 + q) * ic, i0, ic, vec, xv);
                #pragma unroll
                for (int k = 0; k < LN_K; k++) acc[q] = __builtin_fmaf(xv[k], wv[k], acc[q]);
            }
        }
        #pragma unroll
        for (int q = 0; q < MT; q++) {
            if (m0 + q >= m_total) break;
            float a = acc[q];
            #pragma unroll
            for (int s = 32; s > 0; s >>= 1) a += __shfl_xor(a, s, 64);
            if (lane == 0) y[(size_t)(m0 + q) * oc + o] = a + bias;
        }
    }
}

// input gradient: gx[m,i] = sum_o g[m,o] w[o,i].  wave = 4 samples x one 512-wide slice of i
__global__ void __launch_bounds__(256) linear_bwd_x_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ gx,
                                                          int m_total, int ic, int oc, int chunks, int vec)
{
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int mg = task / chunks, chunk = task - mg * chunks;
    const int m0 = mg * 4;
    if (m0 >= m_total) return;                                      // wave-uniform
    const int i0 = chunk * 64 * LN_K + lane * LN_K;
    float acc[4][LN_K];
    #pragma unroll
    for (int q = 0; q < 4; q++)
        #pragma unroll
        for (int k = 0; k < LN_K; k++) acc[q][k] = 0.f;
    for (int o = 0; o < oc; o++) {
        float wv[LN_K];
        ln_load(w + (size_t)o * ic, i0, ic, vec, wv);
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            const float gv = (m0 + q < m_total) ? g[(size_t)(m0 + q) * oc + o] : 0.f;
            #pragma unroll
            for (int k = 0; k < LN_K; k++) acc[q][k] = __builtin_fmaf(gv, wv[k], acc[q][k]);
        }
    }
    #pragma unroll
    for (int q = 0; q < 4; q++) {
        if (m0 + q >= m_total) break;
        #pragma unroll
        for (int k = 0; k < LN_K; k++)
            if (i0 + k < ic) gx[(size_t)(m0 + q) * ic + i0 + k] = acc[q][k];
    }
}

// weight (and bias) gradient: gw[o,i] = sum_m g[m,o] x[m,i];  gb[o] = sum_m g[m,o].  wave = (o, 512-wide slice of i)
__global__ void __launch_bounds__(256) linear_bwd_w_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ gw,
                                                          float* __restrict__ gb, int m_total, int ic, int oc, int chunks, int vec)
{
    const int lane = threadIdx.x & 63, task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int o = task / chunks, chunk = task - o * chunks;
    if (o >= oc) return;                                            // wave-uniform
    const int i0 = chunk * 64 * LN_K + lane * LN_K;
    float acc[LN_K];
    #pragma unroll
    for (int k = 0; k < LN_K; k++) acc[k] = 0.f;
    float sb = 0.f;
    for (int m = 0; m < m_total; m++) {
        const float gv = g[(size_t)m * oc + o];
        sb += gv;
        float xv[LN_K];
        ln_load(x + (size_t)m * ic, i0, ic, vec, xv);
        #pragma unroll
        for (int k = 0; k < LN_K; k++) acc[k] = __builtin_fmaf(gv, xv[k], acc[k]);
    }
    if (gw) {
        #pragma unroll
        for (int k = 0; k < LN_K; k++)
            if (i0 + k < ic) gw[(size_t)o * ic + i0 + k] = acc[k];
    }
    if (gb && chunk == 0 && lane == 0) gb[o] = sb;
}

} // namespace

extern "C" int mg_linear_fwd(const float* x, const float* w, const float* b, float* y, int32_t m, int32_t ic, int32_t oc, void* stream)
{
    MG_REQUIRE(x && w && y, "x must reside on the device");
    MG_REQUIRE(m >= 0 && m <= LN_MAXM && ic >= 1 && oc >= 1, "linear: at most %d samples", LN_MAXM);
    if (m == 0) return MG_OK;
    const int vec = (ic % 4 == 0) && ((((uintptr_t)x | (uintptr_t)w) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_LINEAR, s, 4.0 * ((double)m * ic + (double)oc * ic + (double)m * oc));
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((uint32_t)((oc + 3) / 4)), dim3(256), 0, s, x, w, b, y, m, ic, oc, vec);
    return mg_check_launch(MG_K_LINEAR, "linear_fwd");
}

extern "C" int mg_linear_bwd(const float* g, const float* x, const float* w, float* gx, float* gw, float* gb,
                             int32_t m, int32_t ic, int32_t oc, void* stream)
{
    MG_REQUIRE(g && x && w, "g must reside on the device");
    MG_REQUIRE(m >= 0 && m <= LN_MAXM && ic >= 1 && oc >= 1, "linear: at most %d samples", LN_MAXM);
    if (m == 0) return MG_OK;
    const int vec = (ic % 4 == 0) && ((((uintptr_t)x | (uintptr_t)w) & 15) == 0);
    const int chunks = (ic + 64 * LN_K - 1) / (64 * LN_K);
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_LINEAR, s, 4.0 * (2.0 * m * ic + 2.0 * oc * ic + 2.0 * m * oc));
    if (gx) {
        const int tasks = ((m + 3) / 4) * chunks;
        hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((uint32_t)((tasks + 3) / 4)), dim3(256), 0, s, g, w, gx, m, ic, oc, chunks, vec);
    }
    if (gw || gb) {
        const int tasks = oc * chunks;
        hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((uint32_t)((tasks + 3) / 4)), dim3(256), 0, s, g, x, gw, gb, m, ic, oc, chunks, vec);
    }
    return mg_check_launch(MG_K_LINEAR, "linear_bwd");
}
