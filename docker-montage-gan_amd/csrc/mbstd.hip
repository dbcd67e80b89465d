// mbstd.hip -- the minibatch standard-deviation layer of the discriminator epilogue (reference training/networks.py:667-691)
//
//   y = x.reshape(G, M, F, c, H, W);  y = y - y.mean(0);  y = y.square().mean(0);  y = (y + 1e-8).sqrt()
//   y = y.mean([2, 3, 4]).reshape(M, F, 1, 1).repeat(G, 1, H, W);  out = cat([x, y], dim=1)
//
// is ten aten launches forward and about fifteen backward on a 1 MB tensor ([32, 512, 4, 4] float32), ~100 times per training
// step.  Here: one launch each way.  Sample s = g * M + m belongs to group m; one workgroup owns one (m, f): it walks the
// c * H * W positions of its channel block, holds the G values of a position in registers (mean, variance, sigma on the fly),
// copies them into `out` (the concatenation) and block-reduces sigma.  The backward recomputes mean and sigma from x -- nothing
// is saved but x itself -- and needs dY[m, f] = sum over the G * H * W copies of the appended channel first.
#include "mg_common.h"

namespace {

constexpr int MB_THREADS = 1024;   // M * F workgroups is all the parallelism there is (8 at the step's shape): make them as wide as they go
constexpr int MB_MAXG = 16;      // group sizes with a compiled instance: 1, 2, 3, 4, 8, 16 (the G values of a position live in registers)

__device__ __forceinline__ float block_sum(float v, float* red)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();                                  // red may still be read from a previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    #pragma unroll
    for (int q = 0; q < MB_THREADS / 64; q++) t += red[q];
    return t;
}

// x [N, C, HW] -> out [N, C + F, HW];  N = G * M, C = F * c
template <int G>
__global__ void __launch_bounds__(MB_THREADS) mbstd_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int F, int c, int hw)
{
    __shared__ float red[MB_THREADS / 64];
    const int m = blockIdx.x / F, f = blockIdx.x - m * F;
    const int C = F * c, per = c * hw;                 // positions of this channel block; contiguous in memory per sample
    const size_t xs = (size_t)C * hw, os = (size_t)(C + F) * hw;
    float acc = 0.f;
    for (int p0 = threadIdx.x; p0 < per; p0 += MB_THREADS * 4) {
        float v[4][G];
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + u * MB_THREADS;
            #pragma unroll
            for (int g = 0; g < G; g++) v[u][g] = p < per ? x[(size_t)(g * M + m) * xs + (size_t)f * per + p] : 0.f;
        }
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + u * MB_THREADS;
            if (p >= per) continue;
            float mean = 0.f;
            #pragma unroll
            for (int g = 0; g < G; g++) mean += v[u][g];
            mean /= (float)G;
            float var = 0.f;
            #pragma unroll
            for (int g = 0; g < G; g++) { const float d = v[u][g] - mean; var = __builtin_fmaf(d, d, var); }
            acc += sqrtf(var / (float)G + 1e-8f);
            #pragma unroll
            for (int g = 0; g < G; g++) out[(size_t)(g * M + m) * os + (size_t)f * per + p] = v[u][g];      // the x part of the concatenation
        }
    }
    const float y = block_sum(acc, red) / (float)per;
    for (int e = threadIdx.x; e < G * hw; e += MB_THREADS) {
        const int g = e / hw, q = e - g * hw;
        out[(size_t)(g * M + m) * os + (size_t)(C + f) * hw + q] = y;
    }
}

// dout [N, C + F, HW], x [N, C, HW] -> dx [N, C, HW]:  dx = dout[:, :C] + dY[m, f] / (c HW) * (x - mean) / (G sigma)
template <int G>
__global__ void __launch_bounds__(MB_THREADS) mbstd_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, float* __restrict__ dx,
                                                        int M, int F, int c, int hw)
{
    __shared__ float red[MB_THREADS / 64];
    const int m = blockIdx.x / F, f = blockIdx.x - m * F;
    const int C = F * c, per = c * hw;
    const size_t xs = (size_t)C * hw, os = (size_t)(C + F) * hw;
    float part = 0.f;
    for (int e = threadIdx.x; e < G * hw; e += MB_THREADS) {
        const int g = e / hw, q = e - g * hw;
        part += dout[(size_t)(g * M + m) * os + (size_t)(C + f) * hw + q];
    }
    const float k = block_sum(part, red) / ((float)per * (float)G);
    for (int p0 = threadIdx.x; p0 < per; p0 += MB_THREADS * 4) {
        float v[4][G], d[4][G];
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + u * MB_THREADS;
            #pragma unroll
            for (int g = 0; g < G; g++) {
                v[u][g] = p < per ? x[(size_t)(g * M + m) * xs + (size_t)f * per + p] : 0.f;
                d[u][g] = p < per ? dout[(size_t)(g * M + m) * os + (size_t)f * per + p] : 0.f;
            }
        }
        #pragma unroll
        for (int u = 0; u < 4; u++) {
            const int p = p0 + u * MB_THREADS;
            if (p >= per) continue;
            float mean = 0.f;
            #pragma unroll
            for (int g = 0; g < G; g++) mean += v[u][g];
            mean /= (float)G;
            float var = 0.f;
            #pragma unroll
            for (int g = 0; g < G; g++) { const float t = v[u][g] - mean; var = __builtin_fmaf(t, t, var); }
            const float r = k / sqrtf(var / (float)G + 1e-8f);
            // d sigma / d x_g = (x_g - mean) / (G sigma): the mean's own dependence on x_g cancels (sum of deviations is 0)
            #pragma unroll
            for (int g = 0; g < G; g++) dx[(size_t)(g * M + m) * xs + (size_t)f * per + p] = __builtin_fmaf(v[u][g] - mean, r, d[u][g]);
        }
    }
}

} // namespace

extern "C" int mg_mbstd_fwd(const float* x, float* out, int32_t n, int32_t c_total, int32_t hw, int32_t group, int32_t f, void* stream)
{
    MG_REQUIRE(x && out, "mbstd: x must reside on the device");
    MG_REQUIRE(n >= 1 && c_total >= 1 && hw >= 1 && group >= 1 && group <= MB_MAXG && f >= 1, "mbstd: bad shape (group size <= %d)", MB_MAXG);
    MG_REQUIRE(n % group == 0 && c_total % f == 0, "mbstd: the batch must divide into groups and the channels into %d blocks", (int)f);
    MG_REQUIRE((int64_t)n * (c_total + f) * hw <= INT32_MAX, "x is too large");
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_MBSTD, s, 4.0 * ((double)n * c_total * hw * 2 + (double)n * f * hw));
    const dim3 grid((uint32_t)(n / group * f));
    switch (group) {
    case 1:  hipLaunchKernelGGL(mbstd_fwd_kernel<1>,  grid, dim3(MB_THREADS), 0, s, x, out, n / group, f, c_total / f, hw); break;
    case 2:  hipLaunchKernelGGL(mbstd_fwd_kernel<2>,  grid, dim3(MB_THREADS), 0, s, x, out, n / group, f, c_total / f, hw); break;
    case 3:  hipLaunchKernelGGL(mbstd_fwd_kernel<3>,  grid, dim3(MB_THREADS), 0, s, x, out, n / group, f, c_total / f, hw); break;
    case 4:  hipLaunchKernelGGL(mbstd_fwd_kernel<4>,  grid, dim3(MB_THREADS), 0, s, x, out, n / group, f, c_total / f, hw); break;
    case 8:  hipLaunchKernelGGL(mbstd_fwd_kernel<8>,  grid, dim3(MB_THREADS), 0, s, x, out, n / group, f, c_total / f, hw); break;
    case 16: hipLaunchKernelGGL(mbstd_fwd_kernel<16>, grid, dim3(MB_THREADS), 0, s, x, out, n / group, f, c_total / f, hw); break;
    default: mg_set_error("mbstd: group size %d has no compiled instance (1, 2, 3, 4, 8, 16)", (int)group); return MG_E_ARG;
    }
    return mg_check_launch(MG_K_MBSTD, "mbstd_fwd");
}

extern "C" int mg_mbstd_bwd(const float* dout, const float* x, float* dx, int32_t n, int32_t c_total, int32_t hw, int32_t group, int32_t f, void* stream)
{
    MG_REQUIRE(dout && x && dx, "mbstd: dout must reside on the device");
    MG_REQUIRE(n >= 1 && c_total >= 1 && hw >= 1 && group >= 1 && group <= MB_MAXG && f >= 1, "mbstd: bad shape (group size <= %d)", MB_MAXG);
    MG_REQUIRE(n % group == 0 && c_total % f == 0, "mbstd: the batch must divide into groups and the channels into %d blocks", (int)f);
    MG_REQUIRE((int64_t)n * (c_total + f) * hw <= INT32_MAX, "x is too large");
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_MBSTD, s, 4.0 * ((double)n * c_total * hw * 3 + (double)n * f * hw));
    const dim3 grid((uint32_t)(n / group * f));
    switch (group) {
    case 1:  hipLaunchKernelGGL(mbstd_bwd_kernel<1>,  grid, dim3(MB_THREADS), 0, s, dout, x, dx, n / group, f, c_total / f, hw); break;
    case 2:  hipLaunchKernelGGL(mbstd_bwd_kernel<2>,  grid, dim3(MB_THREADS), 0, s, dout, x, dx, n / group, f, c_total / f, hw); break;
    case 3:  hipLaunchKernelGGL(mbstd_bwd_kernel<3>,  grid, dim3(MB_THREADS), 0, s, dout, x, dx, n / group, f, c_total / f, hw); break;
    case 4:  hipLaunchKernelGGL(mbstd_bwd_kernel<4>,  grid, dim3(MB_THREADS), 0, s, dout, x, dx, n / group, f, c_total / f, hw); break;
    case 8:  hipLaunchKernelGGL(mbstd_bwd_kernel<8>,  grid, dim3(MB_THREADS), 0, s, dout, x, dx, n / group, f, c_total / f, hw); break;
    case 16: hipLaunchKernelGGL(mbstd_bwd_kernel<16>, grid, dim3(MB_THREADS), 0, s, dout, x, dx, n / group, f, c_total / f, hw); break;
    default: mg_set_error("mbstd: group size %d has no compiled instance (1, 2, 3, 4, 8, 16)", (int)group); return MG_E_ARG;
    }
    return mg_check_launch(MG_K_MBSTD, "mbstd_bwd");
}
