// stats.hip -- running moments for torch_utils.training_stats.report(): [count, sum, sum of squares] of a tensor
// accumulated into a float64[3] counter in ONE launch (upstream: flatten, cast, three reductions, a stack, a cast and an
// add per reported value -- reference torch_utils/training_stats.py:58-101 -- on ~75 values per training step).
#include "mg_common.h"

namespace {

template <class T>
__global__ void __launch_bounds__(256) stats_accumulate_kernel(const T* __restrict__ x, int64_t n, double* __restrict__ counter)
{
    __shared__ float sm[2][4];
    // float32 partial sums per lane (the reference reduces in float32 before widening), float64 across workgroups
    float s = 0.f, q = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = (float)to_acc<T>(x[i]);
        s += v;
        q = fmaf(v, v, q);
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        q += __shfl_down(q, off, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sm[0][wave] = s; sm[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ds = (double)sm[0][0] + (double)sm[0][1] + (double)sm[0][2] + (double)sm[0][3];
        double dq = (double)sm[1][0] + (double)sm[1][1] + (double)sm[1][2] + (double)sm[1][3];
        if (blockIdx.x == 0) atomicAdd(&counter[0], (double)n);
        atomicAdd(&counter[1], ds);
        atomicAdd(&counter[2], dq);
    }
}

} // namespace

extern "C" int mg_stats_accumulate(const void* x, int dtype, int64_t n, double* counter, void* stream)
{
    MG_REQUIRE(x != nullptr && counter != nullptr, "mg_stats_accumulate: null pointer");
    MG_REQUIRE(n > 0, "mg_stats_accumulate: n must be positive");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "mg_stats_accumulate: unsupported dtype %d", dtype);
    hipStream_t s = (hipStream_t)stream;
    int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 512) blocks = 512;
    MgLaunchScope scope(MG_K_STATS, s, (double)n * mg_dtype_size(dtype));
    MG_DISPATCH_DTYPE(dtype, if (sizeof(T) <= 4) hipLaunchKernelGGL((stats_accumulate_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s,
                                                                      (const T*)x, n, counter));
    return mg_check_launch(MG_K_STATS, "mg_stats_accumulate");
}
