// mg_common.h -- shared device/host helpers of libmontage_hip.so (gfx950 only).
//
// Element types: float, double, _Float16, __bf16.  All arithmetic is done in
// `acc_t` = float (double for double), the InternalType rule of the reference
// kernels (torch_utils/ops/upfirdn2d.cu:15-18, bias_act.cu:15-18), with bf16
// added.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/montage_hip.h"

typedef _Float16 mg_half;
typedef __bf16 mg_bf16;

template <class T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };

template <class T> __device__ __forceinline__ typename Acc<T>::type to_acc(T v) { return (typename Acc<T>::type)v; }
template <> __device__ __forceinline__ float to_acc<mg_bf16>(mg_bf16 v)
{
    // bf16 -> f32 is a 16-bit shift; spelled out so no library call is emitted.
    unsigned short u = __builtin_bit_cast(unsigned short, v);
    return __builtin_bit_cast(float, (unsigned)u << 16);
}
template <class T> __device__ __forceinline__ T from_acc(typename Acc<T>::type v) { return (T)v; }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950.
template <> __device__ __forceinline__ mg_bf16 from_acc<mg_bf16>(float v) { return (mg_bf16)v; }

// 16-byte vector of T
template <class T> struct Vec16 { static constexpr int N = 16 / sizeof(T); T v[16 / sizeof(T)]; } __attribute__((aligned(16)));

// floor(a / b), b > 0 (upfirdn2d.cu:20-24)
__host__ __device__ __forceinline__ int mg_floor_div(int a, int b)
{
    int q = a / b;
    return q - ((a % b != 0) & ((a ^ b) < 0));
}

// n / d for 0 <= n < 2^31 by multiply-shift (host precomputes mul, shr).
struct FastDiv {
    uint32_t d, mul, shr;
};
static inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    uint64_t p = 31 + l;
    f.mul = (uint32_t)(((1ull << p) + d - 1) / d);
    f.shr = (uint32_t)(p - 32);
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f)
{
    return (f.d <= 1) ? n : (__umulhi(n, f.mul) >> f.shr);
}

// ---- host side runtime (mg_runtime.cpp) -----------------------------------
void mg_set_error(const char* fmt, ...);
int  mg_check_launch(int kernel_id, const char* what);

// RAII timing scope: when profiling is on, records an event pair around the
// launch on `stream` and accounts `bytes` of algorithmic traffic to kernel_id.
struct MgLaunchScope {
    int kid; hipStream_t stream; double bytes; void* rec;
    MgLaunchScope(int kernel_id, hipStream_t s, double algorithmic_bytes);
    ~MgLaunchScope();
};

#define MG_REQUIRE(cond, ...) do { if (!(cond)) { mg_set_error(__VA_ARGS__); return MG_E_ARG; } } while (0)

static inline int mg_dtype_size(int dtype)
{
    switch (dtype) { case MG_F32: return 4; case MG_F16: return 2; case MG_BF16: return 2; case MG_F64: return 8; default: return 0; }
}

#define MG_DISPATCH_DTYPE(dtype, ...)                                              \
    switch (dtype) {                                                               \
    case MG_F32:  { typedef float T;    __VA_ARGS__; break; }                      \
    case MG_F16:  { typedef mg_half T;  __VA_ARGS__; break; }                      \
    case MG_BF16: { typedef mg_bf16 T;  __VA_ARGS__; break; }                      \
    case MG_F64:  { typedef double T;   __VA_ARGS__; break; }                      \
    default: mg_set_error("unsupported dtype %d", (int)(dtype)); return MG_E_ARG;  \
    }
