// composite.hip -- differentiable "over" compositing of an RGBA layer stack.
// Replaces the python loops of alpha_composite_pytorch
// (reference custom_utils/image_utils.py:112-163, default branch):
//   canvas = layer 0; for l in 1..L-1: canvas = layer_l OVER canvas
//   a_o = a1 + a2*(1-a1);  c_o = nan_to_num((c1*a1 + c2*a2*(1-a1)) / a_o)
// One lane owns VEC consecutive pixels of one sample and walks the L layers in
// registers: (L+1)*4 planes of HBM traffic per pixel, nothing else.
// x: [B, L, 4, H, W] dense, y: [B, 4, H, W].
#include "mg_common.h"
#include <float.h>

namespace {

__device__ __forceinline__ float nan_to_num_f(float v)
{
    // NaN -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX  (image_utils.py:100-109 / torch.nan_to_num)
    if (v != v) return 0.f;
    return fminf(fmaxf(v, -FLT_MAX), FLT_MAX);
}

template <class T, int VEC> struct alignas(sizeof(T) * VEC) PV { T v[VEC]; };

template <class T, int VEC>
__global__ void __launch_bounds__(256) alpha_composite_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                  int l, uint32_t hw, uint32_t nvec_per_b, uint32_t total)
{
    typedef PV<T, VEC> V;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t b = idx / nvec_per_b, pv = idx - b * nvec_per_b;
        const T* xb = x + (size_t)b * l * 4 * hw + (size_t)pv * VEC;
        float c2[3][VEC], a2[VEC];
        {
            V r = *(const V*)(xb), g = *(const V*)(xb + hw), bl = *(const V*)(xb + 2 * (size_t)hw), a = *(const V*)(xb + 3 * (size_t)hw);
            #pragma unroll
            for (int e = 0; e < VEC; e++) { c2[0][e] = to_acc<T>(r.v[e]); c2[1][e] = to_acc<T>(g.v[e]); c2[2][e] = to_acc<T>(bl.v[e]); a2[e] = to_acc<T>(a.v[e]); }
        }
        for (int li = 1; li < l; li++) {
            const T* xl = xb + (size_t)li * 4 * hw;
            V c1v[3], a1v;
            c1v[0] = *(const V*)(xl); c1v[1] = *(const V*)(xl + hw); c1v[2] = *(const V*)(xl + 2 * (size_t)hw); a1v = *(const V*)(xl + 3 * (size_t)hw);
            #pragma unroll
            for (int e = 0; e < VEC; e++) {
                float a1 = to_acc<T>(a1v.v[e]);
                float t = a2[e] * (1.f - a1);
                float ao = a1 + t;
                #pragma unroll
                for (int k = 0; k < 3; k++)
                    c2[k][e] = nan_to_num_f((to_acc<T>(c1v[k].v[e]) * a1 + c2[k][e] * t) / ao);
                a2[e] = ao;
            }
        }
        T* yb = y + (size_t)b * 4 * hw + (size_t)pv * VEC;
        V o;
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            #pragma unroll
            for (int e = 0; e < VEC; e++) o.v[e] = from_acc<T>(c2[k][e]);
            *(V*)(yb + (size_t)k * hw) = o;
        }
        #pragma unroll
        for (int e = 0; e < VEC; e++) o.v[e] = from_acc<T>(a2[e]);
        *(V*)(yb + 3 * (size_t)hw) = o;
    }
}

// Backward: the autograd chain of the forward ops, literally (see oracle
// orc_alpha_composite_bwd): nan_to_num' = isfinite mask; div' as torch computes it.
// L is a template parameter so that the per-level canvases stay in registers.
template <class T, int L>
__global__ void __launch_bounds__(256) alpha_composite_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                  T* __restrict__ dx, uint32_t hw, uint32_t total, int safe_grad)
{
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t b = idx / hw, p = idx - b * hw;
        const T* xb = x + (size_t)b * L * 4 * hw + p;
        T* dxb = dx + (size_t)b * L * 4 * hw + p;
        const T* dyb = dy + (size_t)b * 4 * hw + p;
        float cs[L][3], as[L];
        cs[0][0] = to_acc<T>(xb[0]); cs[0][1] = to_acc<T>(xb[hw]); cs[0][2] = to_acc<T>(xb[2 * (size_t)hw]); as[0] = to_acc<T>(xb[3 * (size_t)hw]);
        #pragma unroll
        for (int li = 1; li < L; li++) {
            const T* xl = xb + (size_t)li * 4 * hw;
            float a1 = to_acc<T>(xl[3 * (size_t)hw]), t = as[li - 1] * (1.f - a1), ao = a1 + t;
            #pragma unroll
            for (int k = 0; k < 3; k++)
                cs[li][k] = nan_to_num_f((to_acc<T>(xl[(size_t)k * hw]) * a1 + cs[li - 1][k] * t) / ao);
            as[li] = ao;
        }
        float gc[3] = {to_acc<T>(dyb[0]), to_acc<T>(dyb[hw]), to_acc<T>(dyb[2 * (size_t)hw])}, ga = to_acc<T>(dyb[3 * (size_t)hw]);
        #pragma unroll
        for (int li = L - 1; li >= 1; li--) {
            const T* xl = xb + (size_t)li * 4 * hw;
            T* dxl = dxb + (size_t)li * 4 * hw;
            float a1 = to_acc<T>(xl[3 * (size_t)hw]), a2 = as[li - 1], t = a2 * (1.f - a1), ao = a1 + t;
            float g_ao = ga, g_a1 = 0.f, g_t = 0.f, g_c1[3], g_c2[3];
            #pragma unroll
            for (int k = 0; k < 3; k++) {
                float c1 = to_acc<T>(xl[(size_t)k * hw]), c2 = cs[li - 1][k];
                float num = c1 * a1 + c2 * t, q = num / ao;
                float fin = (fabsf(q) <= FLT_MAX) ? 1.f : 0.f;     // false for NaN and +/-inf
                float g_q = gc[k] * fin;
                float g_num = g_q / ao;
                g_ao += -g_q * ((num / ao) / ao);
                g_c1[k] = g_num * a1;
                g_a1 += g_num * c1;
                g_c2[k] = g_num * t;
                g_t += g_num * c2;
            }
            g_a1 += g_ao; g_t += g_ao;
            float g_a2 = g_t * (1.f - a1);
            g_a1 += -(g_t * a2);
            if (safe_grad && ao == 0.f) { g_c1[0] = g_c1[1] = g_c1[2] = 0.f; g_a1 = 0.f; g_a2 = 0.f; g_c2[0] = g_c2[1] = g_c2[2] = 0.f; }
            dxl[0] = from_acc<T>(g_c1[0]); dxl[hw] = from_acc<T>(g_c1[1]); dxl[2 * (size_t)hw] = from_acc<T>(g_c1[2]);
            dxl[3 * (size_t)hw] = from_acc<T>(g_a1);
            gc[0] = g_c2[0]; gc[1] = g_c2[1]; gc[2] = g_c2[2]; ga = g_a2;
        }
        dxb[0] = from_acc<T>(gc[0]); dxb[hw] = from_acc<T>(gc[1]); dxb[2 * (size_t)hw] = from_acc<T>(gc[2]); dxb[3 * (size_t)hw] = from_acc<T>(ga);
    }
}

template <class T, int L>
void launch_bwd(const T* x, const T* dy, T* dx, uint32_t hw, uint32_t total, int safe, hipStream_t s)
{
    uint32_t blocks = (total + 255u) / 256u;
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    hipLaunchKernelGGL((alpha_composite_bwd_kernel<T, L>), dim3(blocks), dim3(256), 0, s, x, dy, dx, hw, total, safe);
}

template <class T>
int dispatch_bwd(const T* x, const T* dy, T* dx, int l, uint32_t hw, uint32_t total, int safe, hipStream_t s)
{
    switch (l) {
#define C(L_) case L_: launch_bwd<T, L_>(x, dy, dx, hw, total, safe, s); return MG_OK;
    C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
#undef C
    }
    return MG_E_ARG;
}

} // namespace

extern "C" int mg_alpha_composite_fwd(const void* x, void* y, int dtype, int32_t b, int32_t l, int32_t h, int32_t w,
                                      void* stream)
{
    MG_REQUIRE(x && y, "x must reside on the device");
    MG_REQUIRE(b >= 0 && l >= 1 && l <= 16 && h >= 1 && w >= 1, "alpha_composite: need 1 <= L <= 16 layers of RGBA");
    MG_REQUIRE(dtype != MG_F64, "alpha_composite: float64 is not supported");
    int64_t hw = (int64_t)h * w;
    MG_REQUIRE((int64_t)b * l * 4 * hw <= INT32_MAX, "x is too large");
    if (b == 0) return MG_OK;
    int es = mg_dtype_size(dtype);
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_ALPHA_COMPOSITE_FWD, s, (double)b * (l + 1) * 4 * hw * es);
    int vec = 16 / es;
    bool wide = (hw % vec == 0) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        constexpr int VEC = 16 / sizeof(T);
        if (wide) {
            uint32_t nv = (uint32_t)(hw / VEC), total = nv * (uint32_t)b;
            uint32_t blocks = (total + 255u) / 256u; if (blocks > 4096u) blocks = 4096u;
            hipLaunchKernelGGL((alpha_composite_fwd_kernel<T, VEC>), dim3(blocks), dim3(256), 0, s, (const T*)x, (T*)y, l,
                               (uint32_t)hw, nv, total);
        } else {
            uint32_t total = (uint32_t)hw * (uint32_t)b;
            uint32_t blocks = (total + 255u) / 256u; if (blocks > 4096u) blocks = 4096u;
            hipLaunchKernelGGL((alpha_composite_fwd_kernel<T, 1>), dim3(blocks), dim3(256), 0, s, (const T*)x, (T*)y, l,
                               (uint32_t)hw, (uint32_t)hw, total);
        }
    });
    return mg_check_launch(MG_K_ALPHA_COMPOSITE_FWD, "alpha_composite_fwd");
}

extern "C" int mg_alpha_composite_bwd(const void* x, const void* dy, void* dx, int dtype, int32_t b, int32_t l,
                                      int32_t h, int32_t w, int safe_grad, void* stream)
{
    MG_REQUIRE(x && dy && dx, "x must reside on the device");
    MG_REQUIRE(b >= 0 && l >= 1 && l <= 16 && h >= 1 && w >= 1, "alpha_composite: need 1 <= L <= 16 layers of RGBA");
    MG_REQUIRE(dtype != MG_F64, "alpha_composite: float64 is not supported");
    int64_t hw = (int64_t)h * w;
    MG_REQUIRE((int64_t)b * l * 4 * hw <= INT32_MAX, "x is too large");
    if (b == 0) return MG_OK;
    int es = mg_dtype_size(dtype);
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_ALPHA_COMPOSITE_BWD, s, (double)b * (2 * l + 1) * 4 * hw * es);
    uint32_t total = (uint32_t)(hw * b);
    int rc = MG_OK;
    switch (dtype) {
    case MG_F32:  rc = dispatch_bwd<float>((const float*)x, (const float*)dy, (float*)dx, l, (uint32_t)hw, total, safe_grad, s); break;
    case MG_F16:  rc = dispatch_bwd<mg_half>((const mg_half*)x, (const mg_half*)dy, (mg_half*)dx, l, (uint32_t)hw, total, safe_grad, s); break;
    case MG_BF16: rc = dispatch_bwd<mg_bf16>((const mg_bf16*)x, (const mg_bf16*)dy, (mg_bf16*)dx, l, (uint32_t)hw, total, safe_grad, s); break;
    default: rc = MG_E_ARG;
    }
    if (rc != MG_OK) { mg_set_error("alpha_composite_bwd: unsupported configuration"); return rc; }
    return mg_check_launch(MG_K_ALPHA_COMPOSITE_BWD, "alpha_composite_bwd");
}
