// augment.hip -- the two image-sized ops of the ADA pipe that the reference leaves to generic library calls.
//
//  * mg_affine_warp_fwd / _bwd: the geometric transform's resampling step,
//        grid = F.affine_grid(theta, size, align_corners=False)
//        y    = grid_sample_gradfix.grid_sample(x, grid)          # bilinear, zeros, align_corners=False
//    (reference training/augment.py:376-377) without materialising the grid (8 bytes per output pixel) and with a
//    GATHER-form gradient: the adjoint of bilinear sampling under an affine map is evaluated per INPUT pixel by
//    walking the pre-image of its 2x2 footprint on the output lattice -- no atomics (aten scatters 4 float
//    atomics per output element: 3.2 ms per call at the bench shape, the largest single aten kernel of the step).
//  * mg_channel_mix: y[n,i,p] = sum_j A[n,i,j] * x[n,j,p] + t[n,i] for <= 4 channels -- the colour transform
//    `C[:, :3, :3] @ images + C[:, :3, 3:]` (reference training/augment.py:403-406), which as a batched GEMM with
//    M = 3 runs at ~1 % of the HBM rate.
//
// Both are linear in the image, so gradients of any order are the same two kernels with A transposed / fwd<->bwd.
#include "mg_common.h"

namespace {

// Pixel-space affine map of one sample: source (sx, sy) = (a*ox + b*oy + c, d*ox + e*oy + f) for OUTPUT pixel
// (ox, oy).  Derived from theta exactly as affine_grid + grid_sample(align_corners=False) compose:
//   xn = (2*ox + 1)/Wo - 1,  gx = t00*xn + t01*yn + t02,  sx = ((gx + 1)*Wi - 1)/2.
struct PixMap { float a, b, c, d, e, f; };

__device__ __forceinline__ PixMap make_pixmap(const float* th, int wi, int hi, int wo, int ho)
{
    const float t00 = th[0], t01 = th[1], t02 = th[2], t10 = th[3], t11 = th[4], t12 = th[5];
    const float kx = 0.5f * (float)wi, ky = 0.5f * (float)hi;
    const float ux = 2.f / (float)wo, uy = 2.f / (float)ho;           // xn = ux*ox + (1/wo - 1)
    const float x0 = 1.f / (float)wo - 1.f, y0 = 1.f / (float)ho - 1.f;
    PixMap m;
    m.a = kx * t00 * ux; m.b = kx * t01 * uy; m.c = kx * (t00 * x0 + t01 * y0 + t02 + 1.f) - 0.5f;
    m.d = ky * t10 * ux; m.e = ky * t11 * uy; m.f = ky * (t10 * x0 + t11 * y0 + t12 + 1.f) - 0.5f;
    return m;
}

template <class T, int CMAX>
__global__ void __launch_bounds__(256) affine_warp_fwd_kernel(const T* __restrict__ x, const float* __restrict__ theta,
                                                              T* __restrict__ y, int c, int hi, int wi, int ho, int wo,
                                                              uint32_t total)
{
    const uint32_t ohw = (uint32_t)ho * wo;
    const size_t ihw = (size_t)hi * wi;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const uint32_t n = idx / ohw, p = idx - n * ohw;
        const int oy = p / wo, ox = p - oy * wo;
        const PixMap m = make_pixmap(theta + 6 * n, wi, hi, wo, ho);
        const float sx = __builtin_fmaf(m.a, (float)ox, __builtin_fmaf(m.b, (float)oy, m.c));
        const float sy = __builtin_fmaf(m.d, (float)ox, __builtin_fmaf(m.e, (float)oy, m.f));
        const float fx = floorf(sx), fy = floorf(sy);
        // far outside: nothing to read (also keeps the int conversion defined)
        const bool any = sx > -1.f && sy > -1.f && sx < (float)wi && sy < (float)hi;
        const int x0 = any ? (int)fx : 0, y0 = any ? (int)fy : 0;
        const float wx1 = sx - fx, wx0 = 1.f - wx1, wy1 = sy - fy, wy0 = 1.f - wy1;
        const bool vx0 = any && x0 >= 0 && x0 < wi, vx1 = any && x0 + 1 >= 0 && x0 + 1 < wi;
        const bool vy0 = y0 >= 0 && y0 < hi, vy1 = y0 + 1 >= 0 && y0 + 1 < hi;
        const T* xp = x + (size_t)n * c * ihw;
        T* yp = y + (size_t)n * c * ohw + p;
        #pragma unroll
        for (int ch = 0; ch < CMAX; ch++) {
            if (ch >= c) break;
            const T* q = xp + ch * ihw;
            float v = 0.f;
            if (vy0 && vx0) v += to_acc<T>(q[(size_t)y0 * wi + x0]) * (wx0 * wy0);
            if (vy0 && vx1) v += to_acc<T>(q[(size_t)y0 * wi + x0 + 1]) * (wx1 * wy0);
            if (vy1 && vx0) v += to_acc<T>(q[(size_t)(y0 + 1) * wi + x0]) * (wx0 * wy1);
            if (vy1 && vx1) v += to_acc<T>(q[(size_t)(y0 + 1) * wi + x0 + 1]) * (wx1 * wy1);
            yp[(size_t)ch * ohw] = from_acc<T>(v);
        }
    }
}

// Adjoint as a gather.  Input pixel (ix, iy) receives w = (1-|sx-ix|)*(1-|sy-iy|) * dy[o] from every output
// pixel o whose source point lies strictly inside (ix-1, ix+1) x (iy-1, iy+1): exactly the corner weights the
// forward kernel hands out.  Those o lie in the pre-image of that square, a parallelogram centred on
// M^-1 (p - t) with half extents |M^-1| (1,1)^T; its bounding box on the output lattice is scanned.
template <class T, int CMAX>
__global__ void __launch_bounds__(256) affine_warp_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ theta,
                                                              T* __restrict__ dx, int c, int hi, int wi, int ho, int wo,
                                                              uint32_t total)
{
    const uint32_t ihw = (uint32_t)hi * wi;
    const size_t ohw = (size_t)ho * wo;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const uint32_t n = idx / ihw, p = idx - n * ihw;
        const int iy = p / wi, ix = p - iy * wi;
        const PixMap m = make_pixmap(theta + 6 * n, wi, hi, wo, ho);
        float acc[CMAX];
        #pragma unroll
        for (int ch = 0; ch < CMAX; ch++) acc[ch] = 0.f;
        const float det = m.a * m.e - m.b * m.d;
        int ox_lo = 0, ox_hi = wo - 1, oy_lo = 0, oy_hi = ho - 1;          // singular map: scan everything
        if (fabsf(det) > 1e-12f) {
            const float r = 1.f / det;
            const float i00 = m.e * r, i01 = -m.b * r, i10 = -m.d * r, i11 = m.a * r;
            const float px = (float)ix - m.c, py = (float)iy - m.f;
            const float cx = i00 * px + i01 * py, cy = i10 * px + i11 * py;
            // a little slack for the rounding of the inverse; candidates are re-tested with the forward map anyway
            const float hx = (fabsf(i00) + fabsf(i01)) * 1.0001f + 1e-3f, hy = (fabsf(i10) + fabsf(i11)) * 1.0001f + 1e-3f;
            const float lx = fmaxf(ceilf(cx - hx), 0.f), ux = fminf(floorf(cx + hx), (float)(wo - 1));
            const float ly = fmaxf(ceilf(cy - hy), 0.f), uy = fminf(floorf(cy + hy), (float)(ho - 1));
            // NaN / empty ranges fall out of the loops below
            ox_lo = (lx <= ux) ? (int)lx : 1; ox_hi = (lx <= ux) ? (int)ux : 0;
            oy_lo = (ly <= uy) ? (int)ly : 1; oy_hi = (ly <= uy) ? (int)uy : 0;
        }
        const T* gp = dy + (size_t)n * c * ohw;
        for (int oy = oy_lo; oy <= oy_hi; oy++) {
            for (int ox = ox_lo; ox <= ox_hi; ox++) {
                const float sx = __builtin_fmaf(m.a, (float)ox, __builtin_fmaf(m.b, (float)oy, m.c));
                const float sy = __builtin_fmaf(m.d, (float)ox, __builtin_fmaf(m.e, (float)oy, m.f));
                // the forward kernel's corner choice: x0 = floor(sx), weights (1 - frac, frac)
                const float fx = floorf(sx), fy = floorf(sy);
                float wx, wy;
                if ((float)ix == fx) wx = 1.f - (sx - fx); else if ((float)ix == fx + 1.f) wx = sx - fx; else continue;
                if ((float)iy == fy) wy = 1.f - (sy - fy); else if ((float)iy == fy + 1.f) wy = sy - fy; else continue;
                const float w = wx * wy;
                const T* q = gp + (size_t)oy * wo + ox;
                #pragma unroll
                for (int ch = 0; ch < CMAX; ch++) {
                    if (ch >= c) break;
                    acc[ch] = __builtin_fmaf(to_acc<T>(q[(size_t)ch * ohw]), w, acc[ch]);
                }
            }
        }
        T* op = dx + (size_t)n * c * ihw + p;
        #pragma unroll
        for (int ch = 0; ch < CMAX; ch++) {
            if (ch >= c) break;
            op[(size_t)ch * ihw] = from_acc<T>(acc[ch]);
        }
    }
}

// y[n,i,p] = sum_j A[n,i,j] x[n,j,p] + t[n,i];  one lane = VEC consecutive pixels of all channels of one sample
template <class T, int C, int VEC>
__global__ void __launch_bounds__(256) channel_mix_kernel(const T* __restrict__ x, const float* __restrict__ A,
                                                          const float* __restrict__ t, T* __restrict__ y, uint32_t hwv,
                                                          uint32_t total)
{
    struct alignas(sizeof(T) * VEC) V { T v[VEC]; };
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const uint32_t n = idx / hwv, pv = idx - n * hwv;
        float a[C][C], tt[C];
        #pragma unroll
        for (int i = 0; i < C; i++) {
            tt[i] = t ? t[n * C + i] : 0.f;
            #pragma unroll
            for (int j = 0; j < C; j++) a[i][j] = A[(n * C + i) * C + j];
        }
        V in[C];
        #pragma unroll
        for (int j = 0; j < C; j++) in[j] = ((const V*)x)[((size_t)n * C + j) * hwv + pv];
        #pragma unroll
        for (int i = 0; i < C; i++) {
            V out;
            #pragma unroll
            for (int e = 0; e < VEC; e++) {
                float v = tt[i];
                #pragma unroll
                for (int j = 0; j < C; j++) v = __builtin_fmaf(a[i][j], to_acc<T>(in[j].v[e]), v);
                out.v[e] = from_acc<T>(v);
            }
            ((V*)y)[((size_t)n * C + i) * hwv + pv] = out;
        }
    }
}

template <class T, int C>
void launch_mix(const void* x, const float* A, const float* t, void* y, int n, int64_t hw, bool wide, hipStream_t s)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    const uint32_t hwv = (uint32_t)(wide ? hw / VEC : hw);
    const uint32_t total = (uint32_t)n * hwv;
    uint32_t blocks = (total + 255u) / 256u;
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    if (wide) hipLaunchKernelGGL((channel_mix_kernel<T, C, VEC>), dim3(blocks), dim3(256), 0, s, (const T*)x, A, t, (T*)y, hwv, total);
    else      hipLaunchKernelGGL((channel_mix_kernel<T, C, 1>), dim3(blocks), dim3(256), 0, s, (const T*)x, A, t, (T*)y, hwv, total);
}

} // namespace

extern "C" int mg_affine_warp_fwd(const void* x, const float* theta, void* y, int dtype, int32_t n, int32_t c,
                                  int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w, void* stream)
{
    MG_REQUIRE(x && theta && y, "x must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "affine_warp: unsupported dtype %d", dtype);
    MG_REQUIRE(n >= 0 && c >= 1 && c <= 4 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, "affine_warp: bad shape (1..4 channels)");
    MG_REQUIRE((int64_t)n * c * in_h * in_w <= INT32_MAX && (int64_t)n * c * out_h * out_w <= INT32_MAX, "x is too large");
    if (n == 0) return MG_OK;
    const uint32_t total = (uint32_t)n * out_h * out_w;
    uint32_t blocks = (total + 255u) / 256u;
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipStream_t s = (hipStream_t)stream;
    int es = mg_dtype_size(dtype);
    MgLaunchScope scope(MG_K_AFFINE_WARP_FWD, s, ((double)n * c * in_h * in_w + (double)n * c * out_h * out_w) * es);
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        hipLaunchKernelGGL((affine_warp_fwd_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, (const T*)x, theta, (T*)y, c, in_h, in_w, out_h, out_w, total);
    });
    return mg_check_launch(MG_K_AFFINE_WARP_FWD, "affine_warp_fwd");
}

extern "C" int mg_affine_warp_bwd(const void* dy, const float* theta, void* dx, int dtype, int32_t n, int32_t c,
                                  int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w, void* stream)
{
    MG_REQUIRE(dy && theta && dx, "dy must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "affine_warp: unsupported dtype %d", dtype);
    MG_REQUIRE(n >= 0 && c >= 1 && c <= 4 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, "affine_warp: bad shape (1..4 channels)");
    MG_REQUIRE((int64_t)n * c * in_h * in_w <= INT32_MAX && (int64_t)n * c * out_h * out_w <= INT32_MAX, "x is too large");
    if (n == 0) return MG_OK;
    const uint32_t total = (uint32_t)n * in_h * in_w;
    uint32_t blocks = (total + 255u) / 256u;
    if (blocks > 256u * 32u) blocks = 256u * 32u;
    hipStream_t s = (hipStream_t)stream;
    int es = mg_dtype_size(dtype);
    MgLaunchScope scope(MG_K_AFFINE_WARP_BWD, s, ((double)n * c * in_h * in_w + (double)n * c * out_h * out_w) * es);
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        hipLaunchKernelGGL((affine_warp_bwd_kernel<T, 4>), dim3(blocks), dim3(256), 0, s, (const T*)dy, theta, (T*)dx, c, in_h, in_w, out_h, out_w, total);
    });
    return mg_check_launch(MG_K_AFFINE_WARP_BWD, "affine_warp_bwd");
}

extern "C" int mg_channel_mix(const void* x, const float* A, const float* t, void* y, int dtype, int32_t n, int32_t c,
                              int64_t hw, void* stream)
{
    MG_REQUIRE(x && A && y, "x must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "channel_mix: unsupported dtype %d", dtype);
    MG_REQUIRE(n >= 0 && c >= 1 && c <= 4 && hw >= 1, "channel_mix: bad shape (1..4 channels)");
    MG_REQUIRE((int64_t)n * c * hw <= INT32_MAX, "x is too large");
    if (n == 0) return MG_OK;
    int es = mg_dtype_size(dtype), vec = 16 / es;
    const bool wide = (hw % vec == 0) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_CHANNEL_MIX, s, 2.0 * n * c * hw * es);
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        switch (c) {
        case 1: launch_mix<T, 1>(x, A, t, y, n, hw, wide, s); break;
        case 2: launch_mix<T, 2>(x, A, t, y, n, hw, wide, s); break;
        case 3: launch_mix<T, 3>(x, A, t, y, n, hw, wide, s); break;
        default: launch_mix<T, 4>(x, A, t, y, n, hw, wide, s); break;
        }
    });
    return mg_check_launch(MG_K_CHANNEL_MIX, "channel_mix");
}
