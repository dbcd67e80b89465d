// translate.hip -- the STN's sampling step for a pure translation.
// Replaces, for theta = [[1,0,dx],[0,1,dy]]  (reference image_utils.py:316-335),
//   grid = F.affine_grid(theta, size, align_corners=False)
//   y    = F.grid_sample(x + shift, grid, bilinear, zeros, align_corners=False) - shift
// (reference fukuwarai/networks.py:246-255) without materialising the grid
// (4.2 MB/sample at 8x256x256) and with gather-only gradients:
//   fwd        y  <- x           4 neighbour loads per output
//   bwd_input  dx <- dy          exact transpose as a gather (no atomics)
//   bwd_shift  dt <- x, dy       d/d(dx,dy): block reduction + one atomic pair per block
// Coordinates follow aten's float32 op sequence (linspace base grid scaled by
// (S-1)/S, + translation, unnormalise ((g+1)*S-1)/2) so that floor() lands on the
// same texel as the reference even when the shift is a near-integer pixel count.
#include "mg_common.h"

namespace {

// Explicitly rounded (non-contracted) float ops: forward and both backward kernels
// must agree on floor(coordinate) bit for bit, and with the CPU oracle.
__device__ __forceinline__ float base_coord(int i, int s)
{
    if (s <= 1) return 0.f;
    float step = __fdiv_rn(2.f, (float)(s - 1));
    float v = (i < s / 2) ? __fadd_rn(-1.f, __fmul_rn(step, (float)i)) : __fsub_rn(1.f, __fmul_rn(step, (float)(s - i - 1)));
    return __fdiv_rn(__fmul_rn(v, (float)(s - 1)), (float)s);
}

// source coordinate (in pixels) sampled by output index o along an axis of size s
__device__ __forceinline__ float src_coord(int o, int s, float t)
{
    float g = __fadd_rn(base_coord(o, s), t);
    return __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(g, 1.f), (float)s), 1.f), 2.f);
}

template <class T>
__global__ void __launch_bounds__(256) translate_fwd_kernel(const T* __restrict__ x, const float* __restrict__ t,
                                                            T* __restrict__ y, int c, int h, int w, float shift,
                                                            uint32_t total)
{
    const uint32_t hw = (uint32_t)h * w;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t plane = idx / hw, p = idx - plane * hw;
        int oy = p / w, ox = p - oy * w;
        uint32_t m = plane / c;
        float ix = src_coord(ox, w, t[2 * m]), iy = src_coord(oy, h, t[2 * m + 1]);
        float fx = floorf(ix), fy = floorf(iy);
        int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
        const T* xp = x + (size_t)plane * hw;
        bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
        float v = 0.f;
        if (vy0 && vx0) v += (to_acc<T>(xp[y0 * w + x0]) + shift) * wx0 * wy0;
        if (vy0 && vx1) v += (to_acc<T>(xp[y0 * w + x1]) + shift) * wx1 * wy0;
        if (vy1 && vx0) v += (to_acc<T>(xp[y1 * w + x0]) + shift) * wx0 * wy1;
        if (vy1 && vx1) v += (to_acc<T>(xp[y1 * w + x1]) + shift) * wx1 * wy1;
        y[idx] = from_acc<T>(v - shift);
    }
}

// Weight with which output index o reads input index i along one axis (0 if it does not).
__device__ __forceinline__ float axis_weight(int o, int i, int s, float t)
{
    if (o < 0 || o >= s) return 0.f;
    float c = src_coord(o, s, t);
    float f = floorf(c);
    int i0 = (int)f;
    if (i == i0) return (f + 1.f) - c;
    if (i == i0 + 1) return c - f;
    return 0.f;
}

template <class T>
__global__ void __launch_bounds__(256) translate_bwd_input_kernel(const T* __restrict__ dy, const float* __restrict__ t,
                                                                  T* __restrict__ dx, int c, int h, int w, uint32_t total)
{
    const uint32_t hw = (uint32_t)h * w;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        uint32_t plane = idx / hw, p = idx - plane * hw;
        int iy = p / w, ix = p - iy * w;
        uint32_t m = plane / c;
        float tx = t[2 * m], ty = t[2 * m + 1];
        // outputs that can touch this texel: o ~ i - shift_in_pixels, +-1 for rounding
        int cx = ix - (int)floorf(tx * (float)w * 0.5f), cy = iy - (int)floorf(ty * (float)h * 0.5f);
        float wxs[4], wys[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) { wxs[k] = axis_weight(cx - 2 + k, ix, w, tx); wys[k] = axis_weight(cy - 2 + k, iy, h, ty); }
        const T* gp = dy + (size_t)plane * hw;
        float v = 0.f;
        #pragma unroll
        for (int b = 0; b < 4; b++) {
            if (wys[b] == 0.f) continue;
            #pragma unroll
            for (int a = 0; a < 4; a++)
                if (wxs[a] != 0.f) v += to_acc<T>(gp[(cy - 2 + b) * w + (cx - 2 + a)]) * wxs[a] * wys[b];
        }
        dx[idx] = from_acc<T>(v);
    }
}

// d/d(dx, dy): grid gradient summed over pixels and channels (d grid / d t = 1),
// times the unnormalise factor S/2 (aten grid_sampler_2d_backward).
template <class T>
__global__ void __launch_bounds__(256) translate_bwd_shift_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                  const float* __restrict__ t, float* __restrict__ dt,
                                                                  int c, int h, int w, float shift, uint32_t blocks_per_m)
{
    const uint32_t hw = (uint32_t)h * w;
    const uint32_t m = blockIdx.x / blocks_per_m, blk = blockIdx.x - m * blocks_per_m;
    const float tx = t[2 * m], ty = t[2 * m + 1];
    float gx = 0.f, gy = 0.f;
    for (uint32_t p = blk * 256u + threadIdx.x; p < hw; p += blocks_per_m * 256u) {
        int oy = p / w, ox = p - oy * w;
        float ix = src_coord(ox, w, tx), iy = src_coord(oy, h, ty);
        float fx = floorf(ix), fy = floorf(iy);
        int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
        bool vx0 = x0 >= 0 && x0 < w, vx1 = x1 >= 0 && x1 < w, vy0 = y0 >= 0 && y0 < h, vy1 = y1 >= 0 && y1 < h;
        for (int ci = 0; ci < c; ci++) {
            const T* xp = x + ((size_t)m * c + ci) * hw;
            float g = to_acc<T>(dy[((size_t)m * c + ci) * hw + p]);
            float v00 = (vy0 && vx0) ? to_acc<T>(xp[y0 * w + x0]) + shift : 0.f;
            float v01 = (vy0 && vx1) ? to_acc<T>(xp[y0 * w + x1]) + shift : 0.f;
            float v10 = (vy1 && vx0) ? to_acc<T>(xp[y1 * w + x0]) + shift : 0.f;
            float v11 = (vy1 && vx1) ? to_acc<T>(xp[y1 * w + x1]) + shift : 0.f;
            // out-of-image taps contribute neither value nor gradient (aten: masked safe_add / within_bounds)
            float dxv = 0.f, dyv = 0.f;
            if (vy0 && vx0) { dxv -= v00 * wy0; dyv -= v00 * wx0; }
            if (vy0 && vx1) { dxv += v01 * wy0; dyv -= v01 * wx1; }
            if (vy1 && vx0) { dxv -= v10 * wy1; dyv += v10 * wx0; }
            if (vy1 && vx1) { dxv += v11 * wy1; dyv += v11 * wx1; }
            gx += dxv * g; gy += dyv * g;
        }
    }
    gx *= 0.5f * (float)w; gy *= 0.5f * (float)h;
    // wave reduction (64 lanes), then across the 4 waves through LDS
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) { gx += __shfl_down(gx, o, 64); gy += __shfl_down(gy, o, 64); }
    __shared__ float sg[2][4];
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sg[0][wv] = gx; sg[1][wv] = gy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&dt[2 * m], sg[0][0] + sg[0][1] + sg[0][2] + sg[0][3]);
        atomicAdd(&dt[2 * m + 1], sg[1][0] + sg[1][1] + sg[1][2] + sg[1][3]);
    }
}

int check_common(const void* a, const void* b, int dtype, int32_t m, int32_t c, int32_t h, int32_t w)
{
    MG_REQUIRE(a && b, "x must reside on the device");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F16 || dtype == MG_BF16, "translate_sample: unsupported dtype %d", dtype);
    MG_REQUIRE(m >= 0 && c >= 1 && h >= 1 && w >= 1, "translate_sample: bad shape");
    MG_REQUIRE((int64_t)m * c * h * w <= INT32_MAX, "x is too large");
    return MG_OK;
}

} // namespace

extern "C" int mg_translate_sample_fwd(const void* x, const float* t, void* y, int dtype, int32_t m, int32_t c,
                                       int32_t h, int32_t w, float shift, void* stream)
{
    int rc = check_common(x, y, dtype, m, c, h, w);
    if (rc) return rc;
    MG_REQUIRE(t, "t must reside on the device");
    if (m == 0) return MG_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t total = (uint32_t)((int64_t)m * c * h * w);
    MgLaunchScope scope(MG_K_TRANSLATE_FWD, s, 2.0 * total * mg_dtype_size(dtype));
    uint32_t blocks = (total + 255u) / 256u; if (blocks > 8192u) blocks = 8192u;
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        hipLaunchKernelGGL((translate_fwd_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)x, t, (T*)y, c, h, w, shift, total);
    });
    return mg_check_launch(MG_K_TRANSLATE_FWD, "translate_sample_fwd");
}

extern "C" int mg_translate_sample_bwd_input(const void* dy, const float* t, void* dx, int dtype, int32_t m, int32_t c,
                                             int32_t h, int32_t w, void* stream)
{
    int rc = check_common(dy, dx, dtype, m, c, h, w);
    if (rc) return rc;
    MG_REQUIRE(t, "t must reside on the device");
    if (m == 0) return MG_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t total = (uint32_t)((int64_t)m * c * h * w);
    MgLaunchScope scope(MG_K_TRANSLATE_BWD_INPUT, s, 2.0 * total * mg_dtype_size(dtype));
    uint32_t blocks = (total + 255u) / 256u; if (blocks > 8192u) blocks = 8192u;
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        hipLaunchKernelGGL((translate_bwd_input_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)dy, t, (T*)dx, c, h, w, total);
    });
    return mg_check_launch(MG_K_TRANSLATE_BWD_INPUT, "translate_sample_bwd_input");
}

extern "C" int mg_translate_sample_bwd_shift(const void* x, const void* dy, const float* t, float* dt, int dtype,
                                             int32_t m, int32_t c, int32_t h, int32_t w, float shift, void* stream)
{
    int rc = check_common(x, dy, dtype, m, c, h, w);
    if (rc) return rc;
    MG_REQUIRE(t && dt, "t must reside on the device");
    if (m == 0) return MG_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t hw = (uint32_t)h * (uint32_t)w;
    uint32_t bpm = (hw + 1023u) / 1024u; if (bpm < 1) bpm = 1; if (bpm > 64u) bpm = 64u;
    MgLaunchScope scope(MG_K_TRANSLATE_BWD_SHIFT, s, 2.0 * (double)m * c * hw * mg_dtype_size(dtype));
    MG_DISPATCH_DTYPE(dtype, {
        if (sizeof(T) == 8) break;
        hipLaunchKernelGGL((translate_bwd_shift_kernel<T>), dim3(bpm * (uint32_t)m), dim3(256), 0, s, (const T*)x, (const T*)dy,
                           t, dt, c, h, w, shift, bpm);
    });
    return mg_check_launch(MG_K_TRANSLATE_BWD_SHIFT, "translate_sample_bwd_shift");
}
