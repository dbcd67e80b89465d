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
    // One wave = an 8x8 patch of output pixels, one workgroup = 16x16: under a rotation the 64 source points of a patch
    // stay within ~12 rows x 12 columns of the input (a row-major run of 64 output pixels would cut across up to 64 input rows).
    const uint32_t tiles_x = ((uint32_t)wo + 15u) >> 4, tiles_y = ((uint32_t)ho + 15u) >> 4, tiles = tiles_x * tiles_y;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t lx = ((wv & 1u) << 3) + (lane & 7u), ly = ((wv >> 1) << 3) + (lane >> 3);
    // workgroup ids go round-robin over the 8 XCDs (one L2 each): hand every XCD one contiguous eighth of the tile list, so that
    // neighbouring tiles -- whose source footprints overlap -- share an L2
    const uint32_t chunk = (total + 7u) >> 3;
    for (uint32_t t0 = blockIdx.x; t0 < chunk * 8u; t0 += gridDim.x) {
        const uint32_t t = (t0 & 7u) * chunk + (t0 >> 3);
        if (t >= total) continue;
        const uint32_t n = t / tiles, tt = t - n * tiles;
        const uint32_t ty = tt / tiles_x, tx = tt - ty * tiles_x;
        const int oy = (int)(ty * 16u + ly), ox = (int)(tx * 16u + lx);
        if (oy >= ho || ox >= wo) continue;
        const uint32_t p = (uint32_t)oy * wo + ox;
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
        // all 4 x C corner loads go out together from clamped (always valid) addresses; a corner outside the image is dropped by
        // a select AFTER the load, not a branch around it (zeros padding: what lies at the clamped address must not leak in)
        const int xa = min(max(x0, 0), wi - 1), xb = min(max(x0 + 1, 0), wi - 1);
        const size_t ra = (size_t)min(max(y0, 0), hi - 1) * wi, rb = (size_t)min(max(y0 + 1, 0), hi - 1) * wi;
        float q00[CMAX], q01[CMAX], q10[CMAX], q11[CMAX];
        #pragma unroll
        for (int ch = 0; ch < CMAX; ch++) {
            if (ch >= c) break;
            const T* q = xp + ch * ihw;
            q00[ch] = to_acc<T>(q[ra + xa]); q01[ch] = to_acc<T>(q[ra + xb]);
            q10[ch] = to_acc<T>(q[rb + xa]); q11[ch] = to_acc<T>(q[rb + xb]);
        }
        #pragma unroll
        for (int ch = 0; ch < CMAX; ch++) {
            if (ch >= c) break;
            float v = 0.f;
            v = (vy0 && vx0) ? v + q00[ch] * (wx0 * wy0) : v;
            v = (vy0 && vx1) ? v + q01[ch] * (wx1 * wy0) : v;
            v = (vy1 && vx0) ? v + q10[ch] * (wx0 * wy1) : v;
            v = (vy1 && vx1) ? v + q11[ch] * (wx1 * wy1) : v;
            yp[(size_t)ch * ohw] = from_acc<T>(v);
        }
    }
}

// Adjoint as a gather.  Input pixel (ix, iy) receives w = (1-|sx-ix|)*(1-|sy-iy|) * dy[o] from every output
// pixel o whose source point lies strictly inside (ix-1, ix+1) x (iy-1, iy+1): exactly the corner weights the
// forward kernel hands out.  Those o lie in the pre-image of that square, a parallelogram centred on
// M^-1 (p - t) with half extents |M^-1| (1,1)^T; its bounding box on the output lattice is scanned.
// Measured alternatives at the step's shapes (round 2, 9 calls per step, 366 us per call as below): one workgroup per 32x32 input tile
// collecting corner weights with LDS float adds -- 870 us (ds_add_f32 is the bottleneck); this gather with uniform trip counts and
// unconditional loads from clamped addresses -- 570 us (4x the load instructions: bound by the texture path).  Both were dropped.
template <class T, int CMAX>
__global__ void __launch_bounds__(256) affine_warp_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ theta,
                                                              T* __restrict__ dx, int c, int hi, int wi, int ho, int wo,
                                                              uint32_t total)
{
    const uint32_t ihw = (uint32_t)hi * wi;
    const size_t ohw = (size_t)ho * wo;
    // same 8x8-per-wave patches, here over the INPUT lattice (the pre-images of a patch's pixels overlap on the output lattice)
    const uint32_t tiles_x = ((uint32_t)wi + 15u) >> 4, tiles_y = ((uint32_t)hi + 15u) >> 4, tiles = tiles_x * tiles_y;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t lx = ((wv & 1u) << 3) + (lane & 7u), ly = ((wv >> 1) << 3) + (lane >> 3);
    const uint32_t chunk = (total + 7u) >> 3;          // XCD-contiguous tile order, as in the forward kernel
    for (uint32_t t0 = blockIdx.x; t0 < chunk * 8u; t0 += gridDim.x) {
        const uint32_t t = (t0 & 7u) * chunk + (t0 >> 3);
        if (t >= total) continue;
        const uint32_t n = t / tiles, tt = t - n * tiles;
        const uint32_t ty = tt / tiles_x, tx = tt - ty * tiles_x;
        const int iy = (int)(ty * 16u + ly), ix = (int)(tx * 16u + lx);
        if (iy >= hi || ix >= wi) continue;
        const uint32_t p = (uint32_t)iy * wi + ix;
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


// ---------------------------------------------------------------------------------------------------------------
// ADA transform parameters: the per-sample 3x3 geometric matrix (inverse, accumulated right to left), the reflect-padding
// margin it needs, the sampling matrix theta and the 4x4 colour matrix, from two tables of pre-drawn random numbers.
// One lane per sample; ~200 aten launches of 3-4 us per AugmentPipe call upstream (training/augment.py:236-330,380-420 build
// every factor with stack / matmul on [N,3,3] tensors) become one.
struct AdaCfg {
    float xflip, rotate90, xint, xint_max, scale, scale_std, rotate, rotate_max, aniso, aniso_std, xfrac, xfrac_std;
    float brightness, brightness_std, contrast, contrast_std, lumaflip, hue, hue_max, saturation, saturation_std;
    float width, height, hz_pad;
    int channels, has_lr;
    float left[9], right[9];
};

struct M3 { float m[3][3]; };
struct M4 { float m[4][4]; };

__device__ __forceinline__ M3 mul3(const M3& a, const M3& b)
{
    M3 r;
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            float v = 0.f;
            #pragma unroll
            for (int k = 0; k < 3; k++) v += a.m[i][k] * b.m[k][j];       // the summation order of a plain matmul
            r.m[i][j] = v;
        }
    return r;
}
__device__ __forceinline__ M4 mul4(const M4& a, const M4& b)
{
    M4 r;
    #pragma unroll
    for (int i = 0; i < 4; i++)
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            float v = 0.f;
            #pragma unroll
            for (int k = 0; k < 4; k++) v += a.m[i][k] * b.m[k][j];
            r.m[i][j] = v;
        }
    return r;
}
__device__ __forceinline__ M3 scale3(float sx, float sy) { M3 r = {{{sx, 0, 0}, {0, sy, 0}, {0, 0, 1}}}; return r; }
__device__ __forceinline__ M3 trans3(float tx, float ty) { M3 r = {{{1, 0, tx}, {0, 1, ty}, {0, 0, 1}}}; return r; }
__device__ __forceinline__ M3 rot3(float th) { M3 r = {{{cosf(th), sinf(-th), 0}, {sinf(th), cosf(th), 0}, {0, 0, 1}}}; return r; }

__global__ void __launch_bounds__(256) ada_params_kernel(const float* __restrict__ U, const float* __restrict__ Z, const float* __restrict__ p_ptr,
                                                         int n, int nu, int nz, AdaCfg cfg, float* __restrict__ Ginv, float* __restrict__ margin,
                                                         float* __restrict__ theta, float* __restrict__ Cmat, float* __restrict__ A4, float* __restrict__ t4)
{
    __shared__ float red[4][4];
    const float p = p_ptr[0];
    const float PI = 3.14159265358979323846f;
    float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float* u = U + (size_t)i * nu;
        const float* z = Z + (size_t)i * nz;
        int iu = 0, iz = 0;
        M3 G = scale3(1.f, 1.f);
        if (cfg.xflip > 0) {
            float v = floorf(u[iu++] * 2.f);
            v = (u[iu++] < cfg.xflip * p) ? v : 0.f;
            G = mul3(G, scale3(1.f / (1.f - 2.f * v), 1.f));
        }
        if (cfg.rotate90 > 0) {
            float v = floorf(u[iu++] * 4.f);
            v = (u[iu++] < cfg.rotate90 * p) ? v : 0.f;
            G = mul3(G, rot3(-(-(PI / 2.f) * v)));
        }
        if (cfg.xint > 0) {
            float t0 = (u[iu] * 2.f - 1.f) * cfg.xint_max, t1 = (u[iu + 1] * 2.f - 1.f) * cfg.xint_max;
            iu += 2;
            const bool on = u[iu++] < cfg.xint * p;
            t0 = on ? t0 : 0.f; t1 = on ? t1 : 0.f;
            G = mul3(G, trans3(-rintf(t0 * cfg.width), -rintf(t1 * cfg.height)));
        }
        if (cfg.scale > 0) {
            float sc = exp2f(z[iz++] * cfg.scale_std);
            sc = (u[iu++] < cfg.scale * p) ? sc : 1.f;
            G = mul3(G, scale3(1.f / sc, 1.f / sc));
        }
        const float p_rot = 1.f - sqrtf(fminf(fmaxf(1.f - cfg.rotate * p, 0.f), 1.f));
        if (cfg.rotate > 0) {
            float th = (u[iu++] * 2.f - 1.f) * PI * cfg.rotate_max;
            th = (u[iu++] < p_rot) ? th : 0.f;
            G = mul3(G, rot3(-(-th)));
        }
        if (cfg.aniso > 0) {
            float sc = exp2f(z[iz++] * cfg.aniso_std);
            sc = (u[iu++] < cfg.aniso * p) ? sc : 1.f;
            G = mul3(G, scale3(1.f / sc, 1.f / (1.f / sc)));
        }
        if (cfg.rotate > 0) {
            float th = (u[iu++] * 2.f - 1.f) * PI * cfg.rotate_max;
            th = (u[iu++] < p_rot) ? th : 0.f;
            G = mul3(G, rot3(-(-th)));
        }
        if (cfg.xfrac > 0) {
            float t0 = z[iz] * cfg.xfrac_std, t1 = z[iz + 1] * cfg.xfrac_std;
            iz += 2;
            const bool on = u[iu++] < cfg.xfrac * p;
            t0 = on ? t0 : 0.f; t1 = on ? t1 : 0.f;
            G = mul3(G, trans3(-(t0 * cfg.width), -(t1 * cfg.height)));
        }
        if (Ginv) {
            #pragma unroll
            for (int a = 0; a < 3; a++)
                #pragma unroll
                for (int b = 0; b < 3; b++) Ginv[(size_t)i * 9 + a * 3 + b] = G.m[a][b];
            // transformed image corners -> how far the reflect padding must reach (per axis and side)
            const float cx = (cfg.width - 1.f) / 2.f, cy = (cfg.height - 1.f) / 2.f;
            const float xs[4] = {-cx, cx, cx, -cx}, ys[4] = {-cy, -cy, cy, cy};
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                float X = G.m[0][0] * xs[k] + G.m[0][1] * ys[k] + G.m[0][2];
                float Y = G.m[1][0] * xs[k] + G.m[1][1] * ys[k] + G.m[1][2];
                mx[0] = fmaxf(mx[0], -X); mx[1] = fmaxf(mx[1], -Y); mx[2] = fmaxf(mx[2], X); mx[3] = fmaxf(mx[3], Y);
            }
            if (theta && cfg.has_lr) {
                M3 L, R;
                #pragma unroll
                for (int a = 0; a < 9; a++) { L.m[a / 3][a % 3] = cfg.left[a]; R.m[a / 3][a % 3] = cfg.right[a]; }
                M3 T = mul3(mul3(L, G), R);
                #pragma unroll
                for (int a = 0; a < 2; a++)
                    #pragma unroll
                    for (int b = 0; b < 3; b++) theta[(size_t)i * 6 + a * 3 + b] = T.m[a][b];
            }
        }
        if (Cmat) {
            M4 C = {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
            const float vv_ = 0.57735026918962576451f;      // 1 / sqrt(3): the luma axis (1,1,1,0)/sqrt(3)
            const float v[4] = {vv_, vv_, vv_, 0.f};
            if (cfg.brightness > 0) {
                float b = z[iz++] * cfg.brightness_std;
                b = (u[iu++] < cfg.brightness * p) ? b : 0.f;
                M4 T = {{{1, 0, 0, b}, {0, 1, 0, b}, {0, 0, 1, b}, {0, 0, 0, 1}}};
                C = mul4(T, C);
            }
            if (cfg.contrast > 0) {
                float c = exp2f(z[iz++] * cfg.contrast_std);
                c = (u[iu++] < cfg.contrast * p) ? c : 1.f;
                M4 S = {{{c, 0, 0, 0}, {0, c, 0, 0}, {0, 0, c, 0}, {0, 0, 0, 1}}};
                C = mul4(S, C);
            }
            if (cfg.lumaflip > 0) {
                float f = floorf(u[iu++] * 2.f);
                f = (u[iu++] < cfg.lumaflip * p) ? f : 0.f;
                M4 H;
                #pragma unroll
                for (int a = 0; a < 4; a++)
                    #pragma unroll
                    for (int b = 0; b < 4; b++) H.m[a][b] = (a == b ? 1.f : 0.f) - 2.f * (v[a] * v[b]) * f;       // Householder reflection
                C = mul4(H, C);
            }
            if (cfg.hue > 0 && cfg.channels > 1) {
                float th = (u[iu++] * 2.f - 1.f) * PI * cfg.hue_max;
                th = (u[iu++] < cfg.hue * p) ? th : 0.f;
                const float s = sinf(th), c = cosf(th), cc = 1.f - c;
                const float vx = v[0], vy = v[1], vz = v[2];
                M4 R = {{{vx * vx * cc + c, vx * vy * cc - vz * s, vx * vz * cc + vy * s, 0},
                         {vy * vx * cc + vz * s, vy * vy * cc + c, vy * vz * cc - vx * s, 0},
                         {vz * vx * cc - vy * s, vz * vy * cc + vx * s, vz * vz * cc + c, 0},
                         {0, 0, 0, 1}}};
                C = mul4(R, C);
            }
            if (cfg.saturation > 0 && cfg.channels > 1) {
                float sa = exp2f(z[iz++] * cfg.saturation_std);
                sa = (u[iu++] < cfg.saturation * p) ? sa : 1.f;
                M4 S;
                #pragma unroll
                for (int a = 0; a < 4; a++)
                    #pragma unroll
                    for (int b = 0; b < 4; b++) { float vvab = v[a] * v[b]; S.m[a][b] = vvab + ((a == b ? 1.f : 0.f) - vvab) * sa; }
                C = mul4(S, C);
            }
            #pragma unroll
            for (int a = 0; a < 4; a++)
                #pragma unroll
                for (int b = 0; b < 4; b++) {
                    Cmat[(size_t)i * 16 + a * 4 + b] = C.m[a][b];
                    if (A4) A4[(size_t)i * 16 + a * 4 + b] = (a < 3 && b < 3) ? C.m[a][b] : (a == b ? 1.f : 0.f);     // RGBA: colour block only
                }
            if (t4) {
                #pragma unroll
                for (int a = 0; a < 4; a++) t4[(size_t)i * 4 + a] = a < 3 ? C.m[a][3] : 0.f;
            }
        }
    }
    if (margin) {
        #pragma unroll
        for (int k = 0; k < 4; k++)
            for (int off = 32; off > 0; off >>= 1) mx[k] = fmaxf(mx[k], __shfl_down(mx[k], off, 64));
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) { for (int k = 0; k < 4; k++) red[wave][k] = mx[k]; }
        __syncthreads();
        if (threadIdx.x < 4) {
            const int k = threadIdx.x;
            float m = fmaxf(fmaxf(red[0][k], red[1][k]), fmaxf(red[2][k], red[3][k]));
            const float cx = (cfg.width - 1.f) / 2.f, cy = (cfg.height - 1.f) / 2.f;
            m += (k & 1) ? (cfg.hz_pad * 2.f - cy) : (cfg.hz_pad * 2.f - cx);
            m = fminf(fmaxf(m, 0.f), (k & 1) ? cfg.height - 1.f : cfg.width - 1.f);
            margin[k] = m;
        }
    }
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
    const uint32_t total = (uint32_t)n * (((uint32_t)out_w + 15u) >> 4) * (((uint32_t)out_h + 15u) >> 4);      // 16x16-pixel tiles
    uint32_t blocks = ((total + 7u) & ~7u) < 256u * 64u ? ((total + 7u) & ~7u) : 256u * 64u;        // a multiple of 8: a workgroup keeps its XCD across the stride
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
    const uint32_t total = (uint32_t)n * (((uint32_t)in_w + 15u) >> 4) * (((uint32_t)in_h + 15u) >> 4);        // 16x16-pixel tiles
    uint32_t blocks = ((total + 7u) & ~7u) < 256u * 64u ? ((total + 7u) & ~7u) : 256u * 64u;        // a multiple of 8: a workgroup keeps its XCD across the stride
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

extern "C" int mg_ada_params(const float* U, const float* Z, const float* p, int32_t n, int32_t nu, int32_t nz, const float* cfg, int32_t ncfg,
                             float* Ginv, float* margin, float* theta, float* Cmat, float* A4, float* t4, void* stream)
{
    MG_REQUIRE(U && Z && p && cfg, "ada_params: null pointer");
    MG_REQUIRE(ncfg == 26 + 18, "ada_params: the configuration vector has %d entries, expected 44", (int)ncfg);
    MG_REQUIRE(n >= 0 && nu >= 0 && nz >= 0, "ada_params: bad shape");
    if (n == 0) return MG_OK;
    AdaCfg c;
    const float* f = cfg;
    c.xflip = f[0]; c.rotate90 = f[1]; c.xint = f[2]; c.xint_max = f[3]; c.scale = f[4]; c.scale_std = f[5]; c.rotate = f[6]; c.rotate_max = f[7];
    c.aniso = f[8]; c.aniso_std = f[9]; c.xfrac = f[10]; c.xfrac_std = f[11]; c.brightness = f[12]; c.brightness_std = f[13]; c.contrast = f[14];
    c.contrast_std = f[15]; c.lumaflip = f[16]; c.hue = f[17]; c.hue_max = f[18]; c.saturation = f[19]; c.saturation_std = f[20];
    c.width = f[21]; c.height = f[22]; c.hz_pad = f[23]; c.channels = (int)f[24]; c.has_lr = (int)f[25];
    for (int i = 0; i < 9; i++) { c.left[i] = f[26 + i]; c.right[i] = f[35 + i]; }
    // the columns the enabled transforms consume, in the order the kernel walks them
    int need_u = 0, need_z = 0;
    if (Ginv) {
        if (c.xflip > 0) need_u += 2;
        if (c.rotate90 > 0) need_u += 2;
        if (c.xint > 0) need_u += 3;
        if (c.scale > 0) { need_u += 1; need_z += 1; }
        if (c.rotate > 0) need_u += 4;
        if (c.aniso > 0) { need_u += 1; need_z += 1; }
        if (c.xfrac > 0) { need_u += 1; need_z += 2; }
    }
    if (Cmat) {
        if (c.brightness > 0) { need_u += 1; need_z += 1; }
        if (c.contrast > 0) { need_u += 1; need_z += 1; }
        if (c.lumaflip > 0) need_u += 2;
        if (c.hue > 0 && c.channels > 1) need_u += 2;
        if (c.saturation > 0 && c.channels > 1) { need_u += 1; need_z += 1; }
    }
    MG_REQUIRE(nu >= need_u && nz >= need_z, "ada_params: random tables have %d / %d columns, the enabled transforms need %d / %d", (int)nu, (int)nz, need_u, need_z);
    MG_REQUIRE(!theta || (Ginv && c.has_lr), "ada_params: theta needs the geometric stage and the left / right factors");
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_ADA_PARAMS, s, (double)n * (nu + nz + 9 + 6 + 16 + 16 + 4) * 4);
    hipLaunchKernelGGL(ada_params_kernel, dim3(1), dim3(256), 0, s, U, Z, p, n, nu, nz, c, Ginv, margin, theta, Cmat, A4, t4);
    return mg_check_launch(MG_K_ADA_PARAMS, "ada_params");
}
