// upfirdn2d_sep.hip -- 12-tap separable FIR resampling by 2, BOTH axes in one pass over HBM.
//
// The geometric stage of the augmentation pipe (reference training/augment.py:315-333) is
//     reflect-pad -> upsample2d(sym6, x2) -> affine resample -> downsample2d(sym6, /2)
// and upfirdn2d.py:253-257 runs every 1-D filter as a row pass followed by a column pass: four launches and two
// intermediate images of 70-150 MB per call, 36 calls per training step (forward and adjoint).  Here one workgroup
// stages an input tile (with its 5/10-sample apron) in LDS, filters it along x into a second LDS tile and along y
// straight out of that tile into the coalesced stores, so each image crosses HBM once in and once out.
//
//   up   (up = 2, down = 1):  y[o] = g * sum_{t<6}  f[(q+10) - 2t] * x[(o+1-pad0-q)/2 + t],  q = (o+1-pad0) & 1
//   down (up = 1, down = 2):  y[o] = g * sum_{t<12} f[11 - t]      * x[2o - pad0 + t]
// per axis (f mirrored when `flip`), x = 0 outside the image: the receptive-field arithmetic of upfirdn2d.cu:43-48,60-75
// for fsize = 12 written per polyphase branch.  Taps are summed in ascending input order, sqrt(gain) is applied after
// each axis -- the order and the rounding points of the two-pass form.  float32, dense NCHW planes.
#include "mg_common.h"

namespace {

struct SepArgs;
void set_tiles(SepArgs& a, int tile_w, int tile_h, int64_t planes);

struct SepArgs {
    const float* x; const float* f; float* y;
    int ih, iw, oh, ow;
    int cx, cy;             // up: input index of tap 0 of output 0;  down: -pad0
    int flip;
    float gx, gy;           // gain per axis
    // up only: the input is a VIRTUAL reflect-padded image (torch.nn.functional.pad(mode='reflect')) of the sw x sh source at x:
    // ih / iw are the padded sizes, (rx0, ry0) the left / top margins.  sw = 0: x is the input itself.
    int sw, sh, rx0, ry0;
    // up with a reflect-padded input only: device float[4] = (mx0, my0, mx1, my1), the margins actually NEEDED around the source (<= rx0 ... by
    // construction); output tiles that lie entirely outside the up-sampled needed region are not computed (NULL: everything is)
    const float* win;
    // workgroup -> tile: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2; tile W of the list
    // (x fastest, then y, then plane) is given to workgroup (W % chunk) * 8 + W / chunk, so that every XCD works through one
    // contiguous eighth of the list and the aprons shared by neighbouring tiles are L2 hits (PMC before: the input was fetched
    // 1.6x (down) / 3.1x (up) its size, after: see profiles/r02_pmc_traffic.json)
    uint32_t tiles_x, tiles_y, total, chunk;
};

__device__ __forceinline__ bool tile_of_workgroup(const SepArgs& a, uint32_t& tx, uint32_t& ty, uint32_t& plane)
{
    const uint32_t w = (blockIdx.x & 7u) * a.chunk + (blockIdx.x >> 3);
    if (w >= a.total) return false;
    tx = w % a.tiles_x;
    const uint32_t t = w / a.tiles_x;
    ty = t % a.tiles_y;
    plane = t / a.tiles_y;
    return true;
}

__device__ __forceinline__ int reflect(int t, int n) { return t < 0 ? -t : (t >= n ? 2 * (n - 1) - t : t); }

__device__ __forceinline__ float tap(const float* __restrict__ f, int k, int flip) { return f[flip ? 11 - k : k]; }

// ---- up: 64 x 64 outputs per workgroup from a 38 x 38 input tile -----------------------------------------------------------
// QX / QY: parity of (1 - pad0) per axis = the polyphase branch of the EVEN outputs; the odd outputs take the other branch and,
// when Q = 1, start one input sample later.
template <int QX, int QY>
__global__ void __launch_bounds__(256) sep12_up2_kernel(SepArgs a)
{
    constexpr int TI = 38, PITCH = 40;
    __shared__ float s_in[TI * PITCH];
    __shared__ __attribute__((aligned(8))) float s_h[TI * 64];
    const int tid = threadIdx.x;
    uint32_t tile_x, tile_y, plane;
    if (!tile_of_workgroup(a, tile_x, tile_y, plane)) return;          // the whole workgroup, before any barrier
    const int ox0 = (int)tile_x * 64, oy0 = (int)tile_y * 64;
    if (a.win) {
        // the consumer (the affine warp) reads inside the image padded by ceil(win) only: in up-sampled coordinates of THIS (larger) padding
        // that is [2 (r0 - m0), 2 (r0 + size + m1)) per axis; two samples of slack for the warp's bilinear footprint (uniform test)
        const int mx0 = (int)ceilf(a.win[0]), my0 = (int)ceilf(a.win[1]), mx1 = (int)ceilf(a.win[2]), my1 = (int)ceilf(a.win[3]);
        const int lo_x = 2 * (a.rx0 - mx0) - 2, hi_x = 2 * (a.rx0 + a.sw + mx1) + 2, lo_y = 2 * (a.ry0 - my0) - 2, hi_y = 2 * (a.ry0 + a.sh + my1) + 2;
        if (ox0 + 64 <= lo_x || ox0 >= hi_x || oy0 + 64 <= lo_y || oy0 >= hi_y) return;
    }
    const int ix0 = (ox0 >> 1) + a.cx, iy0 = (oy0 >> 1) + a.cy;
    const float* __restrict__ xp = a.x + (size_t)plane * (a.sw ? (size_t)a.sh * a.sw : (size_t)a.ih * a.iw);

    float fxa[6], fxb[6], fya[6], fyb[6];       // uniform: scalar loads
    #pragma unroll
    for (int t = 0; t < 6; t++) {
        fxa[t] = tap(a.f, QX + 10 - 2 * t, a.flip);  fxb[t] = tap(a.f, 11 - QX - 2 * t, a.flip);
        fya[t] = tap(a.f, QY + 10 - 2 * t, a.flip);  fyb[t] = tap(a.f, 11 - QY - 2 * t, a.flip);
    }

    for (int e = tid; e < TI * TI; e += 256) {
        const int r = e / TI, c = e - r * TI;
        const int gy = iy0 + r, gx = ix0 + c;
        float v = 0.f;
        if ((unsigned)gy < (unsigned)a.ih && (unsigned)gx < (unsigned)a.iw)
            v = a.sw ? xp[(size_t)reflect(gy - a.ry0, a.sh) * a.sw + reflect(gx - a.rx0, a.sw)] : xp[(size_t)gy * a.iw + gx];
        s_in[r * PITCH + c] = v;
    }
    __syncthreads();

    // along x: item = (row r, output pair j)
    for (int e = tid; e < TI * 32; e += 256) {
        const int r = e >> 5, j = e & 31;
        const float* p = s_in + r * PITCH + j;
        float v[7];
        #pragma unroll
        for (int t = 0; t < 7; t++) v[t] = p[t];
        float sa = 0.f, sb = 0.f;
        #pragma unroll
        for (int t = 0; t < 6; t++) { sa = __builtin_fmaf(v[t], fxa[t], sa); sb = __builtin_fmaf(v[t + QX], fxb[t], sb); }
        *reinterpret_cast<float2*>(s_h + r * 64 + 2 * j) = make_float2(sa * a.gx, sb * a.gx);
    }
    __syncthreads();

    // along y: lane = output column, wave = 8 output row pairs; 13 (+1) tile rows slide through registers
    const int ox = tid & 63, jy0 = (tid >> 6) * 8;
    float h[14];
    #pragma unroll
    for (int t = 0; t < 13 + QY; t++) h[t] = s_h[(jy0 + t) * 64 + ox];
    const int gx = ox0 + ox;
    if (gx >= a.ow) return;
    float* __restrict__ yp = a.y + (size_t)plane * a.oh * a.ow + gx;
    #pragma unroll
    for (int q = 0; q < 8; q++) {
        float sa = 0.f, sb = 0.f;
        #pragma unroll
        for (int t = 0; t < 6; t++) { sa = __builtin_fmaf(h[q + t], fya[t], sa); sb = __builtin_fmaf(h[q + QY + t], fyb[t], sb); }
        const int gy = oy0 + 2 * (jy0 + q);
        if (gy < a.oh)     yp[(size_t)gy * a.ow] = sa * a.gy;
        if (gy + 1 < a.oh) yp[(size_t)(gy + 1) * a.ow] = sb * a.gy;
    }
}

// ---- down: 64 x TOH outputs per workgroup from a 138 x (2 TOH + 10) input tile -----------------------------------------------
template <int TOH>
__global__ void __launch_bounds__(256) sep12_down2_kernel(SepArgs a)
{
    constexpr int TIW = 138, TIH = 2 * TOH + 10, PITCH = 140, RPT = TOH / 4;      // RPT output rows per lane
    __shared__ __attribute__((aligned(8))) float s_in[TIH * PITCH];
    __shared__ float s_h[TIH * 64];
    const int tid = threadIdx.x;
    uint32_t tile_x, tile_y, plane;
    if (!tile_of_workgroup(a, tile_x, tile_y, plane)) return;          // the whole workgroup, before any barrier
    const int ox0 = (int)tile_x * 64, oy0 = (int)tile_y * TOH;
    const int ix0 = 2 * ox0 + a.cx, iy0 = 2 * oy0 + a.cy;
    const float* __restrict__ xp = a.x + (size_t)plane * a.ih * a.iw;

    float fx[12];
    #pragma unroll
    for (int t = 0; t < 12; t++) fx[t] = tap(a.f, 11 - t, a.flip);

    // the whole tile is requested before the first value is used: the loop below is the only trip to HBM of this workgroup,
    // and with a partial unroll its (TIH * TIW / 256 =) 23 loads per lane went out in three dependent batches
    constexpr int NLOAD = (TIH * TIW + 255) / 256;
    float stage[NLOAD];
    #pragma unroll
    for (int it = 0; it < NLOAD; it++) {
        const int e = tid + it * 256;
        const int r = e / TIW, c = e - r * TIW;
        const int gy = iy0 + r, gx = ix0 + c;
        const bool ok = e < TIH * TIW && (unsigned)gy < (unsigned)a.ih && (unsigned)gx < (unsigned)a.iw;
        const float v = xp[ok ? (size_t)gy * a.iw + gx : 0];        // unconditional load (index 0 is always inside the plane)
        stage[it] = ok ? v : 0.f;
    }
    #pragma unroll
    for (int it = 0; it < NLOAD; it++) {
        const int e = tid + it * 256;
        const int r = e / TIW, c = e - r * TIW;
        if (e < TIH * TIW) s_in[r * PITCH + c] = stage[it];
    }
    __syncthreads();

    // along x: lane j of row r reads its 12 samples as six 8-byte LDS reads (consecutive lanes, consecutive 8-byte words)
    for (int e = tid; e < TIH * 64; e += 256) {
        const int r = e >> 6, j = e & 63;
        const float2* p = reinterpret_cast<const float2*>(s_in + r * PITCH + 2 * j);
        float s = 0.f;
        #pragma unroll
        for (int t = 0; t < 6; t++) {
            const float2 v = p[t];
            s = __builtin_fmaf(v.x, fx[2 * t], s);
            s = __builtin_fmaf(v.y, fx[2 * t + 1], s);
        }
        s_h[r * 64 + j] = s * a.gx;
    }
    __syncthreads();

    // along y (the same taps: one 1-D filter for both axes)
    const int ox = tid & 63, ly0 = (tid >> 6) * RPT;
    float h[2 * (RPT - 1) + 12];
    #pragma unroll
    for (int t = 0; t < 2 * (RPT - 1) + 12; t++) h[t] = s_h[(2 * ly0 + t) * 64 + ox];
    const int gx = ox0 + ox;
    if (gx >= a.ow) return;
    float* __restrict__ yp = a.y + (size_t)plane * a.oh * a.ow + gx;
    #pragma unroll
    for (int q = 0; q < RPT; q++) {
        float s = 0.f;
        #pragma unroll
        for (int t = 0; t < 12; t++) s = __builtin_fmaf(h[2 * q + t], fx[t], s);
        const int gy = oy0 + ly0 + q;
        if (gy < a.oh) yp[(size_t)gy * a.ow] = s * a.gy;
    }
}

void set_tiles(SepArgs& a, int tile_w, int tile_h, int64_t planes)
{
    a.tiles_x = (uint32_t)((a.ow + tile_w - 1) / tile_w);
    a.tiles_y = (uint32_t)((a.oh + tile_h - 1) / tile_h);
    a.total = a.tiles_x * a.tiles_y * (uint32_t)planes;
    a.chunk = (a.total + 7u) / 8u;
}

} // namespace

// x [planes, ih, iw] -> y [planes, oh, ow], float32 dense; f: 12 float32 taps on the device; factor 2 in both axes, `up` nonzero:
// upsample (down = 1), else downsample (up = 1); pad*0 as mg_upfirdn2d (the caller sizes y); gain_axis is applied after each axis.
extern "C" int mg_upfirdn2d_sep12(const float* x, const float* f, float* y, int64_t planes, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                                  int32_t up, int32_t padx0, int32_t pady0, int32_t flip, float gain_axis, void* stream)
{
    MG_REQUIRE(x && f && y, "upfirdn2d_sep12: x, f and y must reside on the device");
    MG_REQUIRE(planes >= 0 && planes <= 65535 && ih >= 1 && iw >= 1 && oh >= 1 && ow >= 1, "upfirdn2d_sep12: bad shape");
    MG_REQUIRE((int64_t)ih * iw < (1ll << 31) && (int64_t)oh * ow < (1ll << 31), "upfirdn2d_sep12: plane too large");
    MG_REQUIRE(((int64_t)(ow + 63) / 64) * ((int64_t)(oh + 15) / 16) * planes < (1ll << 30), "upfirdn2d_sep12: too many tiles");
    if (planes == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_UPFIRDN2D_SEP, st, 4.0 * (double)planes * ((double)ih * iw + (double)oh * ow));
    SepArgs a;
    a.x = x; a.f = f; a.y = y; a.ih = ih; a.iw = iw; a.oh = oh; a.ow = ow; a.flip = flip ? 1 : 0; a.gx = gain_axis; a.gy = gain_axis;
    a.sw = a.sh = a.rx0 = a.ry0 = 0; a.win = nullptr;
    if (up) {
        const int mx = 1 - padx0, my = 1 - pady0, qx = mx & 1, qy = my & 1;
        a.cx = (mx - qx) >> 1; a.cy = (my - qy) >> 1;
        set_tiles(a, 64, 64, planes);
        const dim3 grid(a.chunk * 8u);
        if      (!qx && !qy) hipLaunchKernelGGL((sep12_up2_kernel<0, 0>), grid, dim3(256), 0, st, a);
        else if ( qx && !qy) hipLaunchKernelGGL((sep12_up2_kernel<1, 0>), grid, dim3(256), 0, st, a);
        else if (!qx &&  qy) hipLaunchKernelGGL((sep12_up2_kernel<0, 1>), grid, dim3(256), 0, st, a);
        else                 hipLaunchKernelGGL((sep12_up2_kernel<1, 1>), grid, dim3(256), 0, st, a);
    } else {
        a.cx = -padx0; a.cy = -pady0;
        constexpr int TOH = 16;
        set_tiles(a, 64, TOH, planes);
        hipLaunchKernelGGL((sep12_down2_kernel<TOH>), dim3(a.chunk * 8u), dim3(256), 0, st, a);
    }
    return mg_check_launch(MG_K_UPFIRDN2D_SEP, "upfirdn2d_sep12");
}

// upsample2d(pad(x, [mx0, mx1, my0, my1], mode='reflect'), f12, up=2) of augment.py:315-317 without the padded image:
// x [planes, sh, sw] -> y [planes, 2 (sh + my0 + my1), 2 (sw + mx0 + mx1)]; margins in [0, size - 1] (one reflection).
extern "C" int mg_reflect_pad_upsample12_window(const float* x, const float* f, float* y, int64_t planes, int32_t sh, int32_t sw,
                                                int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, float gain_axis, const float* window, void* stream);

extern "C" int mg_reflect_pad_upsample12(const float* x, const float* f, float* y, int64_t planes, int32_t sh, int32_t sw,
                                         int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, float gain_axis, void* stream)
{
    return mg_reflect_pad_upsample12_window(x, f, y, planes, sh, sw, mx0, mx1, my0, my1, gain_axis, nullptr, stream);
}

// ... with `window` (device float[4] = mx0, my0, mx1, my1 as computed on the device, each <= the padding above): the margins the consumer
// needs.  Output tiles outside the needed region are left UNWRITTEN -- the padding can then be a bound known to the host (a recorded HIP graph
// cannot read the data-dependent margin back) at the cost of the needed region only.
extern "C" int mg_reflect_pad_upsample12_window(const float* x, const float* f, float* y, int64_t planes, int32_t sh, int32_t sw,
                                                int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, float gain_axis, const float* window, void* stream)
{
    MG_REQUIRE(x && f && y, "reflect_pad_upsample12: x, f and y must reside on the device");
    MG_REQUIRE(planes >= 0 && planes <= 65535 && sh >= 1 && sw >= 1, "reflect_pad_upsample12: bad shape");
    MG_REQUIRE(mx0 >= 0 && mx1 >= 0 && my0 >= 0 && my1 >= 0 && mx0 < sw && mx1 < sw && my0 < sh && my1 < sh,
               "reflect_pad_upsample12: margins must lie in [0, size - 1]");
    const int64_t ih = (int64_t)sh + my0 + my1, iw = (int64_t)sw + mx0 + mx1;
    MG_REQUIRE(4 * ih * iw < (1ll << 31), "reflect_pad_upsample12: plane too large");
    MG_REQUIRE(((2 * iw + 63) / 64) * ((2 * ih + 63) / 64) * planes < (1ll << 30), "reflect_pad_upsample12: too many tiles");
    if (planes == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_UPFIRDN2D_SEP, st, 4.0 * (double)planes * ((double)sh * sw + 4.0 * (double)ih * iw));
    SepArgs a;
    a.x = x; a.f = f; a.y = y; a.ih = (int)ih; a.iw = (int)iw; a.oh = (int)(2 * ih); a.ow = (int)(2 * iw); a.flip = 0; a.gx = gain_axis; a.gy = gain_axis;
    a.sw = sw; a.sh = sh; a.rx0 = mx0; a.ry0 = my0; a.win = window;
    // upsample2d's padding for 12 taps: pad0 = (12 + 2 - 1) / 2 = 6 -> q = (1 - 6) & 1 = 1, c = (1 - 6 - 1) >> 1 = -3
    a.cx = -3; a.cy = -3;
    set_tiles(a, 64, 64, planes);
    hipLaunchKernelGGL((sep12_up2_kernel<1, 1>), dim3(a.chunk * 8u), dim3(256), 0, st, a);
    return mg_check_launch(MG_K_UPFIRDN2D_SEP, "reflect_pad_upsample12");
}
