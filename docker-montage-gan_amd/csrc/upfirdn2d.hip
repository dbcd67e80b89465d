// upfirdn2d.hip -- pad -> zero-upsample -> 2-D FIR -> decimate, fused, for gfx950.
// Replaces upfirdn2d_plugin.upfirdn2d (reference torch_utils/ops/upfirdn2d.cu:29-200,
// upfirdn2d.cpp:16-94).
//
// Two kernels:
//  * upfirdn2d_generic: any strides / filter / factors / dtype.  One lane per
//    output element, lanes ordered along the fastest output stride.  Integer
//    receptive field by mg_axis() -- the bit-exact index path.
//  * upfirdn2d_tiled<...>: dense NCHW, small filters, up/down in {1,2}: one
//    workgroup stages the input footprint of a PZ x TH x TW output tile into LDS
//    (fp32, zero filled outside the image), then every lane produces an MY x MX
//    register micro-tile with all filter taps held in registers.  The polyphase
//    structure (which taps hit non-zero samples) is resolved at compile time from
//    the tile phase PX/PY, so an up=2 output costs (FW/2)*(FH/2) MACs.
#include "mg_common.h"
#include <cstdlib>

namespace {

struct UfParams {
    const void* x; const float* f; void* y;
    int upx, upy, downx, downy, padx0, pady0, flip; float gain;
    int n, c, in_h, in_w, out_h, out_w, f_h, f_w;
    int64_t xs_n, xs_c, xs_h, xs_w, ys_n, ys_c, ys_h, ys_w, fs_h, fs_w;
    // mg_upfirdn2d_epilogue (channels-last 1:1 kernels): the modulated-convolution epilogue applied to the filter's output on its
    // way out -- y = bias_act(fma(T(fir), T(d[n,c]), noise[n,oy,ox] * strength), b, act, gain, clamp), the arithmetic and rounding
    // points of fma_bias_act_nhwc_fwd_kernel (csrc/fused.hip) on the T tensor the filter would have stored
    const float* ep_d; const float* ep_noise; const float* ep_strength; const void* ep_b;
    int64_t ep_noise_stride_n; float ep_alpha, ep_gain, ep_clamp;
};

#include "mg_act.h"
#include <type_traits>

// EPI = 0: none; 1 / 3: epilogue with the linear / leaky-ReLU activation
template <class T, int EPI, int VEC> struct UfEpiLane {
    float d[VEC], b[VEC], st;
    // The noise plane behind a buffer descriptor: a lane without a sample (no noise at all, or a pixel outside the output)
    // asks for NZ_BAD and gets 0 back, so the request can be issued early and unconditionally, next to the input rows it
    // travels with (tested at run time in the store loop, the load sat behind `s_waitcnt vmcnt(0)` and the wait drained
    // the prefetched input rows with it).
    static constexpr unsigned NZ_BAD = 0x80000000u;
    __device__ __forceinline__ static __amdgpu_buffer_rsrc_t noise_rsrc(const UfParams& p)
    {
        const int64_t bytes = p.ep_noise ? ((int64_t)(p.n - 1) * p.ep_noise_stride_n + (int64_t)p.out_h * p.out_w) * 4 : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ep_noise ? p.ep_noise : p.f), 0, (int)bytes, 0x00020000);
    }
    __device__ __forceinline__ static float noise_at(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
    {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
    }
    __device__ __forceinline__ void init(const UfParams& p, uint32_t n, uint32_t cvi)
    {
        if (EPI == 0) return;
        // the lane's channel vector of dcoefs and bias as 16-byte loads behind ONE test each (element by element, each load
        // sat behind its own test and `s_waitcnt vmcnt(0)`: 2 * VEC serialised round trips before the first input row)
        typedef float f4_t __attribute__((ext_vector_type(4)));
        struct alignas(16) V { T v[VEC]; };
        #pragma unroll
        for (int e = 0; e < VEC; e++) { d[e] = 1.f; b[e] = 0.f; }
        if (p.ep_d) {
            const f4_t* dp = (const f4_t*)(p.ep_d + (size_t)n * p.c + cvi * VEC);
            f4_t q[VEC / 4];
            #pragma unroll
            for (int h = 0; h < VEC / 4; h++) q[h] = dp[h];
            #pragma unroll
            for (int h = 0; h < VEC / 4; h++) {
                d[4 * h] = to_acc<T>(from_acc<T>(q[h].x)); d[4 * h + 1] = to_acc<T>(from_acc<T>(q[h].y));
                d[4 * h + 2] = to_acc<T>(from_acc<T>(q[h].z)); d[4 * h + 3] = to_acc<T>(from_acc<T>(q[h].w));
            }
        }
        if (p.ep_b) {
            const V bq = *(const V*)((const T*)p.ep_b + cvi * VEC);
            #pragma unroll
            for (int e = 0; e < VEC; e++) b[e] = to_acc<T>(bq.v[e]);
        }
        st = (p.ep_noise && p.ep_strength) ? *p.ep_strength : 1.f;
    }
    // v: the filter's float32 result of one pixel's channel vector; nz: the pixel's noise sample
    __device__ __forceinline__ void apply(const UfParams& p, float (&v)[VEC], float nz) const
    {
        if (EPI == 0) return;
        const bool has_nz = p.ep_noise != nullptr;
        if constexpr (std::is_same<T, mg_bf16>::value) {
            // bfloat16: two values per instruction (v_pk_mul / v_pk_fma / v_pk_add_f32, v_cvt_pk_bf16_f32) -- the epilogue doubled the
            // VALU count of the row kernel, which made the fused pass VALU-bound (214 us against 136 us for the filter alone).  The same
            // operations in the same order as ba_eval: T(fir) * d, + noise * strength, -> T, + b, lrelu as max(t, alpha t) (0 <= alpha <= 1,
            // checked by the entry point), * gain, clamp as v_med3 (a NaN comes out as -clamp, as the comparison chain leaves it).
            typedef float e2_t __attribute__((ext_vector_type(2)));
            typedef __bf16 ebf2_t __attribute__((ext_vector_type(2)));
            auto round2 = [](e2_t x) -> e2_t {
                const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, ebf2_t));
                e2_t r; r.x = __builtin_bit_cast(float, u << 16); r.y = __builtin_bit_cast(float, u & 0xFFFF0000u); return r;
            };
            const e2_t nz2 = {nz, nz}, st2 = {st, st}, g2 = {p.ep_gain, p.ep_gain}, a2 = {p.ep_alpha, p.ep_alpha};
            const float lim = p.ep_clamp;
            #pragma unroll
            for (int h = 0; h < VEC / 2; h++) {
                e2_t t = {v[2 * h], v[2 * h + 1]};
                const e2_t d2 = {d[2 * h], d[2 * h + 1]}, b2 = {b[2 * h], b[2 * h + 1]};
                t = round2(t) * d2;                                         // the filter's output is a T tensor
                const e2_t tn = __builtin_elementwise_fma(nz2, st2, t);
                t = has_nz ? tn : t;                                        // (a select: without noise the value must stay untouched, -0 included)
                t = round2(t) + b2;
                if (EPI == 3) {
                    const e2_t m = t * a2;
                    t.x = __builtin_fmaxf(t.x, m.x); t.y = __builtin_fmaxf(t.y, m.y);
                }
                t = t * g2;
                if (lim >= 0.f) { t.x = __builtin_amdgcn_fmed3f(t.x, -lim, lim); t.y = __builtin_amdgcn_fmed3f(t.y, -lim, lim); }
                v[2 * h] = t.x; v[2 * h + 1] = t.y;
            }
            return;
        }
        const BaConst<float> k = {p.ep_alpha, p.ep_gain, p.ep_clamp};
        #pragma unroll
        for (int e = 0; e < VEC; e++) {
            float t = (float)to_acc<T>(from_acc<T>(v[e])) * d[e];           // the filter's output is a T tensor
            t = has_nz ? __builtin_fmaf(nz, st, t) : t;                     // (a select, not a branch)
            t = (float)to_acc<T>(from_acc<T>(t));
            v[e] = ba_eval<float, (EPI == 0 ? 1 : EPI)>(0, t, b[e], 0.f, 0.f, 1.f, k);
        }
    }
};

// Integer receptive field of one output coordinate along one axis
// (reference upfirdn2d.cu:43-48 / 60-65).  Same arithmetic as oracle orc_axis().
__device__ __forceinline__ void mg_axis(int o, int up, int down, int pad0, int fsize, int in_size, int flip,
                                        int& in0, int& count, int& f0)
{
    int mid = o * down + up - 1 - pad0;
    int i0 = min(max(mg_floor_div(mid, up), 0), in_size);
    int i1 = min(max(mg_floor_div(mid + fsize, up), 0), in_size);
    int ff = mid + fsize - (i0 + 1) * up;
    if (flip) ff = fsize - 1 - ff;
    in0 = i0; count = i1 - i0; f0 = ff;
}

__global__ void upfirdn2d_index_table_kernel(int32_t* table, int out_size, int in_size, int up, int down, int pad0,
                                             int fsize, int flip)
{
    int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= out_size) return;
    int a, b, c;
    mg_axis(o, up, down, pad0, fsize, in_size, flip, a, b, c);
    table[3 * o] = a; table[3 * o + 1] = b; table[3 * o + 2] = c;
}

// ---------------------------------------------------------------------------
// Generic kernel.  CL = channel is the fastest output dimension.
template <class T, bool CL>
__global__ void __launch_bounds__(256) upfirdn2d_generic(UfParams p, uint32_t total)
{
    typedef typename Acc<T>::type acc_t;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        int ox, oy, c, n;
        uint32_t r = idx;
        if (CL) { c = r % p.c; r /= p.c; ox = r % p.out_w; r /= p.out_w; oy = r % p.out_h; n = r / p.out_h; }
        else    { ox = r % p.out_w; r /= p.out_w; oy = r % p.out_h; r /= p.out_h; c = r % p.c; n = r / p.c; }
        int ix0, w, fx0, iy0, h, fy0;
        mg_axis(ox, p.upx, p.downx, p.padx0, p.f_w, p.in_w, p.flip, ix0, w, fx0);
        mg_axis(oy, p.upy, p.downy, p.pady0, p.f_h, p.in_h, p.flip, iy0, h, fy0);
        const int64_t fstep_x = (p.flip ? p.upx : -p.upx) * p.fs_w;
        const int64_t fstep_y = (p.flip ? p.upy : -p.upy) * p.fs_h;
        const T* xp = (const T*)p.x + n * p.xs_n + c * p.xs_c + iy0 * p.xs_h + ix0 * p.xs_w;
        const float* fp = p.f + fy0 * p.fs_h + fx0 * p.fs_w;
        acc_t v = 0;
        for (int j = 0; j < h; j++) {
            const T* xr = xp + j * p.xs_h;
            const float* fr = fp + j * fstep_y;
            for (int i = 0; i < w; i++)
                v += to_acc<T>(xr[i * p.xs_w]) * (acc_t)fr[i * fstep_x];
        }
        v *= (acc_t)p.gain;
        ((T*)p.y)[n * p.ys_n + c * p.ys_c + oy * p.ys_h + ox * p.ys_w] = from_acc<T>(v);
    }
}

// ---------------------------------------------------------------------------
// Tiled kernel, dense NCHW.
//   UPX/UPY, DNX/DNY  resampling factors (1 or 2)
//   FW/FH             template filter footprint; shorter filters are zero extended
//                     at the far end of the flipped filter (harmless, reference
//                     upfirdn2d.cu:118-130 does the same)
//   PX/PY             phase of the tile origin: (tile_out0*DN + UP-1-pad0) mod UP
//   TXT x TYT lanes per plane, MX x MY outputs per lane, PZ planes per workgroup
template <int UPX, int UPY, int DNX, int DNY, int FW, int FH, int PX, int PY, int TXT, int TYT, int MX, int MY>
struct TileGeom {
    static constexpr int PZ = 256 / (TXT * TYT);
    static constexpr int TW = TXT * MX, TH = TYT * MY;                        // output tile
    static constexpr int TIW = (PX + (TW - 1) * DNX + FW - 1) / UPX + 1;     // input footprint
    static constexpr int TIH = (PY + (TH - 1) * DNY + FH - 1) / UPY + 1;
    static constexpr int RX = (PX + (MX - 1) * DNX + FW - 1) / UPX + 1;      // per-lane register window
    static constexpr int RY = (PY + (MY - 1) * DNY + FH - 1) / UPY + 1;
    static constexpr int LANE_DX = MX * DNX / UPX, LANE_DY = MY * DNY / UPY;  // window origin step per lane
    // LDS row layout: neighbouring lanes read columns LANE_DX apart, which would be a LANE_DX-way bank conflict in a
    // plain row.  Columns are de-interleaved by LD = LANE_DX: column c lives at (c % LD) * KP + c / LD, so that for a
    // fixed window offset the lanes of a row hit consecutive words; KP odd keeps the staging writes conflict free too.
    static constexpr int LD = LANE_DX < 1 ? 1 : LANE_DX;
    static constexpr int KP = ((TIW + LD - 1) / LD) | 1;
    static constexpr int LDW = LD * KP;
    static constexpr int col(int c) { return (c % LD) * KP + c / LD; }
    static_assert((MX * DNX) % UPX == 0 && (MY * DNY) % UPY == 0, "micro tile must span whole input samples");
    static_assert((TW * DNX) % UPX == 0 && (TH * DNY) % UPY == 0, "tile phase must repeat");
};

template <class T, int UPX, int UPY, int DNX, int DNY, int FW, int FH, int PX, int PY, int TXT, int TYT, int MX, int MY>
__global__ void __launch_bounds__(256) upfirdn2d_tiled(UfParams p, int tiles_x, int tiles_y)
{
    typedef TileGeom<UPX, UPY, DNX, DNY, FW, FH, PX, PY, TXT, TYT, MX, MY> G;
    __shared__ float sx[G::PZ][G::TIH][G::LDW];

    // taps of the flipped, zero-extended filter; uniform -> scalar registers
    float ft[FH][FW];
    #pragma unroll
    for (int ky = 0; ky < FH; ky++)
        #pragma unroll
        for (int kx = 0; kx < FW; kx++) {
            float v = 0.f;
            if (kx < p.f_w && ky < p.f_h) {
                int sx_ = p.flip ? kx : p.f_w - 1 - kx;
                int sy_ = p.flip ? ky : p.f_h - 1 - ky;
                v = p.f[sy_ * p.fs_h + sx_ * p.fs_w];
            }
            ft[ky][kx] = v;
        }

    const int planes = p.n * p.c;
    const int tile_x = blockIdx.x % tiles_x;
    const int tile_y = (blockIdx.x / tiles_x) % tiles_y;
    const int plane0 = (blockIdx.x / (tiles_x * tiles_y)) * G::PZ;
    const int out_x0 = tile_x * G::TW, out_y0 = tile_y * G::TH;
    // first input sample of the tile footprint (may be negative / beyond the image)
    const int in_x0 = mg_floor_div(out_x0 * DNX + UPX - 1 - p.padx0, UPX);
    const int in_y0 = mg_floor_div(out_y0 * DNY + UPY - 1 - p.pady0, UPY);

    // stage the footprint, zero filled, as fp32
    for (int i = threadIdx.x; i < G::PZ * G::TIH * G::TIW; i += 256) {
        int rx = i % G::TIW, t = i / G::TIW, ry = t % G::TIH, pz = t / G::TIH;
        int ix = in_x0 + rx, iy = in_y0 + ry, plane = plane0 + pz;
        float v = 0.f;
        if (ix >= 0 && iy >= 0 && ix < p.in_w && iy < p.in_h && plane < planes)
            v = (float)to_acc<T>(((const T*)p.x)[(int64_t)plane * p.in_h * p.in_w + (int64_t)iy * p.in_w + ix]);
        sx[pz][ry][G::col(rx)] = v;
    }
    __syncthreads();

    const int tx = threadIdx.x % TXT, ty = (threadIdx.x / TXT) % TYT, pz = threadIdx.x / (TXT * TYT);
    const int plane = plane0 + pz;

    float acc[MY][MX];
    #pragma unroll
    for (int j = 0; j < MY; j++)
        #pragma unroll
        for (int i = 0; i < MX; i++) acc[j][i] = 0.f;

    // walk the register window row by row; every (row, output row) pair that meets a
    // non-zero polyphase tap is a compile-time decision
    #pragma unroll
    for (int ry = 0; ry < G::RY; ry++) {
        float row[G::RX];
        #pragma unroll
        for (int rx = 0; rx < G::RX; rx++) row[rx] = sx[pz][ty * G::LANE_DY + ry][(rx % G::LD) * G::KP + tx + rx / G::LD];   // = col(tx * LANE_DX + rx)
        #pragma unroll
        for (int j = 0; j < MY; j++) {
            // up-sampled position of output j's first tap relative to the lane window
            const int my = PY + j * DNY;                 // = phase + offset, in up-sampled samples
            const int iy_first = my / UPY;               // first input row touched
            const int fy_first = (iy_first + 1) * UPY - my - 1;   // its tap in the flipped filter
            const int ky = fy_first + (ry - iy_first) * UPY;      // tap that row `ry` meets
            if (ry < iy_first || ky >= FH) continue;
            #pragma unroll
            for (int i = 0; i < MX; i++) {
                const int mx = PX + i * DNX;
                const int ix_first = mx / UPX;
                const int fx_first = (ix_first + 1) * UPX - mx - 1;
                #pragma unroll
                for (int kx = fx_first, rx = ix_first; kx < FW; kx += UPX, rx++)
                    acc[j][i] += row[rx] * ft[ky][kx];
            }
        }
    }

    if (plane >= planes) return;
    T* yp = (T*)p.y + (int64_t)plane * p.out_h * p.out_w;
    #pragma unroll
    for (int j = 0; j < MY; j++) {
        int oy = out_y0 + ty * MY + j;
        if (oy >= p.out_h) continue;
        T* rowp = yp + (int64_t)oy * p.out_w + out_x0 + tx * MX;
        if constexpr (MX == 4 || MX == 2) {
            // the lane's MX outputs are adjacent: one aligned vector store (16 / 8 / 4 bytes) instead of MX scalar ones
            // whose lanes sit 4*MX bytes apart (a quarter of each write transaction used)
            if (out_x0 + tx * MX + MX - 1 < p.out_w) {
                T vals[MX];
                #pragma unroll
                for (int i = 0; i < MX; i++) vals[i] = from_acc<T>(acc[j][i] * p.gain);
                const uintptr_t a = (uintptr_t)rowp;
                if ((a & (sizeof(T) * MX - 1)) == 0) {
                    struct alignas(sizeof(T) * MX) VS { T v[MX]; };
                    VS o;
                    #pragma unroll
                    for (int i = 0; i < MX; i++) o.v[i] = vals[i];
                    *(VS*)rowp = o;
                    continue;
                }
                if (MX == 4 && (a & (sizeof(T) * 2 - 1)) == 0) {
                    struct alignas(sizeof(T) * 2) V2 { T v[2]; };
                    V2 lo, hi;
                    lo.v[0] = vals[0]; lo.v[1] = vals[1]; hi.v[0] = vals[MX - 2]; hi.v[1] = vals[MX - 1];
                    ((V2*)rowp)[0] = lo; ((V2*)rowp)[1] = hi;
                    continue;
                }
            }
        }
        #pragma unroll
        for (int i = 0; i < MX; i++) {
            int ox = out_x0 + tx * MX + i;
            if (ox < p.out_w) rowp[i] = from_acc<T>(acc[j][i] * p.gain);
        }
    }
}

template <class T, int UPX, int UPY, int DNX, int DNY, int FW, int FH, int PX, int PY, int TXT, int TYT, int MX, int MY>
void launch_tiled_geom(const UfParams& p, hipStream_t s)
{
    typedef TileGeom<UPX, UPY, DNX, DNY, FW, FH, PX, PY, TXT, TYT, MX, MY> G;
    int tiles_x = (p.out_w + G::TW - 1) / G::TW, tiles_y = (p.out_h + G::TH - 1) / G::TH;
    int planes = p.n * p.c;
    int64_t blocks = (int64_t)tiles_x * tiles_y * ((planes + G::PZ - 1) / G::PZ);
    hipLaunchKernelGGL((upfirdn2d_tiled<T, UPX, UPY, DNX, DNY, FW, FH, PX, PY, TXT, TYT, MX, MY>), dim3((uint32_t)blocks),
                       dim3(256), 0, s, p, tiles_x, tiles_y);
}

// choose lanes-per-plane from the output size: big planes get 32x8 lanes (128x32 outputs
// at MX=MY=4), small planes pack several per workgroup
template <class T, int UPX, int UPY, int DNX, int DNY, int FW, int FH, int PX, int PY, int MX, int MY>
void launch_tiled_phase(const UfParams& p, hipStream_t s)
{
    if (p.out_w > 16 * MX)      launch_tiled_geom<T, UPX, UPY, DNX, DNY, FW, FH, PX, PY, 32, 8, MX, MY>(p, s);
    else if (p.out_w > 8 * MX)  launch_tiled_geom<T, UPX, UPY, DNX, DNY, FW, FH, PX, PY, 16, 16, MX, MY>(p, s);
    else if (p.out_w > 4 * MX)  launch_tiled_geom<T, UPX, UPY, DNX, DNY, FW, FH, PX, PY, 8, 8, MX, MY>(p, s);
    else                        launch_tiled_geom<T, UPX, UPY, DNX, DNY, FW, FH, PX, PY, 4, 4, MX, MY>(p, s);
}

template <class T, int UPX, int UPY, int DNX, int DNY, int FW, int FH, int MX, int MY>
void launch_tiled(const UfParams& p, hipStream_t s)
{
    // tile origins are multiples of TW*DN (a multiple of UP), so the phase is the same for every tile
    int px = ((UPX - 1 - p.padx0) % UPX + UPX) % UPX;
    int py = ((UPY - 1 - p.pady0) % UPY + UPY) % UPY;
    if (UPX == 1 && UPY == 1) { launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, 0, 0, MX, MY>(p, s); return; }
    if (UPX == 2 && UPY == 2) {
        if (px == 0 && py == 0) launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, 0, 0, MX, MY>(p, s);
        if (px == 1 && py == 0) launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, (UPX > 1), 0, MX, MY>(p, s);
        if (px == 0 && py == 1) launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, 0, (UPY > 1), MX, MY>(p, s);
        if (px == 1 && py == 1) launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, (UPX > 1), (UPY > 1), MX, MY>(p, s);
        return;
    }
    if (UPX == 2) {
        if (px == 0) launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, 0, 0, MX, MY>(p, s);
        else         launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, (UPX > 1), 0, MX, MY>(p, s);
        return;
    }
    if (py == 0) launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, 0, 0, MX, MY>(p, s);
    else         launch_tiled_phase<T, UPX, UPY, DNX, DNY, FW, FH, 0, (UPY > 1), MX, MY>(p, s);
}


// ---------------------------------------------------------------------------
// Streaming kernel for up = down = 1, filters up to 4x4, dense NCHW -- the blur that follows every
// up-sampling convolution of G and precedes every down-sampling convolution of D (and their
// gradients): the dominant upfirdn2d configuration of the training step.
//
// No LDS, no barrier.  One WAVE owns a strip of 64*CPL input columns of one plane and walks down
// RH output rows: per input row every lane loads CPL adjacent samples with ONE coalesced access
// (4 bytes: one float, or an aligned pair of 16-bit values), the FW-1 right-hand neighbours arrive
// by wave shuffles, and a ring of FH accumulators per column carries the vertical taps, so every
// input sample is read from HBM once (plus a (FW-1)/(64*CPL) / (FH-1)/RH halo) and costs FW*FH MACs.
template <class T> struct StreamTraits { static constexpr int CPL = 1; };
template <> struct StreamTraits<mg_half> { static constexpr int CPL = 2; };
template <> struct StreamTraits<mg_bf16> { static constexpr int CPL = 2; };

template <class T> __device__ __forceinline__ float raw16_to_f32(unsigned r);
template <> __device__ __forceinline__ float raw16_to_f32<mg_bf16>(unsigned r) { return __builtin_bit_cast(float, r << 16); }
template <> __device__ __forceinline__ float raw16_to_f32<mg_half>(unsigned r) { return (float)__builtin_bit_cast(mg_half, (unsigned short)r); }
template <class T> __device__ __forceinline__ unsigned f32_to_raw16(float v) { return (unsigned)__builtin_bit_cast(unsigned short, from_acc<T>(v)); }

constexpr int US_FW = 4, US_FH = 4;

template <class T, int RH, int OWW>
__global__ void __launch_bounds__(256) upfirdn2d_stream(UfParams p, int nstrips, int nchunks, uint32_t ntasks)
{
    constexpr int CPL = StreamTraits<T>::CPL;
    constexpr int FW = US_FW, FH = US_FH;
    constexpr int ES = (int)sizeof(T);
    constexpr unsigned BAD = 0x80000000u;                           // a byte offset no tensor reaches (sizes < 2^31 bytes)
    static_assert(OWW <= 64 * CPL - (FW - 1) && OWW % CPL == 0, "strip too wide");
    constexpr int GRP = 2 * FH;                                     // input rows in flight per wave
    const int lane = threadIdx.x & 63;
    const uint32_t task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (task >= ntasks) return;                                     // wave-uniform
    const int strip = task % nstrips, chunk = (task / nstrips) % nchunks, plane = task / (nstrips * nchunks);

    float ft[FH][FW];
    #pragma unroll
    for (int ky = 0; ky < FH; ky++)
        #pragma unroll
        for (int kx = 0; kx < FW; kx++) {
            float v = 0.f;
            if (kx < p.f_w && ky < p.f_h)
                v = p.f[(p.flip ? ky : p.f_h - 1 - ky) * p.fs_h + (p.flip ? kx : p.f_w - 1 - kx) * p.fs_w];
            ft[ky][kx] = v * p.gain;                                // fold the gain into the taps
        }

    // Buffer descriptors over the WHOLE tensors: every access is range checked by the hardware (out-of-range
    // loads read 0, out-of-range stores are dropped), so the inner loop has no bounds branches at all.
    const uint32_t in_bytes = (uint32_t)(((int64_t)p.n * p.c * p.in_h * p.in_w * ES + 3) & ~3ll);
    const uint32_t out_bytes = (uint32_t)((int64_t)p.n * p.c * p.out_h * p.out_w * ES);
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)in_bytes, 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)out_bytes, 0x00020000);

    // wave-uniform geometry (up = down = 1: first input sample of output o is o - pad0)
    const int out_x0 = strip * OWW, out_y0 = chunk * RH;
    const int in_x0 = out_x0 - p.padx0, in_y0 = out_y0 - p.pady0;
    const int rows_out = min(RH, p.out_h - out_y0), rows_in = rows_out + FH - 1;
    const int in_pitch = p.in_w * ES, out_pitch = p.out_w * ES;
    // byte offset of (input row in_y0 + j, column in_x0) and of (output row out_y0 + j - (FH-1), column out_x0)
    int in_off = (int)(((int64_t)plane * p.in_h + in_y0) * p.in_w + in_x0) * ES;
    int out_off = (int)(((int64_t)plane * p.out_h + out_y0 - (FH - 1)) * p.out_w + out_x0) * ES;
    int iy = in_y0;

    // per-lane invariants
    const int col0 = in_x0 + lane * CPL;
    const bool v0 = col0 >= 0 && col0 < p.in_w;
    const bool v1 = col0 + 1 >= 0 && col0 + 1 < p.in_w;
    const bool vm1 = col0 - 1 >= 0 && col0 - 1 < p.in_w;
    const int oxl = lane * CPL;
    const bool st0 = oxl < OWW && out_x0 + oxl < p.out_w;
    const bool st1 = oxl + 1 < OWW && out_x0 + oxl + 1 < p.out_w;
    const unsigned lane_in = (unsigned)(lane * CPL * ES);
    // 16-bit pairs: which halves of the lane's aligned pair lie inside the image for an even (M0) / odd (M1) row start
    const unsigned M0 = (v0 ? 0xffffu : 0u) | (v1 ? 0xffff0000u : 0u);
    const unsigned M1 = (vm1 ? 0xffffu : 0u) | (v0 ? 0xffff0000u : 0u);
    // store offsets inside the row; BAD drops the lane
    const unsigned so_a = (CPL == 1 ? st0 : (st0 && st1)) ? (unsigned)(oxl * ES) : BAD;      // full-width store
    const unsigned so_e0 = (CPL == 2 && st0 && !st1) ? (unsigned)(oxl * ES) : BAD;            // 16-bit edge lane (aligned rows)
    const unsigned so_h0 = st0 ? (unsigned)(oxl * ES) : BAD, so_h1 = st1 ? (unsigned)((oxl + 1) * ES) : BAD;   // misaligned rows
    const bool any_edge = __builtin_amdgcn_readfirstlane(__ballot(so_e0 != BAD) != 0);

    float acc[FH][CPL];
    #pragma unroll
    for (int a = 0; a < FH; a++)
        #pragma unroll
        for (int q = 0; q < CPL; q++) acc[a][q] = 0.f;

    // Software pipeline: the loads of row group g+1 are in flight while group g is being consumed.
    unsigned raw[GRP], raw_next[GRP];
    int par[GRP], par_next[GRP];
    int j_load = 0;
    auto load_group = [&](unsigned (&dst)[GRP], int (&dpar)[GRP]) {
        #pragma unroll
        for (int r = 0; r < GRP; r++) {
            const bool row_ok = (j_load + r < rows_in) && iy >= 0 && iy < p.in_h;      // wave-uniform
            if constexpr (CPL == 1) {
                dpar[r] = 0;
                const unsigned ro = row_ok ? (unsigned)in_off : BAD;
                dst[r] = __builtin_amdgcn_raw_buffer_load_b32(rin, lane_in + ro, 0, 0);
            } else {
                // one aligned 4-byte access per lane: the pair starting at the even element below the row start;
                // par = parity of the row start tells how the pair lines up with the lane's columns
                dpar[r] = (in_off >> 1) & 1;
                const unsigned ro = row_ok ? (unsigned)(in_off & ~3) : BAD;
                dst[r] = __builtin_amdgcn_raw_buffer_load_b32(rin, lane_in + ro, 0, 0);
            }
            in_off += in_pitch; iy++;
        }
        j_load += GRP;
    };
    load_group(raw, par);
    for (int jb = 0; jb < rows_in; jb += GRP) {
        if (jb + GRP < rows_in) load_group(raw_next, par_next);    // wave-uniform
        #pragma unroll
        for (int r = 0; r < GRP; r++) {
            const int j = jb + r;
            if (j >= rows_in) break;                                // wave-uniform
            float v[CPL + FW - 1];
            if constexpr (CPL == 1) {
                v[0] = v0 ? __builtin_bit_cast(float, raw[r]) : 0.f;
                #pragma unroll
                for (int k = 1; k < FW; k++) v[k] = __shfl_down(v[0], k, 64);
            } else {
                const unsigned w0 = raw[r] & (par[r] ? M1 : M0);
                const unsigned w1 = (unsigned)__shfl_down((int)w0, 1, 64);
                const unsigned w2 = (unsigned)__shfl_down((int)w0, 2, 64);
                // columns col0 .. col0+4 are elements par .. par+4 of the 6-element run (w0 | w1 | w2)
                const unsigned e[6] = {w0 & 0xffffu, w0 >> 16, w1 & 0xffffu, w1 >> 16, w2 & 0xffffu, w2 >> 16};
                #pragma unroll
                for (int k = 0; k < CPL + FW - 1; k++) v[k] = raw16_to_f32<T>(par[r] ? e[k + 1] : e[k]);
            }
            // row j carries tap ky of output row j - ky; ring slot (j - ky) mod FH = (r - ky) mod FH
            #pragma unroll
            for (int ky = 0; ky < FH; ky++) {
                const int slot = ((r - ky) % FH + FH) % FH;
                #pragma unroll
                for (int q = 0; q < CPL; q++)
                    #pragma unroll
                    for (int kx = 0; kx < FW; kx++) acc[slot][q] = __builtin_fmaf(v[q + kx], ft[ky][kx], acc[slot][q]);
            }
            // output row j - (FH-1) is complete; its slot is (r + 1) mod FH
            const int done = (r + 1) % FH;
            if (j >= FH - 1) {                                      // wave-uniform
                if constexpr (CPL == 1) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)to_acc<T>(from_acc<T>(acc[done][0]))), rout,
                                                          so_a + (unsigned)out_off, 0, 0);
                } else {
                    const unsigned a = f32_to_raw16<T>(acc[done][0]), b = f32_to_raw16<T>(acc[done][CPL - 1]);
                    if ((out_off & 3) == 0) {                       // wave-uniform: 4-byte aligned row start
                        __builtin_amdgcn_raw_buffer_store_b32(a | (b << 16), rout, so_a + (unsigned)out_off, 0, 0);
                        if (any_edge) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)a, rout, so_e0 + (unsigned)out_off, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)a, rout, so_h0 + (unsigned)out_off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)b, rout, so_h1 + (unsigned)out_off, 0, 0);
                    }
                }
            }
            out_off += out_pitch;
            #pragma unroll
            for (int q = 0; q < CPL; q++) acc[done][q] = 0.f;
        }
        #pragma unroll
        for (int r = 0; r < GRP; r++) { raw[r] = raw_next[r]; par[r] = par_next[r]; }
    }
}

template <class T>
bool try_stream(const UfParams& p, hipStream_t s)
{
    constexpr int CPL = StreamTraits<T>::CPL;
    // outputs per strip: a whole number of 128-byte lines, so that neighbouring waves (which may sit on
    // different XCDs with non-coherent L2s) never write parts of the same line
    static int oww_env = getenv("MG_STREAM_OWW") ? atoi(getenv("MG_STREAM_OWW")) : 0;
    constexpr int OWW_WIDE = (64 * CPL - (US_FW - 1)) & ~(CPL - 1), OWW_LINE = 128 / (int)sizeof(T) * (CPL == 1 ? 1 : 1);
    const bool wide = (oww_env != 2);
    const int OWW = wide ? OWW_WIDE : OWW_LINE;
    constexpr int RH = 64;
    if (!(p.upx == 1 && p.upy == 1 && p.downx == 1 && p.downy == 1 && p.f_w <= US_FW && p.f_h <= US_FH)) return false;
    if (p.out_w < 48) return false;                                 // small planes: the tiled kernel packs several per workgroup
    if ((int64_t)p.n * p.c * p.in_h * p.in_w * (int64_t)sizeof(T) >= (1ll << 31) - 8 || (int64_t)p.n * p.c * p.out_h * p.out_w * (int64_t)sizeof(T) >= (1ll << 31) - 8)
        return false;                                               // 32-bit buffer offsets
    int nstrips = (p.out_w + OWW - 1) / OWW, nchunks = (p.out_h + RH - 1) / RH;
    int64_t ntasks = (int64_t)nstrips * nchunks * p.n * p.c;
    if (ntasks > (int64_t)UINT32_MAX / 2) return false;
    uint32_t blocks = (uint32_t)((ntasks + 3) / 4);
    if (wide) hipLaunchKernelGGL((upfirdn2d_stream<T, RH, OWW_WIDE>), dim3(blocks), dim3(256), 0, s, p, nstrips, nchunks, (uint32_t)ntasks);
    else      hipLaunchKernelGGL((upfirdn2d_stream<T, RH, OWW_LINE>), dim3(blocks), dim3(256), 0, s, p, nstrips, nchunks, (uint32_t)ntasks);
    return true;
}

// ---------------------------------------------------------------------------
// Channels-last kernel: dense NHWC input and output (the layout MIOpen's implicit-GEMM convolutions run in
// natively), filters up to 4x4, up/down in {1,2}.  One lane owns ONE 16-byte channel vector of MX adjacent
// output pixels and walks down a chunk of output rows MY at a time; neighbouring lanes hold neighbouring
// channel vectors, then neighbouring pixels, so every access of a wave is one contiguous run of the row.
// No LDS: the FW-1 / FH-1 overlaps between lanes and between consecutive row blocks are served by L1/L2.
// The polyphase structure is the compile-time logic of the tiled kernel (PX/PY = phase of the lane origin).
// All global accesses go through range-checked buffer descriptors; a lane outside the image swaps its
// offset for one beyond the buffer (loads return 0, stores are dropped), so there are no border branches.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <class T> struct NhwcVec { static constexpr int VEC = 16 / (int)sizeof(T); };

template <class T> __device__ __forceinline__ void nhwc_unpack(const u32x4_t& r, float (&v)[NhwcVec<T>::VEC])
{
    // (copy the lanes to scalars first: __builtin_bit_cast applied directly to a vector element reads lane 0)
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    if constexpr (sizeof(T) == 4) {
        #pragma unroll
        for (int q = 0; q < 4; q++) v[q] = __builtin_bit_cast(float, w[q]);
    } else {
        #pragma unroll
        for (int q = 0; q < 4; q++) { v[2 * q] = raw16_to_f32<T>(w[q] & 0xffffu); v[2 * q + 1] = raw16_to_f32<T>(w[q] >> 16); }
    }
}

template <class T> __device__ __forceinline__ u32x4_t nhwc_pack(const float (&v)[NhwcVec<T>::VEC])
{
    u32x4_t r;
    unsigned w[4];
    if constexpr (sizeof(T) == 4) {
        #pragma unroll
        for (int q = 0; q < 4; q++) w[q] = __builtin_bit_cast(unsigned, v[q]);
    } else {
        #pragma unroll
        for (int q = 0; q < 4; q++) w[q] = f32_to_raw16<T>(v[2 * q]) | (f32_to_raw16<T>(v[2 * q + 1]) << 16);
    }
    r.x = w[0]; r.y = w[1]; r.z = w[2]; r.w = w[3];
    return r;
}

typedef float f2_t __attribute__((ext_vector_type(2)));

// Pairs of adjacent channels share every tap, so the MACs are issued as packed v_pk_fma_f32 (two IEEE fmas per lane
// per instruction -- same results, half the VALU issue slots; the 4x4 FIR is VALU-bound at 16 MACs per output).
template <class T> __device__ __forceinline__ void nhwc_unpack2(const u32x4_t& r, f2_t (&v)[NhwcVec<T>::VEC / 2])
{
    float t[NhwcVec<T>::VEC];
    nhwc_unpack<T>(r, t);
    #pragma unroll
    for (int h = 0; h < NhwcVec<T>::VEC / 2; h++) v[h] = (f2_t){t[2 * h], t[2 * h + 1]};
}

template <class T> __device__ __forceinline__ u32x4_t nhwc_pack2(const f2_t (&v)[NhwcVec<T>::VEC / 2])
{
    float t[NhwcVec<T>::VEC];
    #pragma unroll
    for (int h = 0; h < NhwcVec<T>::VEC / 2; h++) { t[2 * h] = v[h].x; t[2 * h + 1] = v[h].y; }
    return nhwc_pack<T>(t);
}

struct NhwcGrid { uint32_t cv, gx, nchunks, total, rows_per_chunk; FastDiv cv_div, gx_div, chunk_div; };

template <class T, int UP, int DN, int PX, int PY, int MX, int MY, int EPI = 0>
__global__ void __launch_bounds__(256) upfirdn2d_nhwc(UfParams p, NhwcGrid g)
{
    constexpr int VEC = NhwcVec<T>::VEC, FW = 4, FH = 4, ES = (int)sizeof(T);
    constexpr int RX = (PX + (MX - 1) * DN + FW - 1) / UP + 1, RY = (PY + (MY - 1) * DN + FH - 1) / UP + 1;
    constexpr int LANE_DX = MX * DN / UP, LANE_DY = MY * DN / UP;
    static_assert((MX * DN) % UP == 0 && (MY * DN) % UP == 0, "micro tile must span whole input samples");
    constexpr unsigned BAD = 0x80000000u;

    float ft[FH][FW];
    #pragma unroll
    for (int ky = 0; ky < FH; ky++)
        #pragma unroll
        for (int kx = 0; kx < FW; kx++) {
            float v = 0.f;
            if (kx < p.f_w && ky < p.f_h)
                v = p.f[(p.flip ? ky : p.f_h - 1 - ky) * p.fs_h + (p.flip ? kx : p.f_w - 1 - kx) * p.fs_w];
            ft[ky][kx] = v * p.gain;
        }

    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= g.total) return;
    const uint32_t t1 = fd_div(t, g.cv_div), cvi = t - t1 * g.cv;
    const uint32_t t2 = fd_div(t1, g.gx_div), gxi = t1 - t2 * g.gx;
    const uint32_t n = fd_div(t2, g.chunk_div), chunk = t2 - n * g.nchunks;

    const uint32_t in_bytes = (uint32_t)((int64_t)p.n * p.c * p.in_h * p.in_w * ES);
    const uint32_t out_bytes = (uint32_t)((int64_t)p.n * p.c * p.out_h * p.out_w * ES);
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)in_bytes, 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)out_bytes, 0x00020000);

    UfEpiLane<T, EPI, VEC> epi;
    epi.init(p, n, cvi);
    const auto rnz = UfEpiLane<T, EPI, VEC>::noise_rsrc(p);
    const int pix_bytes = p.c * ES;
    const int ox0 = (int)gxi * MX;
    const int ix0 = mg_floor_div(UP - 1 - p.padx0, UP) + (int)gxi * LANE_DX;        // first input column of the lane window
    int col_off[RX];                                                                // byte offset inside a row, or -1
    #pragma unroll
    for (int rx = 0; rx < RX; rx++) {
        const int ix = ix0 + rx;
        col_off[rx] = (ix >= 0 && ix < p.in_w) ? ix * pix_bytes + (int)cvi * 16 : -1;
    }
    unsigned col_u[RX], col_bad[RX];
    #pragma unroll
    for (int rx = 0; rx < RX; rx++) { col_u[rx] = col_off[rx] >= 0 ? (unsigned)col_off[rx] : 0u; col_bad[rx] = col_off[rx] >= 0 ? 0u : BAD; }
    unsigned st_off[MX];
    #pragma unroll
    for (int i = 0; i < MX; i++) st_off[i] = (ox0 + i < p.out_w) ? (unsigned)((ox0 + i) * pix_bytes + (int)cvi * 16) : BAD;
    const int in_pitch = p.in_w * pix_bytes, out_pitch = p.out_w * pix_bytes;
    const int oy_begin = (int)(chunk * g.rows_per_chunk), oy_end = min(p.out_h, oy_begin + (int)g.rows_per_chunk);
    const int iy_base = mg_floor_div(UP - 1 - p.pady0, UP);

    for (int oy0 = oy_begin; oy0 < oy_end; oy0 += MY) {
        const int iy0 = iy_base + oy0 / MY * LANE_DY;
        f2_t acc[MY][MX][VEC / 2];
        #pragma unroll
        for (int j = 0; j < MY; j++)
            #pragma unroll
            for (int i = 0; i < MX; i++)
                #pragma unroll
                for (int e = 0; e < VEC / 2; e++) acc[j][i][e] = (f2_t){0.f, 0.f};

        float nzv[MY][MX];
        if (EPI) {
            #pragma unroll
            for (int j = 0; j < MY; j++)
                #pragma unroll
                for (int i = 0; i < MX; i++) {
                    const int oy = oy0 + j;
                    const bool in = oy < p.out_h && ox0 + i < p.out_w;
                    nzv[j][i] = epi.noise_at(rnz, in ? (unsigned)(((int64_t)n * p.ep_noise_stride_n + (int64_t)oy * p.out_w + ox0 + i) * 4) : epi.NZ_BAD);
                }
        }
        // rows are requested RG at a time (at most 16 vectors = 64 VGPRs in flight), then consumed
        constexpr int RG = (RX * RY <= 16) ? RY : (16 / RX);
        #pragma unroll
        for (int ry0 = 0; ry0 < RY; ry0 += RG) {
            u32x4_t raw[RG][RX];
            #pragma unroll
            for (int r = 0; r < RG; r++) {
                const int ry = ry0 + r;
                if (ry >= RY) break;
                const int iy = iy0 + ry;
                // (branch-free: an absent row / column ORs the top bit into the offset and the load returns zeros; as a select the
                // compiler built an if / else with one load per side)
                const unsigned row_bad = (iy >= 0 && iy < p.in_h) ? 0u : BAD;
                const unsigned row_off = (unsigned)(((int)n * p.in_h + iy) * in_pitch);
                #pragma unroll
                for (int rx = 0; rx < RX; rx++)
                    raw[r][rx] = __builtin_amdgcn_raw_buffer_load_b128(rin, (row_off + col_u[rx]) | row_bad | col_bad[rx], 0, 0);
            }
            #pragma unroll
            for (int r = 0; r < RG; r++) {
                const int ry = ry0 + r;
                if (ry >= RY) break;
                f2_t row[RX][VEC / 2];
                #pragma unroll
                for (int rx = 0; rx < RX; rx++) nhwc_unpack2<T>(raw[r][rx], row[rx]);
                #pragma unroll
                for (int j = 0; j < MY; j++) {
                    const int my = PY + j * DN;
                    const int iy_first = my / UP;
                    const int fy_first = (iy_first + 1) * UP - my - 1;
                    const int ky = fy_first + (ry - iy_first) * UP;
                    if (ry < iy_first || ky >= FH) continue;
                    #pragma unroll
                    for (int i = 0; i < MX; i++) {
                        const int mx = PX + i * DN;
                        const int ix_first = mx / UP;
                        const int fx_first = (ix_first + 1) * UP - mx - 1;
                        #pragma unroll
                        for (int kx = fx_first, rx = ix_first; kx < FW; kx += UP, rx++) {
                            const f2_t tap = {ft[ky][kx], ft[ky][kx]};
                            #pragma unroll
                            for (int e = 0; e < VEC / 2; e++) acc[j][i][e] = __builtin_elementwise_fma(row[rx][e], tap, acc[j][i][e]);
                        }
                    }
                }
            }
        }
        #pragma unroll
        for (int j = 0; j < MY; j++) {
            const int oy = oy0 + j;
            const unsigned row_off = (unsigned)(((int)n * p.out_h + oy) * out_pitch);
            #pragma unroll
            for (int i = 0; i < MX; i++) {
                const unsigned off = (oy < p.out_h && st_off[i] != BAD) ? row_off + st_off[i] : BAD;
                if (EPI) {
                    float t[VEC];
                    #pragma unroll
                    for (int h = 0; h < VEC / 2; h++) { t[2 * h] = acc[j][i][h].x; t[2 * h + 1] = acc[j][i][h].y; }
                    epi.apply(p, t, nzv[j][i]);
                    __builtin_amdgcn_raw_buffer_store_b128(nhwc_pack<T>(t), rout, off, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(nhwc_pack2<T>(acc[j][i]), rout, off, 0, 0);
                }
            }
        }
    }
}

// up = down = 1 in channels-last: row streaming.  The lane walks DOWN its chunk one input row at a time; a row
// meets tap ky of output row (row - ky), so a ring of FH accumulator vectors carries the vertical taps and every
// input row is loaded exactly once per chunk (the block kernel above re-reads the FH-1 rows two consecutive
// blocks share, and those re-reads partly miss L2: 1.49x read traffic measured).  The next row is requested
// before the current one is consumed.
template <class T, int EPI = 0>
__global__ void __launch_bounds__(256) upfirdn2d_nhwc_rows(UfParams p, NhwcGrid g)
{
    constexpr int VEC = NhwcVec<T>::VEC, FW = 4, FH = 4, ES = (int)sizeof(T);
    constexpr unsigned BAD = 0x80000000u;

    float ft[FH][FW];
    #pragma unroll
    for (int ky = 0; ky < FH; ky++)
        #pragma unroll
        for (int kx = 0; kx < FW; kx++) {
            float v = 0.f;
            if (kx < p.f_w && ky < p.f_h)
                v = p.f[(p.flip ? ky : p.f_h - 1 - ky) * p.fs_h + (p.flip ? kx : p.f_w - 1 - kx) * p.fs_w];
            ft[ky][kx] = v * p.gain;
        }

    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= g.total) return;
    const uint32_t t1 = fd_div(t, g.cv_div), cvi = t - t1 * g.cv;
    const uint32_t t2 = fd_div(t1, g.gx_div), gxi = t1 - t2 * g.gx;
    const uint32_t n = fd_div(t2, g.chunk_div), chunk = t2 - n * g.nchunks;

    const uint32_t in_bytes = (uint32_t)((int64_t)p.n * p.c * p.in_h * p.in_w * ES);
    const uint32_t out_bytes = (uint32_t)((int64_t)p.n * p.c * p.out_h * p.out_w * ES);
    const auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)in_bytes, 0x00020000);
    const auto rout = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)out_bytes, 0x00020000);

    UfEpiLane<T, EPI, VEC> epi;
    epi.init(p, n, cvi);
    const auto rnz = UfEpiLane<T, EPI, VEC>::noise_rsrc(p);
    const int pix_bytes = p.c * ES;
    const int ox = (int)gxi, ix0 = ox - p.padx0;
    int col_off[FW];
    #pragma unroll
    for (int kx = 0; kx < FW; kx++) {
        const int ix = ix0 + kx;
        col_off[kx] = (ix >= 0 && ix < p.in_w) ? ix * pix_bytes + (int)cvi * 16 : -1;
    }
    const unsigned st_off = (unsigned)(ox * pix_bytes + (int)cvi * 16);             // ox < out_w by construction of the grid
    const int in_pitch = p.in_w * pix_bytes, out_pitch = p.out_w * pix_bytes;
    const int oy_begin = (int)(chunk * g.rows_per_chunk);
    const int rows_out = min((int)g.rows_per_chunk, p.out_h - oy_begin), rows_in = rows_out + FH - 1;
    int iy = oy_begin - p.pady0;
    int in_row_off = ((int)n * p.in_h + iy) * in_pitch;
    unsigned out_row_off = (unsigned)(((int)n * p.out_h + oy_begin - (FH - 1)) * out_pitch);      // row of output j - (FH-1) at j = 0

    // one input row of the lane's FW columns.  Branch-free: a column outside the image, a row outside the image and a row past the
    // end of the chunk (`live` false) all turn into an out-of-range offset by OR-ing the top bit in, and the load returns zeros --
    // written as `ok ? offset : BAD` the compiler built an if / else with one load per side, and every join cost a `vmcnt(0)`
    unsigned col_u[FW], col_bad[FW];
    #pragma unroll
    for (int kx = 0; kx < FW; kx++) { col_u[kx] = col_off[kx] >= 0 ? (unsigned)col_off[kx] : 0u; col_bad[kx] = col_off[kx] >= 0 ? 0u : BAD; }
    auto load_row = [&](u32x4_t (&dst)[FW], bool live) __attribute__((always_inline)) {
        const unsigned row_bad = (live && iy >= 0 && iy < p.in_h) ? 0u : BAD;
        #pragma unroll
        for (int kx = 0; kx < FW; kx++)
            dst[kx] = __builtin_amdgcn_raw_buffer_load_b128(rin, ((unsigned)in_row_off + col_u[kx]) | row_bad | col_bad[kx], 0, 0);
        iy++; in_row_off += in_pitch;
    };

    f2_t acc[FH][VEC / 2];
    #pragma unroll
    for (int a = 0; a < FH; a++)
        #pragma unroll
        for (int e = 0; e < VEC / 2; e++) acc[a][e] = (f2_t){0.f, 0.f};

    // noise of output row `orow` of this lane's column (0 for rows in front of the chunk: they are never stored)
    const unsigned nz_base = (unsigned)(((int64_t)n * p.ep_noise_stride_n + ox) * 4), nz_pitch = (unsigned)(p.out_w * 4);
    auto load_noise = [&](int orow) -> float {
        if (!EPI) return 0.f;
        return UfEpiLane<T, EPI, VEC>::noise_at(rnz, (orow >= oy_begin && orow < p.out_h) ? nz_base + (unsigned)orow * nz_pitch : UfEpiLane<T, EPI, VEC>::NZ_BAD);
    };
    // two rows ahead: row j + 2 is requested before row j is consumed, so a lane has 8 sixteen-byte loads in flight (one row ahead: 4 --
    // ~49 KB per CU against the ~39 KB that 5 TB/s at ~2 us of latency asks for; the loop waited for every next row).  The rows live in
    // a ring of FH buffers indexed by the unrolled row counter: no register copies (a `cur = nxt` copy made the compiler wait for the
    // youngest load), no branches around the requests (a row past the end is requested out of range), whole groups of FH rows without
    // a trip test -- so the wait in front of row j can leave row j + 1 in flight (`s_waitcnt vmcnt(8..11)` in the ISA).  178 VGPRs = 2 waves per
    // SIMD (126 = 4 before): the same 64 KB in flight per CU from half the waves, +6-10 % (122 against 130 us at [32,64,257^2 -> 256^2], 59.5
    // against 64 us at 128^2, fused epilogue 195 against 214 us); capped at 168 VGPRs for 3 waves the ring spills and loses (134 us).
    u32x4_t ring[FH][FW];
    float nzr[FH];
    load_row(ring[0], true);
    nzr[0] = load_noise(oy_begin - (FH - 1));
    load_row(ring[1], rows_in > 1);
    nzr[1] = load_noise(oy_begin + 1 - (FH - 1));
    auto step = [&](auto R, int j) __attribute__((always_inline)) {
        constexpr int r = decltype(R)::value;
        load_row(ring[(r + 2) % FH], j + 2 < rows_in);
        nzr[(r + 2) % FH] = load_noise(oy_begin + j + 2 - (FH - 1));
        f2_t row[FW][VEC / 2];
        #pragma unroll
        for (int kx = 0; kx < FW; kx++) nhwc_unpack2<T>(ring[r][kx], row[kx]);
        #pragma unroll
        for (int ky = 0; ky < FH; ky++) {
            constexpr int dummy = 0; (void)dummy;
            const int slot = ((r - ky) % FH + FH) % FH;                         // output row j - ky
            #pragma unroll
            for (int kx = 0; kx < FW; kx++) {
                const f2_t tap = {ft[ky][kx], ft[ky][kx]};
                #pragma unroll
                for (int e = 0; e < VEC / 2; e++) acc[slot][e] = __builtin_elementwise_fma(row[kx][e], tap, acc[slot][e]);
            }
        }
        constexpr int done = (r + 1) % FH;                                      // output row j - (FH-1) is complete
        // (rows in front of the chunk are "stored" out of range: no branch)
        const unsigned st_bad = j >= FH - 1 ? 0u : BAD;
        if (EPI) {
            float t[VEC];
            #pragma unroll
            for (int h = 0; h < VEC / 2; h++) { t[2 * h] = acc[done][h].x; t[2 * h + 1] = acc[done][h].y; }
            epi.apply(p, t, nzr[r]);
            __builtin_amdgcn_raw_buffer_store_b128(nhwc_pack<T>(t), rout, (out_row_off + st_off) | st_bad, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(nhwc_pack2<T>(acc[done]), rout, (out_row_off + st_off) | st_bad, 0, 0);
        }
        out_row_off += (unsigned)out_pitch;
        #pragma unroll
        for (int e = 0; e < VEC / 2; e++) acc[done][e] = (f2_t){0.f, 0.f};
    };
    int j = 0;
    for (; j + FH <= rows_in; j += FH) {
        step(std::integral_constant<int, 0>(), j); step(std::integral_constant<int, 1>(), j + 1);
        step(std::integral_constant<int, 2>(), j + 2); step(std::integral_constant<int, 3>(), j + 3);
    }
    static_assert(FH == 4, "the row ring is written out for 4 taps");
    if (j < rows_in) step(std::integral_constant<int, 0>(), j);
    if (j + 1 < rows_in) step(std::integral_constant<int, 1>(), j + 1);
    if (j + 2 < rows_in) step(std::integral_constant<int, 2>(), j + 2);
}

static thread_local int g_epi_act = 0;          // set by mg_upfirdn2d_epilogue around its launch (0: plain upfirdn2d)

template <class T, int UP, int DN, int PX, int PY>
void launch_nhwc_phase(const UfParams& p, hipStream_t s)
{
    constexpr int VEC = NhwcVec<T>::VEC, MX = (UP == 2 ? 2 : 1), MY = 4;
    NhwcGrid g;
    g.cv = (uint32_t)(p.c / VEC);
    g.gx = (uint32_t)((p.out_w + MX - 1) / MX);
    // rows per chunk: a multiple of MY; 32 output rows keep the inter-chunk halo small while leaving >= 8 workgroups per CU on the big layers
    // (row streaming from 96 rows up: 16-row chunks at 128^2 re-read 3 rows in 16 where the block kernel re-reads 3 in 4)
    const bool stream_rows = UP == 1 && DN == 1 && p.out_h >= 96;
    g.rows_per_chunk = (stream_rows && p.out_h < 192) ? 16 : 32;
    // (swept on MI355X, bf16 batch 32: 32 / 16 rows beat 64 / 32, 128 / 64 and 16 / 8 -- profiles/r03_fir_chunk_sweep.log: the row
    // loop waits for each next row, so longer chunks buy less halo traffic but not more bytes in flight)
    g.nchunks = (uint32_t)((p.out_h + (int)g.rows_per_chunk - 1) / (int)g.rows_per_chunk);
    g.total = (uint32_t)p.n * g.nchunks * g.gx * g.cv;
    g.cv_div = make_fastdiv(g.cv); g.gx_div = make_fastdiv(g.gx); g.chunk_div = make_fastdiv(g.nchunks);
    // tall images: row streaming (every input row read once per chunk); short ones: the block kernel (more parallel slack)
    const dim3 grid((g.total + 255u) / 256u);
    if constexpr (UP == 1 && DN == 1 && sizeof(T) == 2) {
        if (g_epi_act == 1 || g_epi_act == 3) {      // mg_upfirdn2d_epilogue: same kernels, epilogue on the way out
            const bool rows = stream_rows;
            if (g_epi_act == 3) { if (rows) hipLaunchKernelGGL((upfirdn2d_nhwc_rows<T, 3>), grid, dim3(256), 0, s, p, g); else hipLaunchKernelGGL((upfirdn2d_nhwc<T, 1, 1, 0, 0, MX, MY, 3>), grid, dim3(256), 0, s, p, g); }
            else                { if (rows) hipLaunchKernelGGL((upfirdn2d_nhwc_rows<T, 1>), grid, dim3(256), 0, s, p, g); else hipLaunchKernelGGL((upfirdn2d_nhwc<T, 1, 1, 0, 0, MX, MY, 1>), grid, dim3(256), 0, s, p, g); }
            return;
        }
    }
    if (stream_rows) hipLaunchKernelGGL((upfirdn2d_nhwc_rows<T>), grid, dim3(256), 0, s, p, g);
    else hipLaunchKernelGGL((upfirdn2d_nhwc<T, UP, DN, PX, PY, MX, MY>), grid, dim3(256), 0, s, p, g);
}

template <class T>
bool try_nhwc(const UfParams& p, hipStream_t s)
{
    constexpr int VEC = NhwcVec<T>::VEC;
    const int64_t C = p.c;
    const bool dense = p.c > 1 && p.xs_c == 1 && p.xs_w == C && p.xs_h == C * p.in_w && p.xs_n == C * p.in_w * p.in_h &&
                       p.ys_c == 1 && p.ys_w == C && p.ys_h == C * p.out_w && p.ys_n == C * p.out_w * p.out_h;
    if (!dense || p.c % VEC != 0 || p.f_w > 4 || p.f_h > 4) return false;
    if ((((uintptr_t)p.x | (uintptr_t)p.y) & 15) != 0) return false;
    if (p.upx != p.upy || p.downx != p.downy) return false;
    const int64_t lim = (1ll << 31) - 64;
    if (C * p.n * p.in_h * p.in_w * (int64_t)sizeof(T) >= lim || C * p.n * p.out_h * p.out_w * (int64_t)sizeof(T) >= lim) return false;
    if ((int64_t)p.n * ((p.out_h + 31) / 32) * p.out_w * (C / VEC) >= (int64_t)UINT32_MAX / 2) return false;
    const int px = ((p.upx - 1 - p.padx0) % p.upx + p.upx) % p.upx, py = ((p.upy - 1 - p.pady0) % p.upy + p.upy) % p.upy;
    if (p.upx == 1 && p.downx == 1) { launch_nhwc_phase<T, 1, 1, 0, 0>(p, s); return true; }
    if (p.upx == 1 && p.downx == 2) { launch_nhwc_phase<T, 1, 2, 0, 0>(p, s); return true; }
    if (p.upx == 2 && p.downx == 1) {
        if (px == 0 && py == 0) launch_nhwc_phase<T, 2, 1, 0, 0>(p, s);
        if (px == 1 && py == 0) launch_nhwc_phase<T, 2, 1, 1, 0>(p, s);
        if (px == 0 && py == 1) launch_nhwc_phase<T, 2, 1, 0, 1>(p, s);
        if (px == 1 && py == 1) launch_nhwc_phase<T, 2, 1, 1, 1>(p, s);
        return true;
    }
    return false;
}

static thread_local bool g_last_was_stream = false;
static thread_local bool g_last_was_nhwc = false;

// Returns true when a tiled (or the streaming) specialisation took the call.
template <class T>
bool try_tiled(const UfParams& p, hipStream_t s)
{
    const bool dense = p.xs_w == 1 && p.xs_h == p.in_w && p.xs_c == (int64_t)p.in_h * p.in_w &&
                       p.xs_n == (int64_t)p.c * p.in_h * p.in_w && p.ys_w == 1 && p.ys_h == p.out_w &&
                       p.ys_c == (int64_t)p.out_h * p.out_w && p.ys_n == (int64_t)p.c * p.out_h * p.out_w;
    if (!dense) {
        if (try_nhwc<T>(p, s)) { g_last_was_nhwc = true; return true; }
        return false;
    }
    if (sizeof(T) <= 4 && try_stream<T>(p, s)) { g_last_was_stream = true; return true; }
    const int ux = p.upx, uy = p.upy, dx = p.downx, dy = p.downy, fw = p.f_w, fh = p.f_h;
#define UF_CASE(UX, UY, DX, DY, FW_, FH_, MX_, MY_)                                                       \
    if (ux == UX && uy == UY && dx == DX && dy == DY && fw <= FW_ && fh <= FH_) {                         \
        launch_tiled<T, UX, UY, DX, DY, FW_, FH_, MX_, MY_>(p, s); return true; }
    // 2-D filters of the networks ([1,3,3,1] outer product)
    UF_CASE(1, 1, 1, 1, 4, 4, 4, 4)
    UF_CASE(2, 2, 1, 1, 4, 4, 4, 4)
    UF_CASE(1, 1, 2, 2, 4, 4, 2, 2)
    // separable passes of ADA (12-tap sym6), and their gradients
    UF_CASE(1, 1, 1, 1, 12, 1, 4, 4)
    UF_CASE(1, 1, 1, 1, 1, 12, 4, 4)
    UF_CASE(2, 1, 1, 1, 12, 1, 4, 4)
    UF_CASE(1, 2, 1, 1, 1, 12, 4, 4)
    UF_CASE(1, 1, 2, 1, 12, 1, 4, 4)
    UF_CASE(1, 1, 1, 2, 1, 12, 4, 4)
#undef UF_CASE
    return false;
}

} // namespace

extern "C" int mg_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                            const int32_t x_size[4], const int64_t x_stride[4],
                            const int32_t f_size[2], const int64_t f_stride[2],
                            const int32_t y_size[4], const int64_t y_stride[4],
                            int upx, int upy, int downx, int downy, int padx0, int pady0, int flip, float gain,
                            void* stream)
{
    MG_REQUIRE(x && f && y, "x must reside on the device");
    MG_REQUIRE(f_size[0] >= 1 && f_size[1] >= 1, "f must be at least 1x1");                 // upfirdn2d.cpp:26
    MG_REQUIRE(upx >= 1 && upy >= 1, "upsampling factor must be at least 1");               // upfirdn2d.cpp:27
    MG_REQUIRE(downx >= 1 && downy >= 1, "downsampling factor must be at least 1");         // upfirdn2d.cpp:28
    MG_REQUIRE(y_size[2] >= 1 && y_size[3] >= 1, "output must be at least 1x1");            // upfirdn2d.cpp:34
    MG_REQUIRE(x_size[0] == y_size[0] && x_size[1] == y_size[1], "x and y must agree in batch and channels");
    int64_t xn = (int64_t)x_size[0] * x_size[1] * x_size[2] * x_size[3];
    int64_t yn = (int64_t)y_size[0] * y_size[1] * y_size[2] * y_size[3];
    MG_REQUIRE(xn <= INT32_MAX, "x is too large");                                           // upfirdn2d.cpp:22
    MG_REQUIRE(yn <= INT32_MAX, "output is too large");                                      // upfirdn2d.cpp:36
    int es = mg_dtype_size(dtype);
    MG_REQUIRE(es > 0, "unsupported dtype %d", dtype);
    if (yn == 0) return MG_OK;

    UfParams p;
    p.x = x; p.f = f; p.y = y;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0;
    p.flip = flip ? 1 : 0; p.gain = gain;
    p.n = x_size[0]; p.c = x_size[1]; p.in_h = x_size[2]; p.in_w = x_size[3];
    p.out_h = y_size[2]; p.out_w = y_size[3]; p.f_h = f_size[0]; p.f_w = f_size[1];
    p.xs_n = x_stride[0]; p.xs_c = x_stride[1]; p.xs_h = x_stride[2]; p.xs_w = x_stride[3];
    p.ys_n = y_stride[0]; p.ys_c = y_stride[1]; p.ys_h = y_stride[2]; p.ys_w = y_stride[3];
    p.fs_h = f_stride[0]; p.fs_w = f_stride[1];
    p.ep_d = p.ep_noise = p.ep_strength = nullptr; p.ep_b = nullptr; p.ep_noise_stride_n = 0; p.ep_alpha = 0.f; p.ep_gain = 1.f; p.ep_clamp = -1.f;
    hipStream_t s = (hipStream_t)stream;

    bool tiled = false;
    {
        MgLaunchScope scope(MG_K_UPFIRDN2D_TILED, s, (double)(xn + yn) * es);
        g_last_was_stream = false; g_last_was_nhwc = false;
        switch (dtype) {
        case MG_F32:  tiled = try_tiled<float>(p, s); break;
        case MG_F16:  tiled = try_tiled<mg_half>(p, s); break;
        case MG_BF16: tiled = try_tiled<mg_bf16>(p, s); break;
        default: break;
        }
        if (!tiled) scope.kid = -1;
        else if (g_last_was_stream) scope.kid = MG_K_UPFIRDN2D_STREAM;
        else if (g_last_was_nhwc) scope.kid = MG_K_UPFIRDN2D_NHWC;
    }
    if (tiled) return mg_check_launch(g_last_was_stream ? MG_K_UPFIRDN2D_STREAM : g_last_was_nhwc ? MG_K_UPFIRDN2D_NHWC : MG_K_UPFIRDN2D_TILED, "upfirdn2d");

    MgLaunchScope scope(MG_K_UPFIRDN2D_GENERIC, s, (double)(xn + yn) * es);
    uint32_t total = (uint32_t)yn;
    uint32_t blocks = (total + 255u) / 256u;
    if (blocks > 256u * 16u) blocks = 256u * 16u;
    const bool cl = (p.ys_c == 1 && p.c > 1);
    MG_DISPATCH_DTYPE(dtype, {
        if (cl) hipLaunchKernelGGL((upfirdn2d_generic<T, true>), dim3(blocks), dim3(256), 0, s, p, total);
        else    hipLaunchKernelGGL((upfirdn2d_generic<T, false>), dim3(blocks), dim3(256), 0, s, p, total);
    });
    return mg_check_launch(MG_K_UPFIRDN2D_GENERIC, "upfirdn2d");
}

extern "C" int mg_upfirdn2d_epilogue(const void* x, const float* f, void* y, int dtype, const int32_t x_size[4], const int32_t f_size[2],
                                     const int64_t f_stride[2], const int32_t y_size[4], int padx0, int pady0, int flip, float gain,
                                     const float* d, const float* noise, const float* strength, const void* b, int64_t noise_stride_n,
                                     int act, float alpha, float act_gain, float clamp, void* stream)
{
    MG_REQUIRE(x && f && y, "x must reside on the device");
    MG_REQUIRE(dtype == MG_F16 || dtype == MG_BF16, "upfirdn2d_epilogue: float16 / bfloat16 channels-last tensors only");
    MG_REQUIRE(act == 1 || act == 3, "upfirdn2d_epilogue: activation must be linear (1) or lrelu (3)");
    MG_REQUIRE(act != 3 || (alpha >= 0.f && alpha <= 1.f), "upfirdn2d_epilogue: leaky-ReLU slope must lie in [0, 1]");
    MG_REQUIRE(f_size[0] >= 1 && f_size[1] >= 1 && f_size[0] <= 4 && f_size[1] <= 4, "upfirdn2d_epilogue: filters of at most 4x4 taps");
    MG_REQUIRE(x_size[0] == y_size[0] && x_size[1] == y_size[1] && y_size[2] >= 1 && y_size[3] >= 1, "x and y must agree in batch and channels");
    const int64_t C = x_size[1];
    MG_REQUIRE(C % 8 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)b | (uintptr_t)d) & 15) == 0, "upfirdn2d_epilogue: channels must be a multiple of 8, pointers 16-byte aligned");
    int64_t xn = (int64_t)x_size[0] * C * x_size[2] * x_size[3], yn = (int64_t)y_size[0] * C * y_size[2] * y_size[3];
    const int64_t lim = (1ll << 31) - 64;
    MG_REQUIRE(xn * 2 < lim && yn * 2 < lim && (int64_t)y_size[0] * ((y_size[2] + 31) / 32) * y_size[3] * (C / 8) < (int64_t)UINT32_MAX / 2, "x is too large");
    if (yn == 0) return MG_OK;
    UfParams p;
    p.x = x; p.f = f; p.y = y;
    p.upx = p.upy = p.downx = p.downy = 1; p.padx0 = padx0; p.pady0 = pady0; p.flip = flip ? 1 : 0; p.gain = gain;
    p.n = x_size[0]; p.c = x_size[1]; p.in_h = x_size[2]; p.in_w = x_size[3];
    p.out_h = y_size[2]; p.out_w = y_size[3]; p.f_h = f_size[0]; p.f_w = f_size[1];
    p.xs_c = 1; p.xs_w = C; p.xs_h = C * p.in_w; p.xs_n = C * p.in_w * p.in_h;           // dense channels-last, checked by the caller
    p.ys_c = 1; p.ys_w = C; p.ys_h = C * p.out_w; p.ys_n = C * p.out_w * p.out_h;
    p.fs_h = f_stride[0]; p.fs_w = f_stride[1];
    p.ep_d = d; p.ep_noise = noise; p.ep_strength = strength; p.ep_b = b; p.ep_noise_stride_n = noise_stride_n;
    p.ep_alpha = alpha; p.ep_gain = act_gain; p.ep_clamp = clamp;
    hipStream_t s = (hipStream_t)stream;
    MgLaunchScope scope(MG_K_UPFIRDN2D_NHWC, s, (double)(xn + yn) * 2);
    g_epi_act = act;
    if (dtype == MG_BF16) launch_nhwc_phase<mg_bf16, 1, 1, 0, 0>(p, s); else launch_nhwc_phase<mg_half, 1, 1, 0, 0>(p, s);
    g_epi_act = 0;
    return mg_check_launch(MG_K_UPFIRDN2D_NHWC, "upfirdn2d_epilogue");
}

extern "C" int mg_upfirdn2d_index_table(int32_t* table, int out_size, int in_size, int up, int down, int pad0,
                                        int filter_size, int flip, void* stream)
{
    MG_REQUIRE(table && out_size >= 1, "table must hold out_size*3 ints");
    MG_REQUIRE(up >= 1 && down >= 1 && filter_size >= 1, "up, down and filter_size must be at least 1");
    hipLaunchKernelGGL(upfirdn2d_index_table_kernel, dim3((out_size + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       table, out_size, in_size, up, down, pad0, filter_size, flip ? 1 : 0);
    return mg_check_launch(-1, "upfirdn2d_index_table");
}
