// modconv.hip -- style-modulated 3x3 convolution of the 64-channel, full-resolution layers as ONE kernel on the matrix cores:
//
//     y = clamp( act( conv3x3( x * s[n,ci] ) * d[n,co] + noise[n,h,w] * strength + b[co] ) * gain )
//
// (reference training/networks.py:37-98 `modulated_conv2d` followed by the layer's bias_act, :311-313; with s = d = noise = NULL
// it is the discriminator's conv0 + bias_act, :600-603; with everything NULL and act = linear a plain convolution -- the data
// gradient of those layers is that, on dy and the flipped, transposed weights).  As separate launches: the modulation pass (read x,
// write x*s), the convolution (read, write), the epilogue pass (read, write) -- six trips of the largest activation of the network
// ([N,64,256,256] bf16, 268 MB at batch 32) through HBM.  Here: read x once, write y once.  Bound: HBM (288 FLOP per activation
// byte; at 64 channels the matrix cores are ~1.6x faster than the memory system can feed them: 4.6k MFMA cycles per 256 pixels
// per CU against 7.4k cycles for their 64 KB at ~10 B/clk/CU).
//
// Round 2's version kept the 9x64x64 weights in LDS (83 KB) and therefore ONE workgroup per CU whose phases -- stage the halo,
// multiply, epilogue, store -- ran one after the other: 380 us.  This version is built so that TWO workgroups share a CU and
// cover each other's memory phases:
//   * the WEIGHTS LIVE IN REGISTERS.  v_mfma_f32_32x32x16_bf16 with the weights as A (rows = output channels): a wave owns one
//     half of the output channels (32), i.e. 9 taps x 4 k-steps = 36 A fragments of 4 VGPRs = 144 VGPRs for the whole launch --
//     no LDS for them, no A reads in the loop.  The modulation is folded into them: A = bf16(w * bf16(s[n,ci])), recomputed only
//     when a workgroup's run of tiles crosses into the next sample (the reference's own `fused_modconv` formulation, :91-93).
//   * the input halo (6 x 34 pixels for a 4 x 32 tile) goes global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPRs,
//     out-of-image pixels come back as zeros from the buffer range check), double buffered: the next tile's halo is in flight
//     while this one is multiplied; one barrier per tile.
//   * LDS image: 8 pixels x 128 B per 1-KiB DMA piece, 16-byte slots XOR-swizzled over the 256-B bank row (two pixels) with the
//     pixel-pair index -- applied on the DMA's per-lane SOURCE address and on the ds_read_b128 address alike -- so that the 32
//     pixels of a B fragment spread over the banks.
//   * epilogue on the accumulators (a lane holds 4 consecutive output channels of one pixel per register quad), through a
//     WAVE-PRIVATE 2.5 KB staging strip (no workgroup barrier) into 16-byte stores of 64-byte half pixel rows.
// LDS per workgroup: 2 x 30 KB halo + 10 KB staging + d / bias = 70.5 KB -> two workgroups per CU; <= 256 VGPRs -> 2 waves/SIMD.
// Rounding points: the convolution result to bf16, then d (rounded to bf16), the noise fma, the sum to bf16, bias + act + gain +
// clamp as bias_act -- those of the three-launch path; the modulation is rounded on the weight instead of on the activation.
#include "mg_common.h"
#include "mg_act.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4_t __attribute__((ext_vector_type(4)));

constexpr int MC_C = 64;
constexpr int MC_TH = 4, MC_TW = 32;                // output tile (pixels)
constexpr int MC_HR = MC_TH + 2;                    // halo rows
constexpr int MC_HROW = 40;                         // pixels per halo row in LDS (34 used; 5 DMA pieces of 8 pixels)
constexpr int MC_PIECES = MC_HR * MC_HROW / 8;      // 30 pieces of 1 KiB
constexpr int MC_HALO_BYTES = MC_PIECES * 1024;     // 30,720
constexpr int MC_STAGGER = 1;                       // s_sleep(64) repeats for the workgroup in the odd wave slot (~4k cycles)
constexpr int MC_RING = 5;                          // fragment reads in flight per wave (+1)
constexpr int MC_STAGE_PX = 80;                     // bytes per pixel of the output staging strip (64 + 16: 16-byte aligned, 2-way on the 8-byte writes)
constexpr int MC_STAGE_BYTES = 32 * MC_STAGE_PX;    // one 32-pixel block x 32 channels per wave
constexpr int MC_LDS = 2 * MC_HALO_BYTES + 4 * MC_STAGE_BYTES + 2 * MC_C * 4;      // + d + bias

struct McArgs {
    const mg_bf16* x;        // [N,H,W,64]
    const mg_bf16* w;        // [64][3][3][64]
    const float* s;          // [N,64] modulation (NULL: none)
    const float* d;          // [N,64] demodulation (NULL: none)
    const float* noise;      // [N,H,W] or [H,W] (NULL: none), multiplied by *strength
    const float* strength;   // device scalar (NULL: 1)
    const mg_bf16* b;        // [64] bias (NULL: none)
    mg_bf16* yconv;          // raw convolution output, bf16 (NULL: not wanted)
    mg_bf16* y;              // epilogue output (NULL: not wanted)
    int n, h, w_;
    int64_t noise_stride_n;  // 0 = shared noise
    float alpha, gain, clamp;
    int tiles_x, tiles_y, tiles, per_wg;
    uint32_t x_bytes;        // N*H*W*128: the buffer range of the DMA
    unsigned long long* timing;   // diagnostic build only: cycles per phase of wave 0 of workgroup (0, 0)
};

__device__ __forceinline__ float bf2f(mg_bf16 v) { return to_acc<mg_bf16>(v); }
__device__ __forceinline__ float rbf(float v) { return bf2f(from_acc<mg_bf16>(v)); }

// byte offset of (halo column hx, 16-byte chunk c of the pixel's 64 channels) inside its halo row of the swizzled LDS image: a
// 256-byte bank row holds two pixels = 16 slots; the slot is XORed with the pixel-pair index OF THE COLUMN, so the address of the
// row below is this one plus MC_HROW * 128 -- an immediate offset of the ds_read.
__device__ __forceinline__ uint32_t mc_col_off(uint32_t hx, uint32_t c)
{
    const uint32_t pair = hx >> 1, slot = (((hx & 1u) << 3) | c) ^ (pair & 15u);
    return pair * 256u + slot * 16u;
}

#define LDS_ __attribute__((address_space(3)))
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16(f2_t v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }      // v_cvt_pk_bf16_f32
__device__ __forceinline__ f2_t unpk_bf16(uint32_t u) { f2_t r; r.x = __builtin_bit_cast(float, u << 16); r.y = __builtin_bit_cast(float, u & 0xFFFF0000u); return r; }
__device__ __forceinline__ f2_t rbf2(f2_t v) { return unpk_bf16(pk_bf16(v)); }

template <int ACT, bool MOD, bool DBG>
__global__ void __launch_bounds__(256, 2) modconv3x3_c64_kernel(McArgs p)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const halo = smem;                                   // [2][30 KiB]
    unsigned char* const stage = smem + 2 * MC_HALO_BYTES;              // [4 waves][2,560 B]
    LDS_ float* const dl = (LDS_ float*)(stage + 4 * MC_STAGE_BYTES);   // [64] demodulation of the workgroup's sample
    LDS_ float* const bl = dl + MC_C;                                    // [64] bias

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ob = wave & 1, rh = wave >> 1;                            // this wave: output channels ob*32.., tile rows rh*2, rh*2+1

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

    if (tid < MC_C) bl[tid] = (p.b ? bf2f(p.b[tid]) : 0.f) * p.gain;              // (v + b) * gain = fma(v, gain, b * gain)
    const float strength = (p.noise && p.strength) ? p.strength[0] : 1.f;
    const float clampv = p.clamp >= 0.f ? p.clamp : __builtin_inff();

    // a workgroup stays inside ONE sample (blockIdx.y): its weights are modulated once, before the loop
    const int tiles_per_n = p.tiles_x * p.tiles_y;
    const int n = blockIdx.y;
    const int t_begin = blockIdx.x * p.per_wg;
    const int t_end = t_begin + p.per_wg < tiles_per_n ? t_begin + p.per_wg : tiles_per_n;
    if (t_begin >= t_end) return;

    // ---- halo of tile `t` -> LDS buffer `buf`: this wave's pieces q = wave + 4 * i, i = 0..7, as LDS-DMA (one instruction each)
    // (the per-lane parts of these offsets do not depend on the tile: left alone, the compiler keeps all of them -- and the 72
    // fragment addresses of the multiplication -- in registers across the whole launch, and the 144 weight registers spill.
    // An opaque copy of the lane id makes it recompute them: ~150 VALU instructions per tile in the shadow of 72 MFMAs.)
    auto issue_piece = [&](int t, int buf, int i) {
        const int q = wave + 4 * i;
        if (q >= MC_PIECES) return;                                     // (wave-uniform)
        const int tx = t / p.tiles_y, ty = t - tx * p.tiles_y;       // column-major walk: consecutive tiles share halo rows (L2)
        const int y0 = ty * MC_TH - 1, x0 = tx * MC_TW - 1;
        uint32_t lv = (uint32_t)lane;
        asm volatile("" : "+v"(lv));
        const int hy = q / 5;
        const uint32_t pair = (uint32_t)(q - hy * 5) * 4u + (lv >> 4);      // pixel pair inside the halo row
        const uint32_t l = (lv & 15u) ^ (pair & 15u);
        const int hx = (int)(pair * 2u + (l >> 3));
        const int gy = y0 + hy, gx = x0 + hx;
        const int ok = (hx < MC_TW + 2) & ((uint32_t)gy < (uint32_t)p.h) & ((uint32_t)gx < (uint32_t)p.w_);      // (one select, no branches)
        // out-of-image (and unused) slots: an offset beyond the buffer -> the range check returns zeros
        const uint32_t off = ok ? ((uint32_t)((n * p.h + gy) * p.w_ + gx) * 128u + (l & 7u) * 16u) : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(halo + buf * MC_HALO_BYTES + q * 1024),
                                                 16, (int)off, 0, 0, 0);
    };
    auto issue_halo = [&](int t, int buf) {
        #pragma unroll
        for (int i = 0; i < (MC_PIECES + 3) / 4; i++) issue_piece(t, buf, i);
    };

    // ---- this wave's 36 weight fragments: A[row = co][k = 8*hh + j] = w[ob*32 + r][tap][kc*16 + 8*hh + j] (* s[n][ci], rounded)
    bf16x8 wreg[9][4];
    {
        float sv[4][8];
        if (MOD) {
            #pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                const f4_t a = *(const f4_t*)(p.s + (size_t)n * MC_C + kc * 16 + hh * 8), b4 = *(const f4_t*)(p.s + (size_t)n * MC_C + kc * 16 + hh * 8 + 4);
                sv[kc][0] = rbf(a.x); sv[kc][1] = rbf(a.y); sv[kc][2] = rbf(a.z); sv[kc][3] = rbf(a.w);
                sv[kc][4] = rbf(b4.x); sv[kc][5] = rbf(b4.y); sv[kc][6] = rbf(b4.z); sv[kc][7] = rbf(b4.w);
            }
        }
        #pragma unroll
        for (int tap = 0; tap < 9; tap++)
            #pragma unroll
            for (int kc = 0; kc < 4; kc++) {
                bf16x8 v = *(const bf16x8*)(p.w + ((size_t)(ob * 32 + r) * 9 + tap) * MC_C + kc * 16 + hh * 8);
                if (MOD) {
                    #pragma unroll
                    for (int j = 0; j < 8; j++) v[j] = (__bf16)((float)v[j] * sv[kc][j]);
                }
                wreg[tap][kc] = v;
            }
    }
    issue_halo(t_begin, 0);
    // per-sample constants of the epilogue: the workgroup stays inside sample n
    if (tid < MC_C) dl[tid] = p.d ? rbf(p.d[(size_t)n * MC_C + tid]) : 1.f;     // dcoefs.to(x.dtype)
    float nz_next[2] = {0.f, 0.f};
    if (p.noise && p.y) {
        const int tx0 = t_begin / p.tiles_y, ty0 = t_begin - tx0 * p.tiles_y;
        #pragma unroll
        for (int pb = 0; pb < 2; pb++)
            nz_next[pb] = p.noise[(size_t)n * p.noise_stride_n + (size_t)(ty0 * MC_TH + rh * 2 + pb) * p.w_ + tx0 * MC_TW + r];
    }

    // Two workgroups share a CU and run the same program: left alone they march in step -- both in their multiplication (each
    // wave gets every other slot of its SIMD's matrix pipe: 75 cycles per MFMA measured), then both in their epilogue (pipe idle).
    // The workgroup in the odd wave slot therefore starts half a tile late, once; after that its epilogues fall into the other's
    // multiplications.  (Which workgroups share a CU is the dispatcher's business: the wave slot id is read from the hardware.)
    if (MC_STAGGER) {
        const uint32_t hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11));     // HW_REG_HW_ID, bits 3:0 = wave slot on the SIMD
        if (hw_id & 1u)
            for (int i = 0; i < MC_STAGGER; i++) __builtin_amdgcn_s_sleep(64);                  // 64 x 64 cycles each
    }

    unsigned long long tm[5] = {0, 0, 0, 0, 0}, tc = 0;
#define MC_MARK(i) do { if (DBG) { unsigned long long now = __builtin_readcyclecounter(); tm[i] += now - tc; tc = now; } } while (0)
    if (DBG) tc = __builtin_readcyclecounter();
    for (int t = t_begin; t < t_end; t++) {
        const int buf = (t - t_begin) & 1;
        const int tx = t / p.tiles_y, ty = t - tx * p.tiles_y;       // column-major walk: consecutive tiles share halo rows (L2)
        const int oy0 = ty * MC_TH, ox0 = tx * MC_TW;

        MC_MARK(0);
        __syncthreads();                // halo(t) has landed (vmcnt(0) + barrier); every wave is done with buffer buf^1
        MC_MARK(1);
        // the noise samples of the NEXT tile go into flight now (its halo follows, spread over the first MFMAs)
        float nzv[2] = {nz_next[0], nz_next[1]};
        if (t + 1 < t_end) {
            if (p.noise && p.y) {
                const int tx1 = (t + 1) / p.tiles_y, ty1 = (t + 1) - tx1 * p.tiles_y;
                #pragma unroll
                for (int pb = 0; pb < 2; pb++)
                    nz_next[pb] = p.noise[(size_t)n * p.noise_stride_n + (size_t)(ty1 * MC_TH + rh * 2 + pb) * p.w_ + tx1 * MC_TW + r];
            }
        }
        MC_MARK(2);

        // ---- per pixel row of the wave: 9 taps x 4 k-steps into ONE accumulator tile (16 registers), then its epilogue
        const LDS_ unsigned char* hb = (const LDS_ unsigned char*)(halo + buf * MC_HALO_BYTES);
        unsigned char* const st = stage + wave * MC_STAGE_BYTES;
        #pragma unroll
        for (int pb = 0; pb < 2; pb++) {
            uint32_t lv = (uint32_t)lane;
            asm volatile("" : "+v"(lv));                                // see issue_piece
            const uint32_t rv = lv & 31u, hv = lv >> 5;
            // the lane's three column addresses (kw = 0, 1, 2) at k-step 0; k-step kc is an XOR of bits 5..6, the row an immediate
            uint32_t col[3];
            #pragma unroll
            for (int kw = 0; kw < 3; kw++) col[kw] = mc_col_off(rv + (uint32_t)kw, hv);
            f32x16 acc;
            #pragma unroll
            for (int e = 0; e < 16; e++) acc[e] = 0.f;
            // a ring of MC_RING fragments: the read of k-step i + MC_RING - 1 is issued before the MFMA of k-step i, i.e. four MFMAs
            // (128 cycles) ahead of its use -- with 144 VGPRs of weights there is no room for more, and LDS latency under load is above 100
            bf16x8 frag[MC_RING];
            auto read_step = [&](int i, bf16x8& f) {                    // i = tap * 4 + kc
                const int tap = i >> 2, kh = tap / 3, kw = tap - kh * 3;
                const uint32_t a = col[kw] ^ (uint32_t)((i & 3) << 5);  // (an XOR below bit 8 stays inside the 256-byte bank row)
                f = *(const LDS_ bf16x8*)(hb + a + (rh * 2 + pb + kh) * (MC_HROW * 128));
            };
            #pragma unroll
            for (int i = 0; i < MC_RING - 1; i++) read_step(i, frag[i]);
            #pragma unroll
            for (int i = 0; i < 36; i++) {
                if (i + MC_RING - 1 < 36) read_step(i + MC_RING - 1, frag[(i + MC_RING - 1) % MC_RING]);
                // the next tile's halo: one DMA piece per 2 MFMAs of the first pixel row, issued in their shadow
                if (pb == 0 && (i & 1) == 0 && (i >> 1) < (MC_PIECES + 3) / 4 && t + 1 < t_end) issue_piece(t + 1, buf ^ 1, i >> 1);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[i >> 2][i & 3], frag[i % MC_RING], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (pb == 1) MC_MARK(3);

            // ---- outputs.  acc[reg]: output channel ob*32 + (reg&3) + 8*(reg>>2) + 4*hh of pixel (row rh*2 + pb, column r).
            // Two values per instruction (v_pk_mul/fma_f32, v_cvt_pk_bf16_f32): the epilogue is VALU time the other workgroup's
            // MFMAs have to share the SIMD with.
            #pragma unroll
            for (int which = 0; which < 2; which++) {
                mg_bf16* out = which == 0 ? p.yconv : p.y;
                if (!out) continue;
                // (no run-time switches in here: absent operands are neutral values -- d = 1, noise = 0, clamp = +inf -- because inside
                // the unrolled code hipcc turns a uniform `if` into computing both sides and a v_cndmask per value)
                const f2_t nzs2 = {nzv[pb] * 1.f, nzv[pb] * 1.f}, st2 = {strength, strength}, g2 = {p.gain, p.gain}, al2 = {p.alpha, p.alpha};
                #pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int co = ob * 32 + 8 * g + 4 * hh;
                    f4_t dv = {1.f, 1.f, 1.f, 1.f}, bv = {0.f, 0.f, 0.f, 0.f};
                    if (which == 1) { dv = *(const LDS_ f4_t*)(dl + co); bv = *(const LDS_ f4_t*)(bl + co); }
                    uint2 q;
                    #pragma unroll
                    for (int j = 0; j < 2; j++) {
                        f2_t v = {acc[g * 4 + 2 * j], acc[g * 4 + 2 * j + 1]};
                        uint32_t packed = pk_bf16(v);                               // the convolution's own bf16 result
                        if (which == 1) {
                            v = unpk_bf16(packed);
                            const f2_t d2 = {dv[2 * j], dv[2 * j + 1]}, b2 = {bv[2 * j], bv[2 * j + 1]};
                            v = v * d2;
                            v = __builtin_elementwise_fma(nzs2, st2, v);
                            v = rbf2(v);                                            // the fma output is a bf16 tensor in the three-launch path
                            v = __builtin_elementwise_fma(v, g2, b2);               // (v + b) * gain
                            if (ACT == 3) {                                         // 0 <= alpha <= 1 (checked by the host): lrelu(t) = max(t, alpha * t)
                                const f2_t m = v * al2;
                                v.x = __builtin_fmaxf(v.x, m.x); v.y = __builtin_fmaxf(v.y, m.y);
                            }
                            // v_med3_f32: a NaN comes out as min3 = -clamp, as bias_act's comparison chain leaves it
                            v.x = __builtin_amdgcn_fmed3f(v.x, -clampv, clampv);
                            v.y = __builtin_amdgcn_fmed3f(v.y, -clampv, clampv);
                            packed = pk_bf16(v);
                        }
                        if (j == 0) q.x = packed; else q.y = packed;
                    }
                    *(uint2*)(st + r * MC_STAGE_PX + g * 16 + hh * 8) = q;
                }
                // the wave's own strip: 32 pixels x 64 B, back as 16-byte pieces (lane -> pixel lane/4 (+16), piece lane%4)
                #pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int px = (lane >> 2) + half * 16, piece = lane & 3;
                    const uint4 v = *(const uint4*)(st + px * MC_STAGE_PX + piece * 16);
                    *(uint4*)((unsigned char*)out + ((size_t)((n * p.h + oy0 + rh * 2 + pb) * p.w_ + ox0 + px) * MC_C + ob * 32) * 2 + piece * 16) = v;
                }
            }
        }
        MC_MARK(4);
    }
    if (DBG && p.timing && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
        for (int i = 0; i < 5; i++) p.timing[i] = tm[i];
#undef MC_MARK
}

} // namespace

static unsigned long long* g_mc_timing = nullptr;
extern "C" void mg_modconv_debug_timing(void* buf) { g_mc_timing = (unsigned long long*)buf; }

extern "C" int mg_modconv3x3_c64(const void* x, const void* w, const float* s, const float* d, const float* noise, const float* strength,
                                 const void* b, void* yconv, void* y, void* xs, int32_t n, int32_t h, int32_t wd, int64_t noise_stride_n,
                                 int act, float alpha, float gain, float clamp, void* stream)
{
    MG_REQUIRE(x && w && (y || yconv), "modconv3x3_c64: null pointer");
    MG_REQUIRE(!xs, "modconv3x3_c64: the modulated input is no longer an output (the modulation is folded into the weights)");
    MG_REQUIRE(n >= 0 && h >= MC_TH && wd >= MC_TW && h % MC_TH == 0 && wd % MC_TW == 0,
               "modconv3x3_c64: height must be a multiple of %d and width of %d", MC_TH, MC_TW);
    MG_REQUIRE(act == 1 || act == 3, "modconv3x3_c64: activation must be linear (1) or lrelu (3)");
    MG_REQUIRE(act == 1 || (alpha >= 0.f && alpha <= 1.f), "modconv3x3_c64: leaky ReLU slope must lie in [0, 1]");
    MG_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)yconv | (uintptr_t)s) & 15) == 0, "modconv3x3_c64: pointers must be 16-byte aligned");
    MG_REQUIRE((int64_t)n * h * wd * MC_C * 2 < (int64_t)0xFFFFFFF0u, "modconv3x3_c64: x is too large");
    if (n == 0) return MG_OK;
    McArgs p;
    p.x = (const mg_bf16*)x; p.w = (const mg_bf16*)w; p.s = s; p.d = d; p.noise = noise; p.strength = strength; p.b = (const mg_bf16*)b;
    p.yconv = (mg_bf16*)yconv; p.y = (mg_bf16*)y; p.n = n; p.h = h; p.w_ = wd; p.noise_stride_n = noise_stride_n;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    p.tiles_x = wd / MC_TW; p.tiles_y = h / MC_TH; p.tiles = n * p.tiles_x * p.tiles_y;
    p.x_bytes = (uint32_t)((int64_t)n * h * wd * MC_C * 2);
    // two workgroups per CU (512 in all), each a contiguous run of tiles of one sample
    const int tiles_per_n = p.tiles_x * p.tiles_y;
    int per_n = (512 + n - 1) / n;
    if (per_n > tiles_per_n) per_n = tiles_per_n;
    p.per_wg = (tiles_per_n + per_n - 1) / per_n;
    per_n = (tiles_per_n + p.per_wg - 1) / p.per_wg;
    const dim3 grid(per_n, n);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)modconv3x3_c64_kernel<1, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MC_LDS);
        (void)hipFuncSetAttribute((const void*)modconv3x3_c64_kernel<3, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MC_LDS);
        (void)hipFuncSetAttribute((const void*)modconv3x3_c64_kernel<1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MC_LDS);
        (void)hipFuncSetAttribute((const void*)modconv3x3_c64_kernel<3, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MC_LDS);
        (void)hipFuncSetAttribute((const void*)modconv3x3_c64_kernel<3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MC_LDS);
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    const double px = (double)n * h * wd * MC_C * 2;
    MgLaunchScope scope(MG_K_MODCONV, st, px * (1 + (y ? 1 : 0) + (yconv ? 1 : 0)) + 9.0 * MC_C * MC_C * 2);
    p.timing = g_mc_timing;
    if (s && act == 3 && g_mc_timing) {     // diagnostic instantiation (tools/bench_modconv.py): cycle stamps around the phases
        hipLaunchKernelGGL((modconv3x3_c64_kernel<3, true, true>), grid, dim3(256), MC_LDS, st, p);
    } else if (s) {
        if (act == 3) hipLaunchKernelGGL((modconv3x3_c64_kernel<3, true, false>), grid, dim3(256), MC_LDS, st, p);
        else          hipLaunchKernelGGL((modconv3x3_c64_kernel<1, true, false>), grid, dim3(256), MC_LDS, st, p);
    } else {
        if (act == 3) hipLaunchKernelGGL((modconv3x3_c64_kernel<3, false, false>), grid, dim3(256), MC_LDS, st, p);
        else          hipLaunchKernelGGL((modconv3x3_c64_kernel<1, false, false>), grid, dim3(256), MC_LDS, st, p);
    }
    return mg_check_launch(MG_K_MODCONV, "modconv3x3_c64");
}
