/*
 * montage_hip.h -- C ABI of libmontage_hip.so, the MI355X (gfx950) native code
 * behind the MontageGAN / StyleGAN2 `torch_utils.ops` plugin boundary.
 *
 * Every entry point takes plain device pointers, sizes, strides and a HIP
 * stream handle (passed as void*; NULL = the null stream).  No torch types
 * cross this boundary.  All functions return 0 on success and a negative
 * MG_E_* code on failure; mg_last_error() then holds a message whose text
 * mirrors the TORCH_CHECK string of the reference entry point it replaces.
 * Nothing here allocates, frees, copies to the host or synchronises: every
 * call enqueues kernels on `stream` and returns (graph-capture safe).
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   mg_upfirdn2d            <- upfirdn2d_plugin.upfirdn2d
 *                              montage_gan/torch_utils/ops/upfirdn2d.cpp:16-94,
 *                              kernels upfirdn2d.cu:29-200, params upfirdn2d.h:14-40
 *   mg_bias_act             <- bias_act_plugin.bias_act
 *                              montage_gan/torch_utils/ops/bias_act.cpp:32-90,
 *                              kernel bias_act.cu:23-147, params bias_act.h:12-32
 *   mg_fma_bias_act*        <- fma.fma followed by bias_act.bias_act as called by
 *                              SynthesisLayer (training/networks.py:86-88,311-313),
 *                              fused into one HBM round trip
 *   mg_alpha_composite_*    <- custom_utils/image_utils.py:112-163
 *                              (alpha_composite_pytorch, default branch)
 *   mg_translate_sample_*   <- fukuwarai/networks.py:246-255 (STNv2c:
 *                              affine_grid + grid_sample on a pure translation,
 *                              with the +1/-1 workaround folded in as `shift`)
 *   mg_filtered_lrelu       <- named by the north star; absent upstream;
 *                              defined as upfirdn2d(up) -> bias_act(lrelu) -> upfirdn2d(down)
 */
#ifndef MONTAGE_HIP_H_
#define MONTAGE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 1
#define MG_GS_SLOTS 64          /* length of the `gs` accumulator of mg_bwd_fused / mg_bwd_fused_nhwc */

/* element types of activation tensors */
enum { MG_F32 = 0, MG_F16 = 1, MG_BF16 = 2, MG_F64 = 3 };

/* error codes */
enum {
    MG_OK = 0,
    MG_E_ARG = -1,     /* argument check failed (message in mg_last_error) */
    MG_E_LAUNCH = -2,  /* hipLaunchKernel / hipGetLastError failed */
    MG_E_NODEV = -3    /* no gfx950 device / runtime not initialised */
};

/* kernel ids for mg_launch_count / mg_prof_* */
enum {
    MG_K_UPFIRDN2D_GENERIC = 0,
    MG_K_UPFIRDN2D_TILED = 1,
    MG_K_BIAS_ACT = 2,
    MG_K_FMA_BIAS_ACT = 3,
    MG_K_ALPHA_COMPOSITE_FWD = 4,
    MG_K_ALPHA_COMPOSITE_BWD = 5,
    MG_K_TRANSLATE_FWD = 6,
    MG_K_TRANSLATE_BWD_INPUT = 7,
    MG_K_TRANSLATE_BWD_SHIFT = 8,
    MG_K_FILTERED_LRELU = 9,
    MG_K_BWD_FUSED = 10,
    MG_K_UPFIRDN2D_STREAM = 11,
    MG_K_UPFIRDN2D_NHWC = 12,
    MG_K_AFFINE_WARP_FWD = 13,
    MG_K_AFFINE_WARP_BWD = 14,
    MG_K_CHANNEL_MIX = 15,
    MG_K_DEMOD = 16,
    /* 17: retired */
    MG_K_STATS = 18,
    MG_K_ADA_PARAMS = 19,
    MG_K_MODCONV = 20,
    MG_K_UPFIRDN2D_SEP = 21,
    MG_K_MBSTD = 22,
    MG_K_TORGB = 23,
    MG_K_WGRAD = 24,
    MG_K_FROMRGB = 25,
    MG_K_RESIDUAL = 26,
    MG_K_COUNT = 27
};

int         mg_abi_version(void);
const char* mg_last_error(void);               /* thread-local, never NULL */
uint64_t    mg_launch_count(int kernel_id);    /* kernels enqueued since load */

/* ---- per-kernel timing (HIP events on the launch stream) -------------------
 * mg_prof_enable(1) makes every launch of the listed kernels record a
 * hipEvent pair on its own stream; mg_prof_collect() synchronises those events
 * and returns, per kernel id, the launch count, the summed device time (ms) and
 * the summed algorithmic bytes ((numel_in + numel_out) * sizeof(T)).
 * Used only by bench.py and the profiling tests; off by default.            */
int mg_prof_enable(int on);
int mg_prof_collect(int kernel_id, uint64_t* launches, double* ms, double* bytes);
int mg_prof_reset(void);

/* ---- upfirdn2d -------------------------------------------------------------
 * y[n,c,oy,ox] = gain * sum_{ky,kx} xup_pad[n,c, oy*downy+ky, ox*downx+kx] * ft[ky,kx]
 * ft = f flipped in both axes unless `flip` (upfirdn2d.cu:66-67,125-126).
 * Sizes/strides are in ELEMENTS, order {N, C, H, W}; any strides (NCHW, NHWC,
 * sliced views) are accepted.  The caller computes
 *   out_w = (in_w*upx + padx0 + padx1 - f_w + downx) / downx   (upfirdn2d.cpp:32)
 * and passes it in y_size; only pad*0 enters the kernel, as upstream.
 * f is always float32 [f_h, f_w] on the same device.  Pads may be negative.   */
int mg_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                 const int32_t x_size[4], const int64_t x_stride[4],
                 const int32_t f_size[2], const int64_t f_stride[2],
                 const int32_t y_size[4], const int64_t y_stride[4],
                 int upx, int upy, int downx, int downy,
                 int padx0, int pady0, int flip, float gain, void* stream);

/* The 1-D 12-tap filter by a factor of 2 along BOTH axes in one pass (upfirdn2d.py:253-257 runs such a filter as two
 * mg_upfirdn2d calls): `up` != 0: up = 2, down = 1; else up = 1, down = 2.  x [planes, in_h, in_w] -> y [planes, out_h,
 * out_w], float32, dense; f = 12 float32 taps on the device; pad*0 / flip as above; `gain_axis` multiplies after each
 * axis (sqrt(gain) of the 2-D call).  planes <= 65535.                                                            */
int mg_upfirdn2d_sep12(const float* x, const float* f, float* y, int64_t planes,
                       int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                       int32_t up, int32_t padx0, int32_t pady0, int32_t flip, float gain_axis, void* stream);

/* upsample2d(pad(x, [mx0, mx1, my0, my1], mode='reflect'), f, up=2) for a 12-tap f (training/augment.py:315-317) without
 * materialising the padded image: x [planes, src_h, src_w] -> y [planes, 2 (src_h + my0 + my1), 2 (src_w + mx0 + mx1)],
 * float32 dense, margins in [0, size - 1].                                                                          */
int mg_reflect_pad_upsample12(const float* x, const float* f, float* y, int64_t planes, int32_t src_h, int32_t src_w,
                              int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, float gain_axis, void* stream);
/* ... with `window` = device float[4] (mx0, my0, mx1, my1 as computed on the device, each <= the padding given here): output tiles outside the
 * up-sampled region the consumer needs are left unwritten, so the padding can be a host-known bound (HIP-graph recording) at the cost of the
 * needed region.  NULL: as above. */
int mg_reflect_pad_upsample12_window(const float* x, const float* f, float* y, int64_t planes, int32_t src_h, int32_t src_w,
                                     int32_t mx0, int32_t mx1, int32_t my0, int32_t my1, float gain_axis, const float* window, void* stream);

/* Integer index path only (the part that must be bit exact): for each output
 * coordinate o in [0, out_size) writes {in0, count, filter0} as computed on the
 * DEVICE by the same code the kernels use (upfirdn2d.cu:43-48, 60-65).
 * `table` is device memory, int32[out_size*3].                                */
int mg_upfirdn2d_index_table(int32_t* table, int out_size, int in_size,
                             int up, int down, int pad0, int filter_size,
                             int flip, void* stream);

/* ---- bias_act --------------------------------------------------------------
 * grad 0: y  = clamp(act(x + b) * gain)
 * grad 1: dx = dy(x) * act'(.) * gain, zero where |yref| >= clamp
 * grad 2: second-order term (bias_act.cu:54-142).
 * b/xref/yref/dy may be NULL.  size_x elements, dense; b index is
 * (i / step_b) % size_b (bias_act.cu:44).  act = 1..9 (bias_act.py:26-42),
 * clamp < 0 disables clamping.                                                */
int mg_bias_act(const void* x, const void* b, const void* xref, const void* yref,
                const void* dy, void* y, int dtype, int grad, int act,
                float alpha, float gain, float clamp,
                int64_t size_x, int32_t size_b, int64_t step_b, void* stream);

/* ---- fused modulated-conv epilogue ----------------------------------------
 * y = clamp(act(x * d[n,c] + noise[n,h,w] * noise_gain + b[c]) * gain), NCHW
 * contiguous; d (float32 [N,C]), noise (float32, sample n at noise + n*noise_stride_n:
 * stride 0 = one [H*W] map shared by the batch, H*W = per-sample maps) and b may each be NULL;
 * the noise multiplier is noise_gain * (noise_gain_dev ? *noise_gain_dev : 1) -- a DEVICE scalar, so the
 * learned noise_strength never has to be read back by the host.  d is rounded to the tensor dtype inside the
 * kernels (the reference multiplies by dcoefs.to(x.dtype)), so callers hand over the float32 values as they are.
 * One read and one write of x instead of the three round trips of
 * fma.fma + bias_act.bias_act (training/networks.py:86-88, 311-313).
 * The backward pass is mg_bwd_fused (below): dx and all three reductions in one pass.    */
int mg_fma_bias_act(const void* x, const float* d, const float* noise, const void* b,
                    void* y, int dtype, int act, float alpha, float gain, float clamp,
                    float noise_gain, const float* noise_gain_dev, int64_t noise_stride_n,
                    int32_t n, int32_t c, int32_t hw, void* stream);


/* ---- fused backward with reductions -----------------------------------------
 * One pass for what autograd otherwise does in up to five: NCHW contiguous, planes (n,c) of hw elements.
 *   dpre = yref ? dy * act'(yref) * gain masked by the clamp (bias_act order 1) : dy
 *   dx   = d ? dpre * d[n,c] : dpre                      (written)
 *   gd[n,c] += sum_hw dpre * x        (x, gd non NULL)   -- d(dcoefs) of the epilogue, d(styles) of the modulation
 *   gb[c]   += sum_{n,hw} dpre        (gb non NULL)      -- d(bias)
 *   gs[k]   += partial sums of dpre * noise[n,hw] (noise, gs non NULL) -- d(noise_strength) = sum of the
 *              MG_GS_SLOTS entries of gs (spread over slots so that the atomics do not pile on one address)
 * gd/gb/gs are float32 accumulators the caller zero-fills (float atomics, one per workgroup and sum).
 * Covers: bias_act backward + bias gradient (bias_act.py:186-196), the fma/bias_act epilogue backward
 * (fma.py:31-47) and the backward of x * styles[N,C,1,1] (training/networks.py:70).                      */
int mg_bwd_fused(const void* dy, const void* yref, const void* x, const float* d, const float* noise,
                 void* dx, float* gd, float* gb, float* gs, int dtype, int act, float alpha, float gain,
                 float clamp, int64_t noise_stride_n, int32_t n, int32_t c, int32_t hw, void* stream);

/* ---- channels-last (NHWC) epilogue and fused backward ----------------------------
 * Same arithmetic as mg_fma_bias_act / mg_bwd_fused on a tensor whose MEMORY is [N, HW, C] (torch.channels_last,
 * which the reference's bf16/fp16 blocks may run in: training/networks.py `fp16_channels_last`, bias_act.py:157
 * keeps the layout).  c must be a multiple of the 16-byte vector (8 for 16-bit types, 4 for float32); for the
 * backward additionally (c / vector) must divide 256.  `gb_partial` is [N*C]: per-sample partial bias sums
 * (the caller adds them over N); gd is [N*C], gs is [MG_GS_SLOTS]; all float32, zero-filled by the caller.              */
int mg_fma_bias_act_nhwc(const void* x, const float* d, const float* noise, const void* b, void* y, int dtype,
                         int act, float alpha, float gain, float clamp, float noise_gain, const float* noise_gain_dev,
                         int64_t noise_stride_n, int32_t n, int32_t c, int32_t hw, void* stream);
int mg_bwd_fused_nhwc(const void* dy, const void* yref, const void* x, const float* d, const float* noise,
                      void* dx, float* gd, float* gb_partial, float* gs, int dtype, int act, float alpha, float gain,
                      float clamp, int64_t noise_stride_n, int32_t n, int32_t c, int32_t hw, void* stream);
/* The same with the epilogue's input x REBUILT from its saved output (recon != 0, x = NULL, act linear or lrelu, gain != 0):
 *     pre = yref / gain  (/ alpha where negative),   x = (pre - noise * *strength - bias[c]) / T(d[n,c])
 * -- what gd sums over -- so that a producer that fuses the epilogue into its own store (the matrix-core convolution, the
 * blur of the up-sampling layers) need not write its pre-epilogue result at all.  bias [C] in the activation dtype or NULL,
 * strength a device scalar or NULL (1).  recon = 0 is mg_bwd_fused_nhwc. */
int mg_bwd_fused_nhwc_y(const void* dy, const void* yref, const void* x, const float* d, const float* noise,
                        const float* strength, const void* bias, int recon, void* dx, float* gd, float* gb_partial, float* gs,
                        int dtype, int act, float alpha, float gain, float clamp, int64_t noise_stride_n, int32_t n, int32_t c,
                        int32_t hw, void* stream);

/* ---- demodulation coefficients ---------------------------------------------------
 * d[n,o] = rsqrt(sum_i s[n,i]^2 * wsq[o,i] + 1e-8), wsq[o,i] = sum_k w[o,i,k]^2 -- the demodulation of
 * modulated_conv2d (training/networks.py:76-77) for the shared-weight path, float32.
 * w dense: [O,I,kk] (channels_last = 0) or [O,kk,I] (channels_last = 1, torch.channels_last of [O,I,kh,kw]);
 * s [N,I], d [N,O], wsq [O,I] (written by fwd, read by bwd).
 * mg_demod_bwd: g = d(loss)/d(d) [N,O]; writes ds [N,I] and / or dw in w's layout (either may be NULL).       */
int mg_demod_fwd(const float* w, const float* s, float* d, float* wsq, int32_t n, int32_t oc, int32_t ic, int32_t kk,
                 int32_t channels_last, void* stream);
int mg_demod_bwd(const float* w, const float* s, const float* d, const float* wsq, const float* g, float* ds, float* dw,
                 int32_t n, int32_t oc, int32_t ic, int32_t kk, int32_t channels_last, void* stream);

/* ---- ADA: affine resampling and colour matrix ------------------------------------
 * mg_affine_warp_fwd:  y = grid_sample(x, affine_grid(theta, [N,C,out_h,out_w], align_corners=False),
 *                          mode='bilinear', padding_mode='zeros', align_corners=False)
 *   (reference training/augment.py:376-377 -> torch_utils/ops/grid_sample_gradfix.py:32-41); the grid is never
 *   materialised.  x [N,C,in_h,in_w], y [N,C,out_h,out_w] contiguous, theta [N,2,3] float32, C <= 4.
 * mg_affine_warp_bwd:  dx = adjoint of the above applied to dy (grid_sample_gradfix.py:62-70 without the unused
 *   grid gradient), evaluated as a gather per input pixel -- deterministic, no atomics.
 * mg_channel_mix:      y[n,i,p] = sum_j A[n,i,j] * x[n,j,p] + t[n,i]   (t may be NULL), C <= 4, x/y [N,C,hw]:
 *   the colour transform of training/augment.py:398-411.                                                        */
int mg_affine_warp_fwd(const void* x, const float* theta, void* y, int dtype, int32_t n, int32_t c,
                       int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w, void* stream);
int mg_affine_warp_bwd(const void* dy, const float* theta, void* dx, int dtype, int32_t n, int32_t c,
                       int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w, void* stream);
int mg_channel_mix(const void* x, const float* A, const float* t, void* y, int dtype, int32_t n, int32_t c,
                   int64_t hw, void* stream);

/*
 * mg_stats_accumulate: counter[0..2] += [n, sum(x), sum(x^2)] for a dense tensor of n elements (float32 / float16 / bfloat16),
 * float64 atomics -- what training_stats.report() does with six aten kernels per reported value
 * (reference montage_gan/torch_utils/training_stats.py:58-101).
 */
int mg_stats_accumulate(const void* x, int dtype, int64_t n, double* counter, void* stream);

/*
 * mg_ada_params: the per-sample transform parameters of AugmentPipe (reference montage_gan/training/augment.py:236-330 geometric
 * stage, :380-420 colour stage) from pre-drawn random tables U [n,nu] (uniform) and Z [n,nz] (normal) and the device scalar p:
 *   Ginv [n,3,3]   inverse geometric matrix (NULL: geometric stage off);  margin [4] reflect-padding reach x0,y0,x1,y1 (float, not yet ceil'ed)
 *   theta [n,2,3]  (left @ Ginv @ right)[:2] when cfg carries the two host-known factors (has_lr), else NULL
 *   Cmat [n,4,4]   colour matrix (NULL: colour stage off);  A4 [n,4,4], t4 [n,4]  its RGBA form for mg_channel_mix (optional)
 * cfg: 44 floats -- 21 probabilities / strengths in constructor order of the enabled transforms, width, height, hz_pad, channels,
 * has_lr, left[9], right[9] (see torch_utils/ops/ada_params.py).  One launch instead of ~200 aten launches per call.
 */
int mg_ada_params(const float* U, const float* Z, const float* p, int32_t n, int32_t nu, int32_t nz, const float* cfg, int32_t ncfg,
                  float* Ginv, float* margin, float* theta, float* Cmat, float* A4, float* t4, void* stream);

/*
 * mg_modconv3x3_c64: y = clamp(act(conv3x3(x * s[n,ci]) * d[n,co] + noise * strength + b[co]) * gain), 64 -> 64 channels, stride 1,
 * padding 1, bf16 NHWC, on the matrix cores (v_mfma_f32_32x32x16_bf16) -- modulated_conv2d's training path plus the layer's
 * bias_act in one launch (reference montage_gan/training/networks.py:37-98,311-313; with s = d = noise = NULL the
 * discriminator's conv0 + bias_act, :600-603).  x, y, yconv [N,H,W,64] bf16; w [64][3][3][64] bf16 (channels_last of [O,I,3,3]);
 * s, d [N,64] float32 (rounded to bf16 inside, as the reference's .to(x.dtype)); noise [N,H,W] or [H,W] float32 with
 * noise_stride_n = H*W or 0, multiplied by the device scalar *strength; b [64] bf16.  yconv (the raw convolution output) and xs
 * (the modulated input x * s, what the weight gradient of the convolution reads) are optional extra outputs for the backward pass;
 * y is optional too.  H % 8 == 0, W % 32 == 0.  act = 1 (linear) or 3 (lrelu).
 */
int mg_modconv3x3_c64(const void* x, const void* w, const float* s, const float* d, const float* noise, const float* strength,
                      const void* b, void* yconv, void* y, void* xs, int32_t n, int32_t h, int32_t wd, int64_t noise_stride_n,
                      int act, float alpha, float gain, float clamp, void* stream);
/*
 * mg_wgrad3x3_c64: weight gradient of the same layers on the matrix cores (transposed LDS reads, ds_read_b64_tr_b16), with the
 * modulation and its gradient folded in (reference training/networks.py:62-88):
 *     H_n[co][tap][ci] = sum_p dy[n,p,co] * x[n,p+tap,ci];   dw[co][tap][ci] += sum_n bf16(s[n,ci]) * H_n;   ds[n][ci] += sum_{co,tap} w * H_n
 * dy, x [N,H,W,64] bf16 channels-last; s [N,64] float32 or NULL; w [64][3][3][64] bf16 or NULL (needed for ds); dw [64][3][3][64] and
 * ds [N,64] (or NULL) float32, ZERO-FILLED by the caller (atomic adds).  H % 4 == 0, W % 32 == 0.
 */
int mg_wgrad3x3_c64(const void* dy, const void* x, const float* s, const void* w, float* dw, float* ds,
                    int32_t n, int32_t h, int32_t wd, void* stream);
/*
 * mg_fromrgb_nhwc_{fwd,bwd}: the discriminator's FromRGB layer in one pass each way (reference training/networks.py:589-594
 * `y = self.fromrgb(img)`: Conv2dLayer :138-186 = conv2d_resample 1x1, I = img_channels <= 4 -> C in {64,128,256,512}, then
 * bias_act(lrelu | linear, gain, clamp)).  x [pixels, I], y / dy [pixels, C] channels-last T = bf16 | fp16; w [C, I] T; b [C] T or NULL.
 * fwd: y = T(bias_act(T(conv), b)).  bwd: dpre = T(dy * act'(y) * gain where |y| < clamp) is never stored; dx [pixels, I] (or NULL) =
 * T(sum_c w[c][i] dpre[c]); gw32 [C, I] and gb32 [C] (or NULL) are float32 accumulators ZERO-FILLED by the caller; gw / gb receive them
 * cast to T by a second tiny launch (what aten's convolution_backward / bias_act's bias gradient return).  act = 1 | 3.
 */
int mg_fromrgb_nhwc_fwd(const void* x, const void* w, const void* b, void* y, int dtype, int act, float alpha, float gain, float clamp,
                        int64_t pixels, int32_t c, int32_t i, void* stream);
int mg_fromrgb_nhwc_bwd(const void* x, const void* w, const void* y, const void* dy, void* dx, float* gw32, float* gb32, void* gw, void* gb,
                        int dtype, int act, float alpha, float gain, float clamp, int64_t pixels, int32_t c, int32_t i, void* stream);
/*
 * mg_residual_bias_act_{fwd,bwd}: the tail of the discriminator's residual block (reference training/networks.py:596-600:
 * `y = self.skip(x, gain)`, `x = self.conv1(x, gain)`, `x = y.add_(x)` -- i.e. bias_act(y_raw, None, linear, r_gain), bias_act(x_raw, b, act, gain,
 * clamp) and the add, after the two convolutions) in one pass each way.  x, r, out, dout, dx, dr [pixels, C] channels-last T = bf16 | fp16,
 * C in {64,128,256,512}; b [C] T or NULL.  fwd: out = T(T(bias_act(x, b)) + T(r * r_gain)).  bwd: the activation's output is recomputed from x and b;
 * dx = T(dout * act'(.) * gain where unclamped), dr = T(dout * r_gain) (NULL: not wanted), gb32 [C] float32 ZERO-FILLED by the caller += dx
 * (NULL: not wanted).  act = 1 | 3.
 */
int mg_residual_bias_act_fwd(const void* x, const void* b, const void* r, void* out, int dtype, int act, float alpha, float gain, float clamp,
                             float r_gain, int64_t pixels, int32_t c, void* stream);
int mg_residual_bias_act_bwd(const void* dout, const void* x, const void* b, void* dx, void* dr, float* gb32, int dtype, int act, float alpha,
                             float gain, float clamp, float r_gain, int64_t pixels, int32_t c, void* stream);
/* diagnostics: while `buf` (device memory, uint64[8]) is set, workgroup 0 of mg_modconv3x3_c64 leaves its cycle counts per phase
 * (stage, prefetch issue, MFMA loop, epilogue, store) there; NULL switches it off (tools/bench_modconv.py). */
void mg_modconv_debug_timing(void* buf);

/*
 * mg_upfirdn2d_epilogue: the 4x4 blur that ends an up-sampling SynthesisLayer (reference training/networks.py:299-313 through
 * torch_utils/ops/conv2d_resample.py:123-144: upfirdn2d(x, f, padding, gain) on the transposed convolution's output) and the layer's
 * epilogue -- fma(x, dcoefs, noise * strength), bias_act(b, act, alpha, act_gain, clamp) -- in one pass: up = down = 1, filter <= 4x4,
 * float16 / bfloat16 dense channels-last x [N,C,inH,inW] -> y [N,C,outH,outW] (out = in + pad0 + pad1 - f + 1; the caller sizes y),
 * d [N,C] float32 or NULL, noise [N,outH,outW] / [outH,outW] float32 (noise_stride_n = outH*outW / 0) or NULL, strength device scalar
 * or NULL (1), b [C] in the activation dtype or NULL.  Every intermediate is rounded where the separate launches round it.
 */
int mg_upfirdn2d_epilogue(const void* x, const float* f, void* y, int dtype, const int32_t x_size[4], const int32_t f_size[2],
                          const int64_t f_stride[2], const int32_t y_size[4], int padx0, int pady0, int flip, float gain,
                          const float* d, const float* noise, const float* strength, const void* b, int64_t noise_stride_n,
                          int act, float alpha, float act_gain, float clamp, void* stream);

/*
 * mg_torgb_nhwc_{fwd,bwd}: the generator's ToRGB layer on channels-last float16 / bfloat16 activations, one pass each way
 * (reference montage_gan/training/networks.py:343-359 ToRGBLayer.forward -- modulated_conv2d(demodulate=False), training path
 * :62-88, then bias_act(clamp) -- followed by SynthesisBlock's `y.to(float32, contiguous)`, :460-464):
 *     y[n,o,p] = float( T( clamp( T( sum_c T(x[n,p,c] * T(s[n,c])) * w[o,c] ) + b[o] ) ) )          T = the activation dtype
 * x [N,HW,C] (NHWC), w [O,C], s [N,C] float32, b [O] or NULL, y [N,O,HW] float32 (NCHW).  C in {64,128,256,512}, O <= 4.
 * clamp < 0: none.  Backward: gy [N,O,HW] float32 -> gx [N,HW,C]; with dyc = T(gy) (the linear activation saves no output, so the
 * reference's bias_act backward masks nothing even under a clamp: bias_act.py:26, bias_act.cu:139-141) the ZERO-FILLED float32
 * accumulators G [N,O,C] and gb [O] (may be NULL) receive sum_p dyc[n,p,o] * x[n,p,c] and sum_{n,p} dyc[n,p,o];
 * mg_torgb_bwd_finish turns them into the parameter gradients in one small launch: gw[o,c] = T(sum_n T(s[n,c]) G[n,o,c]),
 * gs[n,c] = sum_o w[o,c] G[n,o,c] (float32), gb[o] = T(gb32[o]).
 */
int mg_torgb_nhwc_fwd(const void* x, const void* w, const float* s, const void* b, float* y, int dtype,
                      int32_t n, int32_t c, int32_t hw, int32_t o, float clamp, void* stream);
int mg_torgb_nhwc_bwd(const void* x, const void* w, const float* s, const float* gy, void* gx, float* G, float* gb,
                      int dtype, int32_t n, int32_t c, int32_t hw, int32_t o, void* stream);
int mg_torgb_bwd_finish(const float* G, const float* gb32, const float* s, const void* w, void* gw, float* gs, void* gb,
                        int dtype, int32_t n, int32_t c, int32_t o, void* stream);

/* ---- alpha composite -------------------------------------------------------
 * x: [B, L, 4, H, W] contiguous (RGB + alpha, values nominally in [0,1]);
 * y: [B, 4, H, W].  canvas = layer 0; for l = 1..L-1: canvas = layer_l OVER canvas
 *   a_o = a1 + a2*(1-a1);  c_o = nan_to_num((c1*a1 + c2*a2*(1-a1)) / a_o)
 * (image_utils.py:129-134,143-147).  nan_to_num: NaN->0, +/-inf->+/-FLT_MAX.
 * Backward follows the autograd chain of those ops literally, including the
 * NaN that torch produces where a_o == 0 (0 * inf), unless safe_grad != 0, in
 * which case those pixels receive a zero gradient.  L <= 16.                  */
int mg_alpha_composite_fwd(const void* x, void* y, int dtype,
                           int32_t b, int32_t l, int32_t h, int32_t w, void* stream);
int mg_alpha_composite_bwd(const void* x, const void* dy, void* dx, int dtype,
                           int32_t b, int32_t l, int32_t h, int32_t w,
                           int safe_grad, void* stream);

/* ---- translate sample (STN) -------------------------------------------------
 * x: [M, C, H, W] contiguous, t: float32 [M, 2] = (dx, dy) in normalised
 * [-1,1] units.  y = grid_sample(x + shift, affine_grid([[1,0,dx],[0,1,dy]]),
 * bilinear, zeros, align_corners=False) - shift   (fukuwarai/networks.py:246-255;
 * shift = 1 for STNv2c, 0 for STNv2b).  bwd_input: dx (a gather, no atomics);
 * bwd_shift: dt[M,2] float32, MUST be zero-filled by the caller (atomic adds). */
int mg_translate_sample_fwd(const void* x, const float* t, void* y, int dtype,
                            int32_t m, int32_t c, int32_t h, int32_t w,
                            float shift, void* stream);
int mg_translate_sample_bwd_input(const void* dy, const float* t, void* dx, int dtype,
                                  int32_t m, int32_t c, int32_t h, int32_t w,
                                  void* stream);
int mg_translate_sample_bwd_shift(const void* x, const void* dy, const float* t,
                                  float* dt, int dtype,
                                  int32_t m, int32_t c, int32_t h, int32_t w,
                                  float shift, void* stream);

/* ---- filtered_lrelu (additive) -----------------------------------
 * y = upfirdn2d(clamp(lrelu(upfirdn2d(x + b[c], fu, up, pad) , alpha) * gain), fd, down)
 * NCHW contiguous, separable-or-not filters given as dense float32 2-D arrays.
 * The intermediate lives in LDS only.  up, down in {1,2}.                     */
int mg_filtered_lrelu(const void* x, const void* b, const float* fu, const float* fd,
                      void* y, int dtype,
                      int32_t n, int32_t c, int32_t in_h, int32_t in_w,
                      int32_t out_h, int32_t out_w,
                      int32_t fu_h, int32_t fu_w, int32_t fd_h, int32_t fd_w,
                      int up, int down, int padx0, int pady0,
                      float alpha, float gain, float clamp, int flip, void* stream);

/* ---- minibatch standard deviation (training/networks.py:667-691) ---------------------------------------------------------
 * x [N, C, HW] float32 dense -> out [N, C + F, HW] = cat([x, y]): y[s, f] = mean over the c = C / F channels of block f and the
 * HW pixels of sqrt(var over the `group` samples {m, m + N/group, ...} of sample s's group + 1e-8).  mg_mbstd_bwd: the gradient
 * w.r.t. x from dout [N, C + F, HW] and x (mean and sigma are recomputed).  group <= 16, N % group == 0, C % F == 0.        */
int mg_mbstd_fwd(const float* x, float* out, int32_t n, int32_t c, int32_t hw, int32_t group, int32_t f, void* stream);
int mg_mbstd_bwd(const float* dout, const float* x, float* dx, int32_t n, int32_t c, int32_t hw, int32_t group, int32_t f, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MONTAGE_HIP_H_ */
