"""StyleGAN2 generator / discriminator modules that sit directly on the plugin ops.

Same classes, constructor arguments, attribute and parameter names (hence ``state_dict``
keys) as the reference's ``training/networks.py`` and ``custom/networks_aio.py`` -- the two
files differ only by ``conv_config_index`` (the log2 of the coarsest block resolution,
2 upstream, 3 in the aio configuration; custom/networks_aio.py:588-594,898-900) and by
``GlobalMappingNetwork`` -- so one implementation serves both and
``custom/networks_aio.py`` re-exports it.

What is MI355X-specific here (everything else is the callers' arithmetic, unchanged):
  * ``modulated_conv2d`` computes the demodulation coefficients as a [N,I] x [I,O] GEMM
    (``rsqrt((s^2) @ (sum_k w^2)^T + eps)``) instead of materialising the [N,O,I,k,k]
    per-sample weights just to reduce them (reference training/networks.py:62-68);
  * ``SynthesisLayer`` finishes with the fused ``modconv_epilogue`` (one HBM round trip instead
    of the three of fma + bias_act);
  * reduced precision is a constructor choice (``fp16_dtype``): torch.float16 as upstream, or
    torch.bfloat16 (BASELINE configs; no overflow pre-normalisation needed in bf16).
"""
import os
import threading

import numpy as np
import torch

from ..torch_utils import misc
from ..torch_utils.ops import bias_act
from ..torch_utils.ops import conv2d_resample
from ..torch_utils.ops import demod
from ..torch_utils.ops import fma
from ..torch_utils.ops import mbstd
from ..torch_utils.ops import torgb
from ..torch_utils.ops import modconv_epilogue
from ..torch_utils.ops import modconv3x3
from ..torch_utils.ops import fromrgb
from ..torch_utils.ops import residual_bias_act
from ..torch_utils.ops import upfirdn2d
from . import weight_prep

# ---- all affine layers of a synthesis network in a few batched launches ---------------------------------------------------------
# Every SynthesisLayer / ToRGBLayer starts with ``styles = self.affine(w)``: twenty [N,512] x [512,C] products per generator pass,
# ~10 us each and three more launches each in the backward pass (two products and a bias sum) -- about 2.4 k tiny GEMM launches
# per training step.  SynthesisNetwork.forward computes them up front instead, one ``baddbmm`` per group of layers with the same
# width (four groups at 256^2: 512 / 256 / 128 / 64 channels), and hands each layer its row through ``_affine``; autograd's
# backward of the group is two batched products and one sum.  Same float32 arithmetic per element (a different GEMM kernel: the
# sums may differ in the last bit).  CPU tensors and layers called outside their network compute ``self.affine(w)`` as before.
batched_styles = os.environ.get('MG_BATCHED_STYLES', '1') != '0'
# Both tables live for the duration of ONE SynthesisNetwork.forward on ONE thread (thread-local, nested passes restore the outer
# tables): a layer takes its entry out when it runs; a layer invoked again in the same pass, or outside a network, finds nothing
# and computes ``affine(w)`` / draws its own noise exactly as the reference does.
_tables = threading.local()
# Likewise the per-layer ``torch.randn([N, 1, H, W])`` of noise_mode='random' (13 launches per pass at 256^2): ONE draw for the
# whole pass, each layer takes its contiguous share.  (Other positions of the generator's sequence than with per-layer draws;
# both are i.i.d. normal, but a GPU run from a given seed does NOT reproduce the per-layer sequence of the reference -- "parity
# unpinned" for noise_mode='random' on the GPU; the CPU path keeps the per-layer draws and reproduces the reference's numbers.)


def _affine(layer, w, gain=None):
    """styles of `layer` for this pass: its row of the batched product (already multiplied by `gain`), or affine(w) [* gain]."""
    table = getattr(_tables, 'styles', None)
    if table is not None:
        s = table.pop(id(layer), None)
        if s is not None:
            return s
    s = layer.affine(w)
    return s if gain is None else s * gain

# Module switch: use the fused epilogue op inside SynthesisLayer (GPU tensors only).
fused_epilogue = True
# Module switch: ToRGB + the cast to the float32 image path as one native pass (channels-last reduced-precision blocks on the GPU).
fused_torgb = os.environ.get('MG_FUSED_TORGB', '1') != '0'


def _default(v, d):
    return d if v is None else v


@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


def _demod_coefs(weight, styles):
    """rsqrt(sum_{i,kh,kw} (w[o,i,kh,kw] * s[n,i])^2 + 1e-8) -> [N,O]: one native kernel on the GPU, one small GEMM otherwise."""
    return demod.demod_coefs(weight, styles)


@misc.profiled_function
def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True):
    """Style-modulated convolution (reference training/networks.py:37-98).

    x [N,I,H,W]; weight [O,I,kh,kw]; styles [N,I]; noise broadcastable to the output or None.
    fused_modconv=True folds styles and demodulation into per-sample weights and runs one grouped
    convolution (inference); False scales the activations before and after a shared-weight convolution
    (training)."""
    n = int(x.shape[0])
    oc, ic, kh, kw = weight.shape
    misc.assert_shape(x, [n, ic, None, None])
    misc.assert_shape(styles, [n, ic])

    # float16 only: keep the products inside the fp16 range (:58-61).  bf16/fp32 have the range.
    if x.dtype == torch.float16 and demodulate:
        weight = weight * (1 / np.sqrt(ic * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)

    dcoefs = _demod_coefs(weight, styles) if demodulate else None

    if not fused_modconv:
        # modulation: native one-pass scale on the GPU (its backward is one pass as well), aten otherwise
        x = modconv_epilogue.channel_scale(x, styles)
        x = conv2d_resample.conv2d_resample(x=x, w=weight_prep.get(weight, 1, x.dtype), f=resample_filter, up=up, down=down,
                                            padding=padding, flip_weight=flip_weight)
        if demodulate and noise is not None:
            x = fma.fma(x, dcoefs.to(x.dtype).reshape(n, -1, 1, 1), noise.to(x.dtype))
        elif demodulate:
            x = x * dcoefs.to(x.dtype).reshape(n, -1, 1, 1)
        elif noise is not None:
            x = x.add_(noise.to(x.dtype))
        return x

    w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)            # [N,O,I,kh,kw]
    if demodulate:
        w = w * dcoefs.reshape(n, -1, 1, 1, 1)
    x = x.reshape(1, -1, *x.shape[2:])
    x = conv2d_resample.conv2d_resample(x=x, w=w.reshape(-1, ic, kh, kw).to(x.dtype), f=resample_filter, up=up,
                                        down=down, padding=padding, groups=n, flip_weight=flip_weight)
    x = x.reshape(n, -1, *x.shape[2:])
    if noise is not None:
        x = x.add_(noise)
    return x


class FullyConnectedLayer(torch.nn.Module):
    """y = act(x @ (W * lr_mul/sqrt(in)).T + b * lr_mul)  (reference :104-133)."""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1.0, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def prep_entries(self, dtype=torch.float32):
        ents = [(self.weight, self.weight_gain, dtype)]
        if self.bias is not None and self.bias_gain != 1:
            ents.append((self.bias, self.bias_gain, dtype))
        return ents

    def forward(self, x):
        if x.dtype == self.weight.dtype:
            w = weight_prep.get(self.weight, self.weight_gain, x.dtype)
        else:
            w = self.weight.to(x.dtype) * self.weight_gain
        b = self.bias
        if b is not None:
            if self.bias_gain != 1 and x.dtype == b.dtype:
                b = weight_prep.get(b, self.bias_gain, x.dtype)
            else:
                b = b.to(x.dtype)
                if self.bias_gain != 1:
                    b = b * self.bias_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)


class Conv2dLayer(torch.nn.Module):
    """conv2d_resample + bias_act (reference :138-186)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=None, conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.activation = activation
        self.up, self.down = up, down
        self.conv_clamp = conv_clamp
        self.out_channels_ = out_channels
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(_default(resample_filter, [1, 3, 3, 1])))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt)
        bias = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(bias) if bias is not None else None
        else:
            self.register_buffer('weight', weight)
            if bias is not None:
                self.register_buffer('bias', bias)
            else:
                self.bias = None

    def prep_entries(self, dtype):
        ents = [(self.weight, self.weight_gain, dtype)]
        if self.bias is not None and dtype != torch.float32 and isinstance(self.bias, torch.nn.Parameter):
            ents.append((self.bias, 1, dtype))
        return ents

    def conv_only(self, x):
        """The layer without its bias_act: (raw convolution output, bias in x.dtype or None) -- for callers that fuse the element-wise
        tail with what follows (DiscriminatorBlock's residual add)."""
        w = weight_prep.get(self.weight, self.weight_gain, x.dtype)
        b = weight_prep.get(self.bias, 1, x.dtype) if self.bias is not None else None
        y = conv2d_resample.conv2d_resample(x=x, w=w, f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=(self.up == 1))
        return y, b

    def forward(self, x, gain=1):
        w = weight_prep.get(self.weight, self.weight_gain, x.dtype)           # (weight * weight_gain).to(x.dtype)
        b = weight_prep.get(self.bias, 1, x.dtype) if self.bias is not None else None        # bias.to(x.dtype)
        if (self.up == 1 and self.down == 1 and self.padding == 0 and self.activation in ('linear', 'lrelu') and fromrgb.applicable(x, w)):
            # the discriminator's FromRGB (<= 4 -> 64..512 channels, 1x1): convolution + bias_act in ONE pass over the wide tensor
            return fromrgb.fromrgb(x, w, b, act=self.activation, gain=self.act_gain * gain,
                                   clamp=self.conv_clamp * gain if self.conv_clamp is not None else None)
        if (self.activation in ('linear', 'lrelu') and modconv3x3.applicable(x, w, up=self.up, down=self.down)
                and w.is_contiguous(memory_format=torch.channels_last)):
            # 64 -> 64 channels at full resolution (the discriminator's first conv0): convolution + bias_act in ONE kernel
            return modconv3x3.fused_modconv3x3(x, w, b=b, act=self.activation, gain=self.act_gain * gain,
                                               clamp=self.conv_clamp * gain if self.conv_clamp is not None else None)
        x = conv2d_resample.conv2d_resample(x=x, w=w, f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=(self.up == 1))
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class MappingNetwork(torch.nn.Module):
    """z (and c) -> w, broadcast to num_ws, with a running average for truncation (reference :191-267).
    ``num_splits`` > 1 gives the aio GlobalMappingNetwork layout (custom/networks_aio.py:264-352)."""

    def __init__(self, z_dim, w_dim, num_ws, c_dim=0, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995, _num_splits=1):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws = z_dim, c_dim, w_dim, num_ws
        self.num_layers, self.w_avg_beta = num_layers, w_avg_beta
        embed_features = 0 if c_dim == 0 else _default(embed_features, w_dim)
        layer_features = _default(layer_features, w_dim)
        sizes = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim * _num_splits]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(sizes[idx], sizes[idx + 1], activation=activation,
                                                          lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim * _num_splits]))
        self._prep = None

    def prep_entries(self):
        layers = ([self.embed] if self.c_dim > 0 else []) + [getattr(self, f'fc{idx}') for idx in range(self.num_layers)]
        return [e for layer in layers for e in layer.prep_entries()]

    def forward(self, z, c=None, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False, w_avg_rows=None):
        """w_avg_rows: only the first that many rows feed the running average (the loss maps z and its style-mixing partner
        in ONE call of 2N rows -- the rows are independent -- and only z's rows belong to the statistic)."""
        if self._prep is None:
            self._prep = weight_prep.Plan(self.prep_entries())
        with weight_prep.scope(self._prep):
            x = None
            if self.z_dim > 0:
                misc.assert_shape(z, [None, self.z_dim])
                x = normalize_2nd_moment(z.to(torch.float32))
            if self.c_dim > 0:
                misc.assert_shape(c, [None, self.c_dim])
                y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
                x = torch.cat([x, y], dim=1) if x is not None else y
            for idx in range(self.num_layers):
                x = getattr(self, f'fc{idx}')(x)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:
            xs = x.detach() if w_avg_rows is None else x.detach()[:w_avg_rows]
            self.w_avg.copy_(xs.mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            assert self.w_avg_beta is not None
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class GlobalMappingNetwork(MappingNetwork):
    """One mapping network whose last layer emits num_splits latents at once; output [B, num_splits,
    num_ws, w_dim] -- one w stack per image layer (custom/networks_aio.py:264-352)."""

    def __init__(self, z_dim, w_dim, num_ws, c_dim=0, num_layers=8, num_splits=9, **kwargs):
        super().__init__(z_dim=z_dim, w_dim=w_dim, num_ws=num_ws, c_dim=c_dim, num_layers=num_layers,
                         _num_splits=num_splits, **kwargs)
        self.num_splits = num_splits

    def forward(self, z, c=None, **kwargs):
        x = super().forward(z, c, **kwargs)                                   # [B, num_ws, splits*w_dim]
        return x.reshape(x.shape[0], self.num_ws, self.num_splits, self.w_dim).transpose(1, 2)


def _spatial(resolution, init_res, conv_config_index):
    """Feature-map size of a block: resolution scaled by init_res / 2**conv_config_index per axis."""
    return [resolution * init_res[0] // 2 ** conv_config_index, resolution * init_res[1] // 2 ** conv_config_index]


class SynthesisLayer(torch.nn.Module):
    """affine(w) -> modulated 3x3 conv (optionally x2 up) -> noise -> bias + lrelu (reference :272-338)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, init_res=None, kernel_size=3, up=1, use_noise=True,
                 activation='lrelu', resample_filter=None, conv_clamp=None, channels_last=False, conv_config_index=2):
        super().__init__()
        self.resolution = resolution
        self.init_res = _default(init_res, [4, 4])
        self.conv_config_index = conv_config_index
        self.up, self.use_noise, self.activation, self.conv_clamp = up, use_noise, activation, conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(_default(resample_filter, [1, 3, 3, 1])))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        if use_noise:
            self.register_buffer('noise_const', torch.randn(_spatial(resolution, self.init_res, conv_config_index)))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1):
        assert noise_mode in ['random', 'const', 'none']
        misc.assert_shape(x, [None, self.weight.shape[1], *_spatial(self.resolution // self.up, self.init_res, self.conv_config_index)])
        styles = _affine(self, w)
        noise_raw = None            # unscaled noise; multiplied by noise_strength below (or inside the fused kernel)
        if self.use_noise and noise_mode == 'random':
            table = getattr(_tables, 'noise', None)
            noise_raw = table.pop(id(self), None) if table is not None else None
            if noise_raw is None or noise_raw.shape[0] != x.shape[0]:
                noise_raw = torch.randn([x.shape[0], 1, *_spatial(self.resolution, self.init_res, self.conv_config_index)],
                                        device=x.device)
        if self.use_noise and noise_mode == 'const':
            noise_raw = self.noise_const
        act_gain = self.act_gain * gain
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        flip_weight = (self.up == 1)

        if fused_epilogue and not fused_modconv and x.device.type == 'cuda' and x.dtype != torch.float64:
            # native path: modulate (1 pass) -> shared-weight convolution -> ONE pass for
            # demodulate * x + noise * strength + bias -> lrelu -> gain -> clamp
            n = int(x.shape[0])
            weight, st = self.weight, styles
            if x.dtype == torch.float16:
                weight = weight * (1 / np.sqrt(np.prod(weight.shape[1:])) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
                st = st / st.norm(float('inf'), dim=1, keepdim=True)
            dcoefs = _demod_coefs(weight, st)
            wc = weight_prep.get(weight, 1, x.dtype) if weight is self.weight else weight.to(x.dtype)
            if modconv3x3.applicable(x, wc, up=self.up) and wc.is_contiguous(memory_format=torch.channels_last):
                # the 64-channel full-resolution layer: modulation, convolution and epilogue in ONE matrix-core kernel
                return modconv3x3.fused_modconv3x3(x, wc, st, dcoefs, noise_raw, self.noise_strength if noise_raw is not None else None,
                                                   self.bias, act=self.activation, gain=act_gain, clamp=act_clamp)
            y = modconv_epilogue.channel_scale(x, st)
            if self.up > 1 and self.activation in ('linear', 'lrelu') and modconv_epilogue.blur_fusion and modconv_epilogue._is_nhwc(y):
                # up-sampling layer on channels-last 16-bit tensors: the blur that ends conv2d_resample and the epilogue in ONE pass
                strength = self.noise_strength if noise_raw is not None else None
                return conv2d_resample.conv2d_resample(
                    x=y, w=wc, f=self.resample_filter, up=self.up, padding=self.padding, flip_weight=flip_weight,
                    tail=lambda t, f, pad, fgain, flip: modconv_epilogue.blur_epilogue(
                        t, f, pad, flip_filter=flip, fgain=fgain, dcoefs=dcoefs, noise=noise_raw, b=self.bias, act=self.activation,
                        gain=act_gain, clamp=act_clamp, noise_strength=strength))
            y = conv2d_resample.conv2d_resample(x=y, w=wc, f=self.resample_filter, up=self.up,
                                                padding=self.padding, flip_weight=flip_weight)
            if y.is_contiguous() or modconv_epilogue._is_nhwc(y):
                return modconv_epilogue.modconv_epilogue(y, dcoefs, noise_raw, self.bias, act=self.activation, gain=act_gain,
                                                         clamp=act_clamp, noise_strength=self.noise_strength if noise_raw is not None else None)
            noise = noise_raw * self.noise_strength if noise_raw is not None else None
            y = fma.fma(y, dcoefs.to(y.dtype).reshape(n, -1, 1, 1), noise.to(y.dtype)) if noise is not None else \
                y * dcoefs.to(y.dtype).reshape(n, -1, 1, 1)
            return bias_act.bias_act(y, self.bias.to(y.dtype), act=self.activation, gain=act_gain, clamp=act_clamp)

        noise = noise_raw * self.noise_strength if noise_raw is not None else None
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=self.resample_filter, flip_weight=flip_weight, fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=act_gain, clamp=act_clamp)


class ToRGBLayer(torch.nn.Module):
    """Modulated 1x1 conv without demodulation, linear, clamped (reference :343-359)."""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=True):
        styles = _affine(self, w, self.weight_gain)
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        return bias_act.bias_act(x, weight_prep.get(self.bias, 1, x.dtype), clamp=self.conv_clamp)

    def forward_image(self, x, w, fused_modconv=True):
        """``forward`` followed by the block's cast to the float32 NCHW image path (reference :460-464) -- on the training path of
        the channels-last reduced-precision blocks one native pass over x (ops/torgb.py) instead of four launches."""
        if not fused_modconv and fused_torgb and x.is_cuda and x.dtype != torch.float32:
            styles = _affine(self, w, self.weight_gain)
            return torgb.torgb(x, weight_prep.get(self.weight, 1, x.dtype), styles, weight_prep.get(self.bias, 1, x.dtype), clamp=self.conv_clamp)
        return self.forward(x, w, fused_modconv=fused_modconv).to(dtype=torch.float32, memory_format=torch.contiguous_format)


class SynthesisBlock(torch.nn.Module):
    """One resolution of the generator: (up-conv) conv, toRGB with skip-summed image (reference :364-466)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, init_res=None,
                 architecture='skip', resample_filter=None, conv_clamp=None, use_fp16=False, fp16_channels_last=False,
                 conv_config_index=2, fp16_dtype=torch.float16, **layer_kwargs):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        resample_filter = _default(resample_filter, [1, 3, 3, 1])
        self.in_channels, self.w_dim, self.resolution = in_channels, w_dim, resolution
        self.init_res = _default(init_res, [4, 4])
        self.conv_config_index = conv_config_index
        self.img_channels, self.is_last, self.architecture = img_channels, is_last, architecture
        self.use_fp16, self.fp16_dtype = use_fp16, fp16_dtype
        self.channels_last = (use_fp16 and fp16_channels_last)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = self.num_torgb = 0
        common = dict(w_dim=w_dim, resolution=resolution, init_res=self.init_res, conv_clamp=conv_clamp,
                      channels_last=self.channels_last, conv_config_index=conv_config_index, **layer_kwargs)
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, *_spatial(resolution, self.init_res, conv_config_index)]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, up=2, resample_filter=resample_filter, **common)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, **common)
        self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last)
            self.num_torgb += 1
        if in_channels != 0 and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2, resample_filter=resample_filter,
                                    channels_last=self.channels_last)

    def prep_entries(self, force_fp32=False):
        dtype = self.fp16_dtype if self.use_fp16 and not force_fp32 else torch.float32
        ents = []
        for name in ('conv0', 'conv1', 'torgb'):
            layer = getattr(self, name, None)
            if layer is not None:
                ents += layer.affine.prep_entries()
                if dtype not in (torch.float32, torch.float16):         # fp16 renormalises the weight first (:58-61)
                    ents.append((layer.weight, 1, dtype))
                if dtype != torch.float32 and name == 'torgb':
                    ents.append((layer.bias, 1, dtype))
        if hasattr(self, 'skip'):
            ents += self.skip.prep_entries(dtype)
        return ents

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, **layer_kwargs):
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        w_iter = iter(ws.unbind(dim=1))
        dtype = self.fp16_dtype if self.use_fp16 and not force_fp32 else torch.float32
        fmt = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        if fused_modconv is None:
            fused_modconv = (not self.training) and (dtype == torch.float32 or int(ws.shape[0]) == 1)
        prev = _spatial(self.resolution // 2, self.init_res, self.conv_config_index)

        if self.in_channels == 0:
            x = self.const.to(dtype=dtype, memory_format=fmt).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
        else:
            misc.assert_shape(x, [None, self.in_channels, *prev])
            x = x.to(dtype=dtype, memory_format=fmt)

        if self.in_channels == 0:
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        elif self.architecture == 'resnet':
            y = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, gain=np.sqrt(0.5), **layer_kwargs)
            x = y.add_(x)
        else:
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)

        if img is not None:
            misc.assert_shape(img, [None, self.img_channels, *prev])
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        if self.is_last or self.architecture == 'skip':
            y = self.torgb.forward_image(x, next(w_iter), fused_modconv=fused_modconv)       # image path stays fp32 (:460-464)
            img = img.add_(y) if img is not None else y
        assert x.dtype == dtype and (img is None or img.dtype == torch.float32)
        return x, img


class SynthesisNetwork(torch.nn.Module):
    """Stack of SynthesisBlocks from 2**conv_config_index to img_resolution (reference :471-521)."""

    def __init__(self, w_dim, img_resolution, img_channels, init_res=None, channel_base=32768, channel_max=512,
                 num_fp16_res=0, conv_config_index=2, **block_kwargs):
        super().__init__()
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        self.w_dim, self.img_resolution, self.img_channels = w_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.init_res = _default(init_res, [4, 4])
        self.conv_config_index = conv_config_index
        self.block_resolutions = [2 ** i for i in range(conv_config_index, self.img_resolution_log2 + 1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        self.num_ws = 0
        for res in self.block_resolutions:
            first = (res == 2 ** conv_config_index)
            block = SynthesisBlock(0 if first else channels[res // 2], channels[res], w_dim=w_dim, resolution=res,
                                   init_res=self.init_res, img_channels=img_channels, is_last=(res == img_resolution),
                                   use_fp16=(res >= fp16_resolution), conv_config_index=conv_config_index, **block_kwargs)
            self.num_ws += block.num_conv + (block.num_torgb if res == img_resolution else 0)
            setattr(self, f'b{res}', block)
        self._prep = {}          # force_fp32 -> weight_prep entries (static per network)
        self._style_plan = {}    # device -> groups of layers whose affine products are batched

    def forward(self, ws, **block_kwargs):
        misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
        ws = ws.to(torch.float32)
        x = img = None
        w_idx = 0
        force_fp32 = bool(block_kwargs.get('force_fp32', False))
        entries = self._prep.get(force_fp32)
        if entries is None:
            entries = self._prep[force_fp32] = weight_prep.Plan([e for res in self.block_resolutions for e in getattr(self, f'b{res}').prep_entries(force_fp32)])
        with weight_prep.scope(entries):
            outer, outer_nz = getattr(_tables, 'styles', None), getattr(_tables, 'noise', None)
            _tables.styles = self._all_styles(ws) if (batched_styles and ws.is_cuda) else None
            _tables.noise = self._all_noise(ws) if (batched_styles and ws.is_cuda and block_kwargs.get('noise_mode', 'random') == 'random') else None
            try:
                for res in self.block_resolutions:
                    block = getattr(self, f'b{res}')
                    # a block's toRGB shares the w of the next block's first conv
                    x, img = block(x, img, ws.narrow(1, w_idx, block.num_conv + block.num_torgb), **block_kwargs)
                    w_idx += block.num_conv
            finally:
                _tables.styles, _tables.noise = outer, outer_nz
        return img

    def _all_noise(self, ws):
        """One normal draw for every noisy layer of the pass: {id(layer): [N, 1, H, W] contiguous share}."""
        n = int(ws.shape[0])
        layers = [getattr(getattr(self, f'b{res}'), name) for res in self.block_resolutions for name in ('conv0', 'conv1')
                  if hasattr(getattr(self, f'b{res}'), name) and getattr(getattr(self, f'b{res}'), name).use_noise]
        shapes = [_spatial(l.resolution, l.init_res, l.conv_config_index) for l in layers]
        buf = torch.randn([sum(n * h * w for h, w in shapes)], device=ws.device)
        table, off = {}, 0
        for l, (h, w) in zip(layers, shapes):
            table[id(l)] = buf[off:off + n * h * w].view(n, 1, h, w)
            off += n * h * w
        return table

    def _style_groups(self, device):
        """[(layers, index tensor into ws)] per affine width, in network order (static per network and device)."""
        key = str(device)
        groups = self._style_plan.get(key)
        if groups is None:
            by_width, w_idx = {}, 0
            for res in self.block_resolutions:
                block = getattr(self, f'b{res}')
                names = [n for n in ('conv0', 'conv1') if hasattr(block, n)] + (['torgb'] if hasattr(block, 'torgb') else [])
                for j, name in enumerate(names):         # the order SynthesisBlock.forward draws its ws in
                    layer = getattr(block, name)
                    if layer.affine.bias is not None and layer.affine.bias_gain == 1 and layer.affine.activation == 'linear':
                        # ToRGB multiplies its styles by weight_gain (:357): folded into the group's product below
                        by_width.setdefault(layer.affine.weight.shape[0], []).append((layer, w_idx + j, float(getattr(layer, 'weight_gain', 1.0)) if name == 'torgb' else 1.0))
                w_idx += block.num_conv
            groups = self._style_plan[key] = [
                ([l for l, _, _ in v], torch.tensor([i for _, i, _ in v], device=device),
                 None if all(g == 1.0 for _, _, g in v) else torch.tensor([g for _, _, g in v], dtype=torch.float32, device=device).reshape(-1, 1, 1))
                for v in by_width.values() if len(v) > 1]
        return groups

    def _all_styles(self, ws):
        table = {}
        for layers, idx, gains in self._style_groups(ws.device):
            weight = torch.stack([weight_prep.get(l.affine.weight, l.affine.weight_gain, torch.float32) for l in layers])    # [L, C, w_dim]
            bias = torch.stack([l.affine.bias for l in layers]).unsqueeze(1)                                               # [L, 1, C]
            x = ws.index_select(1, idx).transpose(0, 1)                                                                    # [L, N, w_dim]
            st_all = torch.baddbmm(bias, x, weight.transpose(1, 2))
            if gains is not None:
                st_all = st_all * gains
            for layer, st in zip(layers, st_all.unbind(0)):
                table[id(layer)] = st
        return table


class Generator(torch.nn.Module):
    """mapping + synthesis (reference :526-564)."""

    def __init__(self, z_dim, w_dim, img_resolution, img_channels, c_dim=0, init_res=None, mapping_kwargs=None,
                 synthesis_kwargs=None):
        super().__init__()
        init_res = _default(init_res, [4, 4])
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution = self.res = img_resolution
        self.init_res, self.img_channels = init_res, img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, init_res=init_res,
                                          img_channels=img_channels, **(synthesis_kwargs or {}))
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **(mapping_kwargs or {}))
        cci = self.synthesis.conv_config_index
        self.output_shape = [1, img_channels, *_spatial(img_resolution, init_res, cci)]

    def forward(self, z, c=None, truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, **synthesis_kwargs)


class DiscriminatorBlock(torch.nn.Module):
    """fromRGB (first block / skip arch), 3x3 conv, blur + stride-2 3x3 conv, residual 1x1 skip (reference :569-662)."""

    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx, init_res=None,
                 architecture='resnet', activation='lrelu', resample_filter=None, conv_clamp=None, use_fp16=False,
                 fp16_channels_last=False, freeze_layers=0, conv_config_index=2, fp16_dtype=torch.float16):
        assert in_channels in [0, tmp_channels]
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        resample_filter = _default(resample_filter, [1, 3, 3, 1])
        self.in_channels, self.resolution, self.img_channels = in_channels, resolution, img_channels
        self.init_res = _default(init_res, [4, 4])
        self.conv_config_index = conv_config_index
        self.first_layer_idx, self.architecture = first_layer_idx, architecture
        self.use_fp16, self.fp16_dtype = use_fp16, fp16_dtype
        self.channels_last = (use_fp16 and fp16_channels_last)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_layers = 0

        def next_trainable():          # Freeze-D: layers are numbered across the whole discriminator
            idx = self.first_layer_idx + self.num_layers
            self.num_layers += 1
            return idx >= freeze_layers

        if in_channels == 0 or architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation,
                                       trainable=next_trainable(), conv_clamp=conv_clamp, channels_last=self.channels_last)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, activation=activation, trainable=next_trainable(),
                                 conv_clamp=conv_clamp, channels_last=self.channels_last)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=2,
                                 trainable=next_trainable(), resample_filter=resample_filter, conv_clamp=conv_clamp,
                                 channels_last=self.channels_last)
        if architecture == 'resnet':
            self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2, trainable=next_trainable(),
                                    resample_filter=resample_filter, channels_last=self.channels_last)

    def prep_entries(self, force_fp32=False):
        dtype = self.fp16_dtype if self.use_fp16 and not force_fp32 else torch.float32
        return [e for name in ('fromrgb', 'conv0', 'conv1', 'skip') if hasattr(self, name)
                for e in getattr(self, name).prep_entries(dtype) if isinstance(getattr(self, name).weight, torch.nn.Parameter)]

    def forward(self, x, img, force_fp32=False):
        dtype = self.fp16_dtype if self.use_fp16 and not force_fp32 else torch.float32
        fmt = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        size = _spatial(self.resolution, self.init_res, self.conv_config_index)
        if x is not None:
            misc.assert_shape(x, [None, self.in_channels, *size])
            x = x.to(dtype=dtype, memory_format=fmt)
        if self.in_channels == 0 or self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, *size])
            img = img.to(dtype=dtype, memory_format=fmt)
            y = self.fromrgb(img)
            x = x + y if x is not None else y
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == 'skip' else None
        if self.architecture == 'resnet':
            g = np.sqrt(0.5)
            if (residual_bias_act.enabled and x.is_cuda and x.dtype != torch.float32 and fmt == torch.channels_last
                    and self.skip.bias is None and self.skip.activation == 'linear' and self.skip.conv_clamp is None
                    and self.conv1.activation in ('linear', 'lrelu') and self.conv1.out_channels_ in (64, 128, 256, 512)):
                # the two bias_act calls and the add as one pass over the block's output (ops/residual_bias_act.py); same values
                y_raw, _ = self.skip.conv_only(x)
                x = self.conv0(x)
                x_raw, b1 = self.conv1.conv_only(x)
                x = residual_bias_act.residual_bias_act(
                    x_raw, b1, y_raw, act=self.conv1.activation, gain=self.conv1.act_gain * g,
                    clamp=self.conv1.conv_clamp * g if self.conv1.conv_clamp is not None else None, r_gain=self.skip.act_gain * g)
            else:
                y = self.skip(x, gain=g)
                x = self.conv0(x)
                x = self.conv1(x, gain=g)
                x = y.add_(x)
        else:
            x = self.conv0(x)
            x = self.conv1(x)
        assert x.dtype == dtype
        return x, img


class MinibatchStdLayer(torch.nn.Module):
    """Appends the per-group standard deviation as extra channels (reference :667-691)."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x):
        # one native launch each way on the GPU (torch_utils/ops/mbstd.py), the aten composition of the reference elsewhere
        return mbstd.minibatch_std(x, self.group_size, self.num_channels)


class DiscriminatorEpilogue(torch.nn.Module):
    """mbstd -> 3x3 conv -> FC -> FC(1 or cmap)  (reference :696-756); always fp32."""

    def __init__(self, in_channels, cmap_dim, resolution, img_channels, init_res=None, architecture='resnet',
                 mbstd_group_size=4, mbstd_num_channels=1, activation='lrelu', conv_clamp=None):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels, self.cmap_dim, self.img_channels, self.architecture = in_channels, cmap_dim, img_channels, architecture
        self.init_res = _default(init_res, [4, 4])
        if architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * (self.init_res[0] * self.init_res[1]), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)

    def prep_entries(self):
        ents = self.conv.prep_entries(torch.float32) + self.fc.prep_entries() + self.out.prep_entries()
        if self.architecture == 'skip':
            ents += self.fromrgb.prep_entries(torch.float32)
        return ents

    def forward(self, x, img, cmap, force_fp32=False):
        misc.assert_shape(x, [None, self.in_channels, *self.init_res])
        x = x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        if self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, *self.init_res])
            x = x + self.fromrgb(img.to(dtype=torch.float32, memory_format=torch.contiguous_format))
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.out(self.fc(self.conv(x).flatten(1)))
        if self.cmap_dim > 0:
            misc.assert_shape(cmap, [None, self.cmap_dim])
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        assert x.dtype == torch.float32
        return x


class Discriminator(torch.nn.Module):
    """Residual discriminator from img_resolution down to 2**conv_config_index (reference :761-829)."""

    def __init__(self, img_resolution, img_channels, c_dim=0, init_res=None, architecture='resnet', channel_base=32768,
                 channel_max=512, num_fp16_res=0, conv_clamp=None, cmap_dim=None, block_kwargs=None, mapping_kwargs=None,
                 epilogue_kwargs=None, conv_config_index=2):
        super().__init__()
        self.c_dim, self.img_resolution, self.img_channels = c_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.init_res = _default(init_res, [4, 4])
        self.conv_config_index = conv_config_index
        base = 2 ** conv_config_index
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, conv_config_index, -1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [base]}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        if cmap_dim is None:
            cmap_dim = channels[base]
        if c_dim == 0:
            cmap_dim = 0
        common = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        layer_idx = 0
        for res in self.block_resolutions:
            block = DiscriminatorBlock(channels[res] if res < img_resolution else 0, channels[res], channels[res // 2],
                                       resolution=res, init_res=self.init_res, first_layer_idx=layer_idx,
                                       use_fp16=(res >= fp16_resolution), conv_config_index=conv_config_index,
                                       **(block_kwargs or {}), **common)
            setattr(self, f'b{res}', block)
            layer_idx += block.num_layers
        if c_dim > 0:
            self.mapping = MappingNetwork(z_dim=0, c_dim=c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None, **(mapping_kwargs or {}))
        self.b4 = DiscriminatorEpilogue(channels[base], cmap_dim=cmap_dim, resolution=base, init_res=self.init_res,
                                        **(epilogue_kwargs or {}), **common)
        self._prep = {}

    def forward(self, img, c=None, **block_kwargs):
        x = None
        force_fp32 = bool(block_kwargs.get('force_fp32', False))
        entries = self._prep.get(force_fp32)
        if entries is None:
            entries = self._prep[force_fp32] = weight_prep.Plan([e for res in self.block_resolutions for e in getattr(self, f'b{res}').prep_entries(force_fp32)] + self.b4.prep_entries())
        with weight_prep.scope(entries):
            for res in self.block_resolutions:
                x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
            cmap = self.mapping(None, c) if self.c_dim > 0 else None
            return self.b4(x, img, cmap)
