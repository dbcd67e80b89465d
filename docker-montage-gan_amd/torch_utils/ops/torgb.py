"""The generator's ToRGB layer as one pass over the activation each way.

What the reference executes for one ToRGB layer on the training path (training/networks.py:343-359 ``ToRGBLayer.forward``
-> ``modulated_conv2d(demodulate=False, fused_modconv=False)`` :62-88 -> ``bias_act(clamp)``; then ``SynthesisBlock.forward``
casts the result to float32 NCHW for the image path, :460-464):

    xs  = x * styles.to(x.dtype).reshape(N, C, 1, 1)            # modulation: read x, write xs
    yc  = conv2d(xs, weight.to(x.dtype))                        # 1x1, C -> img_channels
    out = bias_act(yc, bias.to(x.dtype), clamp=conv_clamp)
    y   = out.to(torch.float32, memory_format=torch.contiguous_format)

``torgb(x, weight, styles, bias, clamp)`` returns that ``y``.  On channels-last float16 / bfloat16 GPU activations with
64...512 channels it is ONE kernel (csrc/torgb.hip ``mg_torgb_nhwc_fwd``: every intermediate rounded exactly where the four
launches round it), and its plain backward is one kernel plus two [N,O,C]-sized contractions; everything else -- CPU tensors,
float32 blocks, ``create_graph=True`` (the path-length regulariser differentiates the generator twice) -- runs the composition
above out of this repo's own ops, so gradients of any order are available.  An additive op: callers that keep the four launches
get the same values.
"""
import torch

from .. import native
from . import bias_act as _ba
from . import conv2d_gradfix as _cg
from . import conv2d_resample as _cr
from . import modconv_epilogue as _me

enabled = True          # module switch (tests compare both ways)


def _composition(x, weight, styles, bias, clamp):
    xs = _me.channel_scale(x, styles)
    yc = _cr.conv2d_resample(x=xs, w=weight, f=None, up=1, down=1, padding=0, flip_weight=True)
    out = _ba.bias_act(yc, bias, clamp=clamp)
    return out.to(dtype=torch.float32, memory_format=torch.contiguous_format)


def _native_ok(x, weight):
    return (enabled and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.ndim == 4 and _me._is_nhwc(x)
            and weight.ndim == 4 and weight.shape[2] == weight.shape[3] == 1 and weight.shape[0] <= 4
            and x.shape[1] in (64, 128, 256, 512) and x.numel() > 0)


class _ToRGB(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, styles, bias, clamp):
        n, c, h, w = x.shape
        o = weight.shape[0]
        w2 = weight.detach().reshape(o, c).contiguous()
        s32 = styles.detach().to(torch.float32).contiguous()
        b = bias.detach().to(x.dtype).contiguous() if bias is not None else None
        y = torch.empty([n, o, h, w], dtype=torch.float32, device=x.device)
        with native.guard(x.device):
            native.check(native.lib().mg_torgb_nhwc_fwd(native.ptr(x), native.ptr(w2), native.ptr(s32), native.ptr(b), native.ptr(y),
                                                        native.dtype_code(x.dtype), n, c, h * w, o, clamp, native.stream_ptr(x.device)))
        ctx.save_for_backward(x, weight, styles, bias if bias is not None else _ba._null_tensor)
        ctx.clamp, ctx.has_bias = clamp, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, styles, bias = ctx.saved_tensors
        bias = bias if ctx.has_bias else None
        clamp = ctx.clamp
        n, c, h, w = x.shape
        o = weight.shape[0]
        need = ctx.needs_input_grad
        if torch.is_grad_enabled():
            # create_graph=True (the path-length regulariser differentiates the generator twice): the same gradients written out of
            # differentiable ops on the ORIGINAL tensors.  (Not torch.autograd.grad through a recomputed composition: asked for the
            # gradient w.r.t. `styles` -- one output of the network's batched affine product -- the engine would also run every
            # backward node between x and the sibling outputs of that product, i.e. the whole conv1 layer below x, once per ToRGB.)
            g = gy.to(dtype=x.dtype, memory_format=torch.channels_last)                         # the cast's backward
            st = styles.to(x.dtype)
            dxs = _cg.conv2d(g, weight.transpose(0, 1).contiguous(memory_format=torch.channels_last)) if (need[0] or need[2]) else None
            gx = dxs * st.reshape(n, c, 1, 1) if need[0] else None
            gs = (dxs * x).sum(dim=[2, 3]).to(styles.dtype) if need[2] else None
            gw = gbias = None
            if need[1]:
                xs = x * st.reshape(n, c, 1, 1)
                gw = torch.einsum('nop,ncp->oc', g.flatten(2), xs.flatten(2)).to(weight.dtype).reshape(weight.shape)
            if ctx.has_bias and need[3]:
                gbias = g.sum(dim=[0, 2, 3]).to(bias.dtype)
            return gx, gw, gs, gbias, None
        w2 = weight.reshape(o, c).contiguous()
        s32 = styles.to(torch.float32).contiguous()
        gy = gy.contiguous()
        gx = torch.empty_like(x)                    # keeps the channels-last layout
        # the linear activation saves no output: like the reference's bias_act backward, nothing is masked where the clamp saturated
        L = native.lib()
        acc = torch.zeros([n * o * c + o], dtype=torch.float32, device=x.device)
        G, gb32 = acc[:n * o * c], acc[n * o * c:]
        gs = torch.empty([n, c], dtype=torch.float32, device=x.device)
        gwb = torch.empty([o * c + o], dtype=x.dtype, device=x.device)
        gw, gb = gwb[:o * c], gwb[o * c:]
        dt, st = native.dtype_code(x.dtype), native.stream_ptr(x.device)
        with native.guard(x.device):
            native.check(L.mg_torgb_nhwc_bwd(native.ptr(x), native.ptr(w2), native.ptr(s32), native.ptr(gy), native.ptr(gx),
                                             native.ptr(G), native.ptr(gb32 if ctx.has_bias else None), dt, n, c, h * w, o, st))
            # dW[o,c] = sum_n styles.to(x.dtype)[n,c] * G[n,o,c], ds[n,c] = sum_o W[o,c] * G[n,o,c], db: one small launch
            native.check(L.mg_torgb_bwd_finish(native.ptr(G), native.ptr(gb32 if ctx.has_bias else None), native.ptr(s32), native.ptr(w2),
                                               native.ptr(gw), native.ptr(gs), native.ptr(gb if ctx.has_bias else None), dt, n, c, o, st))
        return ((gx if need[0] else None), (gw.view(weight.shape).to(weight.dtype) if need[1] else None),
                (gs.to(styles.dtype) if need[2] else None), (gb.to(bias.dtype) if ctx.has_bias and need[3] else None), None)


def torgb(x, weight, styles, bias=None, clamp=None):
    """x [N,C,H,W]; weight [O,C,1,1] (already in x.dtype); styles [N,C] (already multiplied by the layer's weight_gain);
    bias [O] or None; clamp: the layer's conv_clamp or None.  -> float32 contiguous [N,O,H,W]."""
    if _native_ok(x, weight):
        return _ToRGB.apply(x, weight, styles, bias, float(-1 if clamp is None else clamp))
    return _composition(x, weight, styles, bias, clamp)
