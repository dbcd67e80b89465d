"""Demodulation coefficients of the shared-weight modulated convolution as one native op.

``demod_coefs(weight, styles) = rsqrt(sum_{i,kh,kw} (weight[o,i,kh,kw] * styles[n,i])^2 + 1e-8)``  -> [N, O]
(reference training/networks.py:76-77, evaluated there on the materialised per-sample weights).  The composite form
``(styles.square() @ weight.square().sum([2,3]).t() + 1e-8).rsqrt()`` is six small kernels forward and nine backward
per SynthesisLayer call; csrc/demod.hip does it in one and two.  Double backward (path-length regularisation
differentiates the generator twice) goes through the composite expression, which is what the gradient is built from
whenever grad mode is on inside backward.  CPU tensors, non-float32 inputs and fp16's pre-normalised weights (new
tensors, not parameters -- still fine) take whichever path their device allows; there is no CPU fallback for GPU tensors.
"""
import torch

from .. import native


def _composite(weight, styles):
    wsq = weight.square().sum(dim=[2, 3])                   # [O,I]
    return (styles.square() @ wsq.t() + 1e-8).rsqrt()       # [N,O]


def _dense(weight):
    """(tensor, channels_last flag): the weight as the kernels read it -- in place when it is contiguous or channels-last
    (the layout of the reduced-precision blocks' parameters), a contiguous copy otherwise."""
    if weight.is_contiguous():
        return weight, 0
    if weight.is_contiguous(memory_format=torch.channels_last):
        return weight, 1
    return weight.contiguous(), 0


class _Demod(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, styles):
        (w, cl), s = _dense(weight), styles.contiguous()
        oc, ic, kh, kw = w.shape
        n = s.shape[0]
        d = torch.empty([n, oc], dtype=torch.float32, device=w.device)
        wsq = torch.empty([oc, ic], dtype=torch.float32, device=w.device)
        with native.guard(w.device):
            native.check(native.lib().mg_demod_fwd(w.data_ptr(), s.data_ptr(), d.data_ptr(), wsq.data_ptr(), n, oc, ic, kh * kw, cl,
                                                   native.stream_ptr(w.device)))
        ctx.save_for_backward(weight, styles, d, wsq)
        return d

    @staticmethod
    def backward(ctx, g):
        weight, styles, d, wsq = ctx.saved_tensors
        need_w, need_s = ctx.needs_input_grad
        if torch.is_grad_enabled():
            # create_graph=True: the gradient as a differentiable function of (g, weight, styles)
            wsq_ = weight.square().sum(dim=[2, 3])
            d_ = (styles.square() @ wsq_.t() + 1e-8).rsqrt()
            t = -0.5 * g * d_.pow(3)
            dw = (2 * weight * (t.t() @ styles.square())[:, :, None, None]) if need_w else None
            ds = (2 * styles * (t @ wsq_)) if need_s else None
            return dw, ds
        (w, cl), s, g = _dense(weight), styles.contiguous(), g.contiguous()
        oc, ic, kh, kw = w.shape
        n = s.shape[0]
        dw = torch.empty_like(w) if need_w else None           # same dense layout as w (preserve_format)
        ds = torch.empty_like(s) if need_s else None
        with native.guard(w.device):
            native.check(native.lib().mg_demod_bwd(w.data_ptr(), s.data_ptr(), d.data_ptr(), wsq.data_ptr(), g.data_ptr(),
                                                   ds.data_ptr() if need_s else None, dw.data_ptr() if need_w else None, n, oc, ic, kh * kw, cl,
                                                   native.stream_ptr(w.device)))
        return dw, ds


def demod_coefs(weight, styles, impl='cuda'):
    """weight [O,I,kh,kw], styles [N,I] -> [N,O]."""
    if (impl == 'cuda' and weight.device.type == 'cuda' and weight.dtype == torch.float32 and styles.dtype == torch.float32
            and weight.ndim == 4 and styles.ndim == 2 and styles.shape[0] <= 256 and weight.shape[1] == styles.shape[1]):
        return _Demod.apply(weight, styles)
    return _composite(weight, styles)
