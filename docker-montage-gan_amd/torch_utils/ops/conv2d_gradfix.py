"""conv2d / conv_transpose2d with a hand-written autograd graph.

Same interface as the reference (montage_gan/torch_utils/ops/conv2d_gradfix.py:22-51):
module switches ``enabled`` and ``weight_gradients_disabled``, the context manager
``no_weight_gradients()``, and ``conv2d`` / ``conv_transpose2d`` with
torch.nn.functional's signatures.

Why it exists: (1) R1 and path-length regularisation differentiate through the
gradient of every convolution, (2) inside ``no_weight_gradients()`` those passes skip
the weight-gradient GEMMs altogether, about a third of the backward FLOPs.

The reference implementation is pinned to torch 1.7-1.9 (it calls
``aten::cudnn_convolution_backward_weight``, gone since; :56-66,159-163) and silently
degrades to ``F.conv2d`` on anything newer.  This one is written against
``aten::convolution_backward`` (torch >= 1.11) with an output mask, runs on MIOpen
(ROCm) and on the CPU, and keeps the recursive structure: the input gradient of a
convolution is the opposite (transposed) convolution, expressed through the same
Function family so that it can be differentiated again.
"""
import contextlib

import torch

enabled = False                     # Enable the custom op by setting this to true.
weight_gradients_disabled = False   # Forcefully disable computation of gradients with respect to the weights.


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if _should_use_custom_op(input):
        return _conv2d_gradfix(transpose=False, weight_shape=weight.shape, stride=stride, padding=padding,
                               output_padding=0, dilation=dilation, groups=groups).apply(input, weight, bias)
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                      dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    if _should_use_custom_op(input):
        return _conv2d_gradfix(transpose=True, weight_shape=weight.shape, stride=stride, padding=padding,
                               output_padding=output_padding, groups=groups, dilation=dilation).apply(input, weight, bias)
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)


def _should_use_custom_op(input):
    assert isinstance(input, torch.Tensor)
    return bool(enabled)


def _pair(v):
    v = tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    assert len(v) == 2 and all(isinstance(e, int) for e in v)
    return v


_conv2d_gradfix_cache = dict()


_raw_cache = dict()


def _conv2d_gradfix(transpose, weight_shape, stride, padding, output_padding, dilation, groups):
    # first level: the arguments exactly as the layers pass them (ints / tuples / lists), no validation
    raw = (transpose, tuple(weight_shape), stride if type(stride) is int else tuple(stride), padding if type(padding) is int else tuple(padding),
           output_padding if type(output_padding) is int else tuple(output_padding), dilation if type(dilation) is int else tuple(dilation), groups)
    cls = _raw_cache.get(raw)
    if cls is None:
        cls = _raw_cache[raw] = _conv2d_gradfix_build(*raw)
    return cls


def _conv2d_gradfix_build(transpose, weight_shape, stride, padding, output_padding, dilation, groups):
    weight_shape = tuple(weight_shape)
    stride, padding, output_padding, dilation = _pair(stride), _pair(padding), _pair(output_padding), _pair(dilation)
    key = (transpose, weight_shape, stride, padding, output_padding, dilation, groups)
    cls = _conv2d_gradfix_cache.get(key)
    if cls is not None:
        return cls

    assert groups >= 1 and len(weight_shape) == 4
    assert all(s >= 1 for s in stride) and all(p >= 0 for p in padding) and all(d >= 0 for d in dilation)
    if not transpose:
        assert output_padding == (0, 0)
    else:
        assert all(0 <= output_padding[i] < max(stride[i], dilation[i]) for i in range(2))
    common = dict(stride=stride, padding=padding, dilation=dilation, groups=groups)

    def opposite_output_padding(input_shape, output_shape):
        """output_padding that makes the opposite op reproduce input_shape exactly."""
        if transpose:
            return (0, 0)
        return tuple(input_shape[i + 2] - (output_shape[i + 2] - 1) * stride[i] - (1 - 2 * padding[i])
                     - dilation[i] * (weight_shape[i + 2] - 1) for i in range(2))

    class Conv2d(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input, weight, bias):
            assert weight.shape == weight_shape
            ctx.save_for_backward(input, weight)
            if not transpose:
                return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, **common)
            return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias,
                                                        output_padding=output_padding, **common)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            if not torch.is_grad_enabled():
                # plain backward: both gradients from ONE aten::convolution_backward call (the same MIOpen kernels as the
                # two differentiable ops below, without two more python autograd nodes per convolution)
                need_w = bool(ctx.needs_input_grad[1] and not weight_gradients_disabled)
                if ctx.needs_input_grad[0] or need_w:
                    grad_input, grad_weight, _ = torch.ops.aten.convolution_backward(
                        grad_output, input, weight, None, list(stride), list(padding), list(dilation), transpose,
                        list(output_padding), groups, [bool(ctx.needs_input_grad[0]), need_w, False])
                    if not ctx.needs_input_grad[0]:
                        grad_input = None
                    if not need_w:
                        grad_weight = None
                if ctx.needs_input_grad[2]:
                    grad_bias = grad_output.sum([0, 2, 3])
                return grad_input, grad_weight, grad_bias
            if ctx.needs_input_grad[0]:
                op = _conv2d_gradfix(transpose=(not transpose), weight_shape=weight_shape,
                                     output_padding=opposite_output_padding(input.shape, grad_output.shape), **common)
                grad_input = op.apply(grad_output, weight, None)
                assert grad_input.shape == input.shape
            if ctx.needs_input_grad[1] and not weight_gradients_disabled:
                grad_weight = Conv2dGradWeight.apply(grad_output, input)
                assert grad_weight.shape == weight_shape
            if ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum([0, 2, 3])
            return grad_input, grad_weight, grad_bias

    class Conv2dGradWeight(torch.autograd.Function):
        @staticmethod
        def forward(ctx, grad_output, input):
            ctx.save_for_backward(grad_output, input)
            dummy_w = torch.empty(weight_shape, dtype=input.dtype, device=input.device)
            _, grad_weight, _ = torch.ops.aten.convolution_backward(
                grad_output, input, dummy_w, None, list(stride), list(padding), list(dilation), transpose,
                list(output_padding), groups, [False, True, False])
            assert grad_weight.shape == weight_shape
            return grad_weight

        @staticmethod
        def backward(ctx, grad2_grad_weight):
            grad_output, input = ctx.saved_tensors
            grad2_grad_output = grad2_input = None
            if ctx.needs_input_grad[0]:
                grad2_grad_output = Conv2d.apply(input, grad2_grad_weight, None)
                assert grad2_grad_output.shape == grad_output.shape
            if ctx.needs_input_grad[1]:
                op = _conv2d_gradfix(transpose=(not transpose), weight_shape=weight_shape,
                                     output_padding=opposite_output_padding(input.shape, grad_output.shape), **common)
                grad2_input = op.apply(grad_output, grad2_grad_weight, None)
                assert grad2_input.shape == input.shape
            return grad2_grad_output, grad2_input

    _conv2d_gradfix_cache[key] = Conv2d
    return Conv2d
