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


# Measurement hook (bench.py's MFMA roofline leg): while ``prof`` is a list, every convolution call made through this
# module appends (flops, dtype, start_event, end_event) -- HIP events on the launch stream around the aten call, so the
# interval covers the library's helper kernels (layout transposes, split-K zero fills) as well as the MFMA kernel.
prof = None


class _Timed:
    __slots__ = ('flops', 'dtype', 'a', 'tag')

    def __init__(self, flops, dtype, tag=None):
        self.flops, self.dtype, self.tag = flops, dtype, tag

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.a.record()
        return self

    def __exit__(self, *exc):
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        prof.append((self.flops, self.dtype, self.a, b, self.tag))
        return False


class _NoTime:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_no_time = _NoTime()


def _timed_call(input, weight_shape, n_products, kind):
    """Measurement scope for a stride-1, padding-k//2 convolution computed outside this module's autograd classes (the fused
    matrix-core kernel of ops/modconv3x3.py and the library gradients it calls): same accounting as ``timed`` below."""
    if prof is None or not input.is_cuda:
        return _no_time
    oc, ic, kh, kw = weight_shape
    tag = f'conv{kh}x{kw}s1 w{list(weight_shape)} x{list(input.shape)} nhwc {kind}'
    return _Timed(2.0 * input.shape[0] * input.shape[2] * input.shape[3] * oc * ic * kh * kw * n_products, input.dtype, tag)


# Data gradient of the square (C -> C, C >= 128) stride-1 3x3 convolutions on 16-bit channels-last tensors as a forward
# convolution of dy with the flipped, transposed weights (plain backward only).  Measured per problem in
# profiles/r02_conv_backward_split.log: 204 / 167 / 165 us against 280 / 244 / 224 us at 128 / 256 / 512 channels, batch 32.
data_gradient_as_forward = True


def _modconv3x3():
    from . import modconv3x3        # (imports this module)
    return modconv3x3


def _lowp_nhwc(t):
    return t.is_cuda and t.dtype in (torch.bfloat16, torch.float16) and t.dim() == 4 and t.stride(1) == 1 and t.shape[1] > 1


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

    def timed(input, grad_output_or_none, n_products=1):
        """2 * N * positions * Cin * Cout * kh * kw / groups per product (forward, data gradient and weight gradient cost the same)."""
        if prof is None or not input.is_cuda:
            return _no_time
        oc, icg, kh, kw = weight_shape
        if not transpose:       # dense side = the output lattice
            pos = (grad_output_or_none.shape[2] * grad_output_or_none.shape[3]) if grad_output_or_none is not None else None
            macs_per_pos = oc * icg * kh * kw
        else:                   # transposed: weight is [Cin, Cout/groups, kh, kw]; dense side = the input lattice
            pos = input.shape[2] * input.shape[3]
            macs_per_pos = oc * icg * kh * kw
        if pos is None:
            hout = (input.shape[2] + 2 * padding[0] - dilation[0] * (kh - 1) - 1) // stride[0] + 1
            wout = (input.shape[3] + 2 * padding[1] - dilation[1] * (kw - 1) - 1) // stride[1] + 1
            pos = hout * wout
        kind = ('fwd', 'bwd1', 'bwd2')[n_products if grad_output_or_none is not None else 0]
        tag = (('convT' if transpose else 'conv') + f'{kh}x{kw}s{stride[0]} w{list(weight_shape)} x{list(input.shape)} '
               + ('nhwc' if input.dim() == 4 and input.stride(1) == 1 and input.shape[1] > 1 else 'nchw') + ' ' + kind)
        return _Timed(2.0 * input.shape[0] * pos * macs_per_pos * n_products, input.dtype, tag)

    def opposite_output_padding(input_shape, output_shape):
        """output_padding that makes the opposite op reproduce input_shape exactly."""
        if transpose:
            return (0, 0)
        return tuple(input_shape[i + 2] - (output_shape[i + 2] - 1) * stride[i] - (1 - 2 * padding[i])
                     - dilation[i] * (weight_shape[i + 2] - 1) for i in range(2))

    square3x3s1 = (not transpose and stride == (1, 1) and dilation == (1, 1) and groups == 1 and weight_shape[2:] == (3, 3)
                   and weight_shape[0] == weight_shape[1] >= 128 and max(padding) <= 2)

    c64_3x3s1 = (not transpose and stride == (1, 1) and dilation == (1, 1) and groups == 1 and weight_shape == (64, 64, 3, 3))

    class Conv2d(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input, weight, bias):
            assert weight.shape == weight_shape
            ctx.save_for_backward(input, weight)
            with timed(input, None):
                if not transpose:
                    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, **common)
                return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias,
                                                            output_padding=output_padding, **common)

        @staticmethod
        def backward(ctx, grad_output):
            input, weight = ctx.saved_tensors
            grad_input = grad_weight = grad_bias = None
            if not torch.is_grad_enabled():
                # plain backward: aten::convolution_backward directly (the same MIOpen kernels as the two differentiable ops
                # below, without two more python autograd nodes per convolution) -- one call PER gradient: on every problem
                # of the step the two single-gradient calls together take no longer than the call that asks for both, and
                # up to 100 us less (profiles/r02_conv_backward_split.log)
                need_x = bool(ctx.needs_input_grad[0])
                need_w = bool(ctx.needs_input_grad[1] and not weight_gradients_disabled)
                if need_x or need_w:
                    with timed(input, grad_output, int(need_x) + int(need_w)):
                        bwd_args = (grad_output, input, weight, None, list(stride), list(padding), list(dilation), transpose,
                                    list(output_padding), groups)
                        if need_x and data_gradient_as_forward and square3x3s1 and _lowp_nhwc(grad_output):
                            # dx = dy (*) flipped, transposed w: the SAME problem as the layer's forward, whose solver runs at
                            # 0.8-1.05 PFLOP/s where the library's data-gradient solver reaches 0.5-0.7
                            wt = weight.flip([2, 3]).transpose(0, 1).contiguous(memory_format=torch.channels_last)
                            grad_input = torch.convolution(grad_output, wt, None, [1, 1], [2 - padding[0], 2 - padding[1]], [1, 1], False, [0, 0], 1)
                        elif need_x and c64_3x3s1 and padding == (1, 1) and _modconv3x3().applicable(grad_output, weight):
                            # 64 -> 64: the matrix-core kernel of ops/modconv3x3.py as a plain convolution of dy with the flipped,
                            # transposed weights
                            M = _modconv3x3()
                            grad_input = M.conv3x3_c64(grad_output.contiguous(memory_format=torch.channels_last), M.flipped_transposed(weight))
                        elif need_x:
                            grad_input = torch.ops.aten.convolution_backward(*bwd_args, [True, False, False])[0]
                        if need_w and c64_3x3s1 and padding == (1, 1) and _modconv3x3().wgrad_enabled and _modconv3x3().applicable(grad_output, weight) \
                                and _modconv3x3().applicable(input, weight):
                            grad_weight = _modconv3x3().wgrad3x3_c64(grad_output.contiguous(memory_format=torch.channels_last), input)[0]
                        elif need_w:
                            grad_weight = torch.ops.aten.convolution_backward(*bwd_args, [False, True, False])[1]
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
            with timed(input, grad_output):
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
