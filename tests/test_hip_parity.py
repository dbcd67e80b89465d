"""Parity tests proper: the hand-written HIP kernels (called through the C ABI, via the plugin
API) against the CPU oracle and the reference's golden vectors.  Need a real MI355X.

Tolerances: integer index path bit exact; float32 values within 1e-3 (BASELINE north star; most
checks use 1e-4/1e-5); 16-bit results within two ulps of the result's magnitude (float16 2^-10,
bfloat16 2^-7: LOWP_TOL below) unless a call states a wider bound and why."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden, load_pkg
from test_ops_ref_golden import run_bias_act_case, run_conv_case, run_upfirdn2d_case, T

pytestmark = pytest.mark.gpu
PKG = 'docker-montage-gan_amd'
DEV = 'cuda:0'


def op(name):
    load_pkg()
    return importlib.import_module(f'{PKG}.torch_utils.ops.{name}')


def custom_ops():
    load_pkg()
    return importlib.import_module(f'{PKG}.torch_utils.custom_ops')


@pytest.fixture(scope='module')
def native():
    n = load_pkg().native
    n.lib()
    return n


# Bound of one comparison, as a fraction of max(1, |want|max).  16-bit outputs are rounded ONCE from a float32 accumulator
# (half an ulp: 2^-12 fp16, 2^-9 bf16); the defaults allow two ulps (2^-10 / 2^-7) so that a reference that is itself a
# composition of 2-3 rounded T operations passes.  Anything looser is passed as `tol=` with the reason next to the call.
LOWP_TOL = {torch.float32: 1e-4, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7, torch.float64: 1e-6}


def close_lowp(got, want, dtype, what='', tol=None):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    scale = max(1.0, float(np.abs(want).max()))
    tol = LOWP_TOL[dtype] if tol is None else tol
    err = np.abs(got - want).max() / scale
    assert err <= tol, f'{what}: max err {err:.3e} > {tol:.1e} ({dtype})'


# --------------------------------------------------------------------------------------------
# upfirdn2d

def test_upfirdn2d_golden_fwd_bwd(native):
    U = op('upfirdn2d')
    cases, arr = load_golden('upfirdn2d')
    kinds = (native.K_UPFIRDN2D_GENERIC, native.K_UPFIRDN2D_TILED, native.K_UPFIRDN2D_STREAM, native.K_UPFIRDN2D_NHWC, native.K_UPFIRDN2D_SEP)
    before = sum(native.launch_count(k) for k in kinds)
    for c in cases:
        i = c['id']
        y, dx = run_upfirdn2d_case(U, c, arr, device=DEV, impl='cuda')
        assert list(y.shape) == c['out_shape'], c
        if c['memfmt'] == 'channels_last':
            assert y.is_contiguous(memory_format=torch.channels_last), c      # suggest_memory_format (upfirdn2d.cpp:35)
        np.testing.assert_allclose(y.detach().cpu().numpy(), arr[f'c{i}_y'], rtol=1e-4, atol=1e-5, err_msg=str(c))
        np.testing.assert_allclose(dx.cpu().numpy(), arr[f'c{i}_dx'], rtol=1e-4, atol=1e-5, err_msg=str(c))
    after = sum(native.launch_count(k) for k in kinds)
    assert after - before >= 2 * len(cases), 'the native kernels did not run'      # >= one forward and one backward launch per case


def test_upfirdn2d_index_table_bit_exact(native, oracle):
    """The integer index path computed ON THE DEVICE equals the oracle's, bit for bit."""
    for up, down, fs, pad0, in_size, flip in [(1, 1, 4, 1, 257, 0), (2, 1, 4, 2, 128, 0), (1, 2, 4, 1, 256, 1),
                                              (2, 1, 12, 6, 134, 0), (1, 2, 12, 5, 268, 1), (3, 2, 7, -3, 50, 0),
                                              (4, 3, 5, 11, 17, 1), (1, 1, 1, 0, 9, 0), (2, 2, 4, -1, 33, 0)]:
        out = oracle.out_size(in_size, up, down, pad0, pad0 + 1, fs)
        tab = torch.empty([out, 3], dtype=torch.int32, device=DEV)
        native.check(native.lib().mg_upfirdn2d_index_table(native.ptr(tab), out, in_size, up, down, pad0, fs, flip,
                                                           native.stream_ptr(tab.device)))
        assert np.array_equal(tab.cpu().numpy(), oracle.index_table(out, in_size, up, down, pad0, fs, flip))


HOT = [  # (shape, up, down, padding, gain, filter)  -- SURVEY 2.3 call sites and their gradients
    ([2, 3, 257, 257], 1, 1, [1, 1, 1, 1], 4.0, 'k4'),     # G conv0 post-blur
    ([2, 3, 256, 256], 1, 1, [2, 2, 2, 2], 1.0, 'k4'),     # D conv1 pre-blur / gradient of the above
    ([2, 4, 128, 128], 2, 1, [2, 1, 2, 1], 4.0, 'k4'),     # ToRGB image upsample
    ([2, 3, 256, 256], 1, 2, [1, 1, 1, 1], 1.0, 'k4'),     # D skip downsample
    ([2, 3, 128, 128], 2, 1, [1, 2, 1, 2], 1.0, 'k4'),     # gradient of the downsample (odd phase)
    ([3, 5, 65, 33], 1, 1, [1, 1, 1, 1], 4.0, 'k4'),
    ([3, 5, 17, 17], 1, 1, [1, 1, 1, 1], 4.0, 'k4'),
    ([2, 7, 16, 16], 1, 2, [1, 1, 1, 1], 1.0, 'k4'),
    ([2, 7, 8, 8], 2, 1, [2, 1, 2, 1], 4.0, 'k4'),
    ([1, 2, 100, 70], 1, 1, [1, 1, 0, 0], 1.0, 'k3'),
    ([1, 2, 70, 100], 2, 1, [3, -1, 0, 2], 1.0, 'k4'),
    ([2, 3, 134, 140], [2, 1], 1, [6, 5, 0, 0], 2.0, 'r12'),    # ADA separable passes
    ([2, 3, 140, 134], [1, 2], 1, [0, 0, 6, 5], 2.0, 'c12'),
    ([2, 3, 140, 268], 1, [2, 1], [5, 5, 0, 0], 1.0, 'r12'),
    ([2, 3, 268, 140], 1, [1, 2], [0, 0, 5, 5], 1.0, 'c12'),
    ([2, 3, 64, 64], 1, 1, [6, 5, 0, 0], 1.0, 'r12'),
    ([1, 1, 40, 40], 3, 2, [2, 2, 2, 2], 1.0, 'k5'),            # generic kernel
]


def _int_filter(kind):
    k = {'k4': [1, 3, 3, 1], 'k3': [1, 2, 1], 'k5': [1, 4, 6, 4, 1]}
    if kind in k:
        v = np.asarray(k[kind], dtype=np.float32)
        return np.outer(v, v)
    taps = np.asarray([1, -2, 3, -4, 5, 6, 7, -5, 4, -3, 2, -1], dtype=np.float32)
    return taps[None, :] if kind == 'r12' else taps[:, None]


@pytest.mark.parametrize('flip', [False, True])
def test_upfirdn2d_integer_data_bit_exact(native, oracle, flip):
    """Small-integer inputs and taps make every product and partial sum exact in fp32, so ANY error in the
    index path (which sample meets which tap) shows as a bit difference.  Covers every tiled
    specialisation, both tile phases and the generic kernel."""
    rng = np.random.RandomState(5)
    plugin = custom_ops().get_plugin('upfirdn2d_plugin')
    for shape, up, down, pad, gain, fk in HOT:
        x = rng.randint(-8, 9, size=shape).astype(np.float32)
        f = _int_filter(fk)
        up2 = (up, up) if isinstance(up, int) else tuple(up)
        dn2 = (down, down) if isinstance(down, int) else tuple(down)
        want = oracle.upfirdn2d(x, f, up2, dn2, tuple(pad), flip, gain)
        got = plugin.upfirdn2d(T(x).to(DEV), T(f).to(DEV), up2[0], up2[1], dn2[0], dn2[1], pad[0], pad[1], pad[2], pad[3],
                               flip, gain)
        assert got.shape == want.shape, (shape, up, down, pad)
        assert np.array_equal(got.cpu().numpy(), want), (shape, up, down, pad, fk, flip)


@pytest.mark.parametrize('flip', [False, True])
def test_upfirdn2d_sep12_one_pass_bit_exact(native, oracle, flip):
    """mg_upfirdn2d_sep12 (12-tap 1-D filter, x2 up or down in both axes, one launch) behind upfirdn2d(f.ndim == 1): small-integer
    data against the oracle's 2-D form with outer(f, f) -- every polyphase branch (pad0 parity per axis), negative pads, ragged
    sizes, several tiles per axis, a 1-pixel input; one launch per call."""
    U = op('upfirdn2d')
    rng = np.random.RandomState(12)
    taps = np.asarray([1, -2, 3, -4, 5, 6, 7, -5, 4, -3, 2, -1], dtype=np.float32)
    f1 = T(taps).to(DEV)
    f2 = np.outer(taps, taps)
    cases = []
    for shape in ([2, 3, 17, 23], [1, 4, 64, 64], [2, 4, 70, 130], [1, 1, 1, 1], [1, 2, 33, 97]):
        for pad in ([6, 5, 6, 5], [5, 6, 6, 5], [6, 5, 5, 6], [7, 7, 7, 7], [-3, 8, -2, 9], [11, 11, 11, 11]):
            cases.append((shape, 2, 1, pad))
    for shape in ([2, 3, 40, 51], [1, 4, 128, 128], [1, 2, 150, 300], [1, 1, 12, 12], [2, 1, 75, 141]):
        for pad in ([-7, -7, -7, -7], [5, 5, 5, 5], [0, 3, 7, -1], [6, 5, 5, 6], [1, 0, 0, 1]):
            cases.append((shape, 1, 2, pad))
    ran = 0
    for shape, up, down, pad in cases:
        oh = (shape[2] * up + pad[2] + pad[3] - 12 + down) // down
        ow = (shape[3] * up + pad[0] + pad[1] - 12 + down) // down
        if oh < 1 or ow < 1:
            continue
        x = rng.randint(-8, 9, size=shape).astype(np.float32)
        want = oracle.upfirdn2d(x, f2, (up, up), (down, down), tuple(pad), flip, 4.0)
        before = native.launch_count(native.K_UPFIRDN2D_SEP)
        got = U.upfirdn2d(T(x).to(DEV), f1, up=up, down=down, padding=pad, flip_filter=flip, gain=4)
        assert native.launch_count(native.K_UPFIRDN2D_SEP) == before + 1, 'the one-pass kernel did not run'
        assert got.shape == want.shape, (shape, up, down, pad)
        assert np.array_equal(got.cpu().numpy(), want), (shape, up, down, pad, flip)
        ran += 1
    assert ran >= 50


def test_upfirdn2d_sep12_matches_two_passes_and_its_adjoint(native):
    """Random float data: the one-pass kernel against the two mg_upfirdn2d passes it replaces (same taps, same order, sqrt(gain)
    after each axis: 1e-6), forward and gradient (the adjoint runs through the same kernel with up <-> down), at the ADA sizes."""
    U = op('upfirdn2d')
    AU = importlib.import_module(f'{PKG}.training.augment')
    f12 = U.setup_filter(AU.wavelets['sym6']).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    for kind, shape in (('up', [4, 4, 300, 280]), ('down', [4, 4, 536, 600]), ('up', [2, 3, 37, 41]), ('down', [2, 3, 90, 77])):
        x = torch.randn(shape, device=DEV, generator=g).requires_grad_(True)
        res = []
        for enabled in (True, False):
            U.sep12_enabled = enabled
            try:
                before = native.launch_count(native.K_UPFIRDN2D_SEP)
                y = U.upsample2d(x, f12, up=2) if kind == 'up' else U.downsample2d(x, f12, down=2, padding=-12, flip_filter=True)
                dy = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
                dx, = torch.autograd.grad([y], [x], [dy])
                assert native.launch_count(native.K_UPFIRDN2D_SEP) == before + (2 if enabled else 0)
                res.append((y.detach(), dx))
            finally:
                U.sep12_enabled = True
        for a, b, nm in zip(res[0], res[1], ('y', 'dx')):
            assert a.shape == b.shape
            err = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
            assert err <= 1e-6, f'{kind} {shape} {nm}: {err:.3g}'


def test_reflect_pad_upsample_one_launch(native, oracle):
    """reflect_pad_upsample2d (augment.py:315-317 in one launch, the padded image never written): integer data bit-exact against
    numpy's reflect padding + the oracle's upfirdn2d; random data, first and second derivatives against the two-op composition."""
    U = op('upfirdn2d')
    AU = importlib.import_module(f'{PKG}.training.augment')
    rng = np.random.RandomState(8)
    taps = np.asarray([1, -2, 3, -4, 5, 6, 7, -5, 4, -3, 2, -1], dtype=np.float32)
    for shape, m in (([2, 3, 9, 14], [3, 5, 8, 0]), ([1, 4, 64, 64], [63, 63, 63, 63]), ([2, 2, 40, 33], [0, 0, 0, 0]), ([1, 1, 70, 90], [17, 2, 1, 30])):
        x = rng.randint(-8, 9, size=shape).astype(np.float32)
        xp = np.pad(x, [(0, 0), (0, 0), (m[2], m[3]), (m[0], m[1])], mode='reflect')
        want = oracle.upfirdn2d(xp, np.outer(taps, taps), (2, 2), (1, 1), (6, 5, 6, 5), False, 4.0)
        before = native.launch_counts()
        got = U.reflect_pad_upsample2d(T(x).to(DEV), T(taps).to(DEV), m)
        after = native.launch_counts()
        assert after['upfirdn2d_sep'] == before['upfirdn2d_sep'] + 1 and sum(after.values()) == sum(before.values()) + 1
        assert np.array_equal(got.cpu().numpy(), want), (shape, m)
    f12 = U.setup_filter(AU.wavelets['sym6']).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn([3, 4, 48, 40], device=DEV, generator=g).requires_grad_(True)
    m = [7, 20, 11, 3]
    res = []
    for fused in (True, False):
        y = U.reflect_pad_upsample2d(x, f12, m) if fused else U.upsample2d(torch.nn.functional.pad(x, m, mode='reflect'), f12, up=2)
        dy = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)).requires_grad_(True)
        dx, = torch.autograd.grad([y], [x], [dy], create_graph=True)
        ddy, = torch.autograd.grad([dx.square().sum()], [dy])          # second order: d/d(dy) of |adjoint(dy)|^2
        res.append((y.detach(), dx.detach(), ddy))
    for a, b, nm in zip(res[0], res[1], ('y', 'dx', 'ddy')):
        err = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
        assert a.shape == b.shape and err <= 1e-5, f'{nm}: {err:.3g}'


def test_ada_resamplers_full_size_adjoint_pairs(native):
    """At the benchmark's sizes (batch 32, RGBA, 256^2 with its margins) the oracle is too slow; the size-independent property of a
    linear map and its gradient is <A x, y> = <x, A^T y>.  Checked for the one-pass upsampler (with the folded reflect padding), the
    downsampler, and the affine warp between them, with float64 dot products: 1e-5 of |Ax||y|."""
    U, W = op('upfirdn2d'), op('affine_warp')
    AU = importlib.import_module(f'{PKG}.training.augment')
    f12 = U.setup_filter(AU.wavelets['sym6']).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    ang = torch.rand([32], device=DEV, generator=g) * 6.283
    theta = (torch.stack([torch.cos(ang), -torch.sin(ang), torch.zeros_like(ang), torch.sin(ang), torch.cos(ang), torch.zeros_like(ang)], 1) * 0.8).reshape(32, 2, 3)
    maps = (('reflect pad + upsample', [32, 4, 256, 256], lambda x: U.reflect_pad_upsample2d(x, f12, [31, 33, 32, 30])),
            ('downsample', [32, 4, 536, 536], lambda x: U.downsample2d(x, f12, down=2, padding=-12, flip_filter=True)),
            ('affine warp', [32, 4, 640, 640], lambda x: W.affine_warp(x, theta, 536, 536)))
    for name, shape, fn in maps:
        x = torch.randn(shape, device=DEV, generator=g).requires_grad_(True)
        before = native.launch_counts()
        ax = fn(x)
        y = torch.randn(ax.shape, device=DEV, generator=g)
        aty, = torch.autograd.grad([ax], [x], [y])
        after = native.launch_counts()
        assert sum(after.values()) == sum(before.values()) + 2, f'{name}: one native launch forward, one backward'
        lhs, rhs = (ax.detach().double() * y.double()).sum().item(), (x.detach().double() * aty.double()).sum().item()
        scale = ax.detach().double().norm().item() * y.double().norm().item()
        assert abs(lhs - rhs) <= 1e-5 * scale, f'{name}: <Ax,y> = {lhs:.9g}, <x,A^T y> = {rhs:.9g}'


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_upfirdn2d_stream_kernel_bit_exact_and_edges(native, oracle, dtype):
    """The wave-shuffle streaming kernel (up = down = 1, <= 4x4 taps, width >= 48): odd and even row
    pitches (both 16-bit pair alignments), every pad sign, strip/chunk boundaries, exact on small integers."""
    rng = np.random.RandomState(21)
    plugin = custom_ops().get_plugin('upfirdn2d_plugin')
    lo, hi = (-8, 9) if dtype != torch.bfloat16 else (-1, 2)
    taps = {'k4': [1, 3, 3, 1], 'k3': [1, 2, 1], 'k2': [1, 1]} if dtype != torch.bfloat16 else {'k2': [1, 1], 'k4b': [1, 1, 1, 1]}
    for shape in ([2, 3, 257, 257], [1, 2, 64, 125], [1, 2, 70, 124], [3, 1, 131, 250], [1, 1, 5, 49], [2, 2, 48, 48], [1, 3, 129, 383]):
        for pad in ([1, 1, 1, 1], [2, 2, 2, 2], [0, 0, 0, 0], [3, 0, -1, 2], [-2, 3, 1, 1]):
            for fk, tp in taps.items():
                v = np.asarray(tp, dtype=np.float32)
                f = np.outer(v, v)
                if shape[2] + pad[2] + pad[3] - f.shape[0] + 1 < 1 or shape[3] + pad[0] + pad[1] - f.shape[1] + 1 < 48:
                    continue
                x = rng.randint(lo, hi, size=shape).astype(np.float32)
                for flip in (False, True):
                    want = oracle.upfirdn2d(x, f, (1, 1), (1, 1), tuple(pad), flip, 1.0)
                    got = plugin.upfirdn2d(T(x).to(dtype).to(DEV), T(f).to(DEV), 1, 1, 1, 1, pad[0], pad[1], pad[2], pad[3], flip, 1.0)
                    assert got.dtype == dtype and tuple(got.shape) == want.shape
                    assert np.array_equal(got.float().cpu().numpy(), want), (shape, pad, fk, flip, dtype)
    # random data, asymmetric filter, gain
    f = rng.randn(3, 4).astype(np.float32)
    x = rng.randn(2, 3, 100, 131).astype(np.float32)
    want = oracle.upfirdn2d(T(x).to(dtype).float().numpy(), f, (1, 1), (1, 1), (2, 1, 1, 1), False, 2.5)
    got = plugin.upfirdn2d(T(x).to(dtype).to(DEV), T(f).to(DEV), 1, 1, 1, 1, 2, 1, 1, 1, False, 2.5)
    close_lowp(got.float().cpu().numpy(), want, dtype, 'stream random')


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_upfirdn2d_nhwc_kernel_bit_exact_and_edges(native, oracle, dtype):
    """The channels-last kernel (dense NHWC, <= 4x4 taps, up/down in {1,2}): every polyphase instantiation (both
    tile phases per axis for up=2), odd sizes, chunk boundaries (> 32 output rows), negative pads, flipped taps --
    exact on small integers, so any index-path slip is a bit difference."""
    rng = np.random.RandomState(33)
    plugin = custom_ops().get_plugin('upfirdn2d_plugin')
    vec = 16 // torch.empty([], dtype=dtype).element_size()
    lo, hi = (-8, 9) if dtype != torch.bfloat16 else (-1, 2)
    taps = {'k4': [1, 3, 3, 1], 'k3': [1, 2, 1]} if dtype != torch.bfloat16 else {'k2': [1, 1], 'k4b': [1, 1, 1, 1]}
    cfgs = [(1, 1, [1, 1, 1, 1]), (1, 1, [2, 2, 2, 2]), (1, 1, [3, 0, -1, 2]), (2, 1, [2, 1, 2, 1]), (2, 1, [1, 2, 2, 1]),
            (2, 1, [2, 1, 1, 2]), (2, 1, [3, 2, 1, 0]), (1, 2, [1, 1, 1, 1]), (1, 2, [0, 2, 2, 0]), (1, 2, [-1, 3, 2, 1])]
    before = native.launch_count(native.K_UPFIRDN2D_NHWC)
    ncalls = 0
    for shape in ([2, vec, 37, 45], [1, 2 * vec, 70, 33], [2, 3 * vec, 8, 8], [1, vec, 131, 5], [1, vec, 230, 9]):    # >= 192 rows: the row-streaming variant
        for up, down, pad in cfgs:
            for fk, tp in taps.items():
                v = np.asarray(tp, dtype=np.float32)
                f = np.outer(v, v)
                if (shape[2] * up + pad[2] + pad[3] - f.shape[0]) // down + 1 < 1 or (shape[3] * up + pad[0] + pad[1] - f.shape[1]) // down + 1 < 1:
                    continue
                x = rng.randint(lo, hi, size=shape).astype(np.float32)
                for flip in (False, True):
                    want = oracle.upfirdn2d(x, f, (up, up), (down, down), tuple(pad), flip, 1.0)
                    xg = T(x).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
                    got = plugin.upfirdn2d(xg, T(f).to(DEV), up, up, down, down, pad[0], pad[1], pad[2], pad[3], flip, 1.0)
                    ncalls += 1
                    assert got.dtype == dtype and tuple(got.shape) == want.shape
                    assert got.is_contiguous(memory_format=torch.channels_last)
                    assert np.array_equal(got.float().cpu().numpy(), want), (shape, up, down, pad, fk, flip, dtype)
    assert native.launch_count(native.K_UPFIRDN2D_NHWC) - before == ncalls, 'the channels-last kernel did not take these calls'
    # random data, asymmetric filter, gain, network-like shape
    f = rng.randn(4, 3).astype(np.float32)
    x = rng.randn(2, 4 * vec, 65, 65).astype(np.float32)
    for up, down, pad, gain in [(1, 1, (1, 1, 1, 1), 4.0), (2, 1, (2, 1, 2, 1), 4.0), (1, 2, (1, 1, 1, 1), 1.0)]:
        want = oracle.upfirdn2d(T(x).to(dtype).float().numpy(), f, (up, up), (down, down), pad, False, gain)
        xg = T(x).to(dtype).to(DEV).contiguous(memory_format=torch.channels_last)
        got = plugin.upfirdn2d(xg, T(f).to(DEV), up, up, down, down, pad[0], pad[1], pad[2], pad[3], False, gain)
        close_lowp(got.float().cpu().numpy(), want, dtype, f'nhwc random up{up} down{down}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_channels_last_epilogue_and_backward_match_nchw(native, dtype):
    """The NHWC epilogue / fused-backward / bias_act kernels against the NCHW kernels on the same values (the NCHW
    ones are pinned to the oracle and the golden vectors above): forward and dx bit-identical (same arithmetic,
    different addressing), the float32-accumulated reductions within rounding of the summation order."""
    E, B = op('modconv_epilogue'), op('bias_act')
    g = torch.Generator().manual_seed(8)
    vec = 16 // torch.empty([], dtype=dtype).element_size()
    for n, c, h, w in [(3, 4 * vec, 9, 7), (2, vec, 33, 32), (2, 16 * vec, 5, 5)]:
        x0 = torch.randn([n, c, h, w], generator=g).to(dtype).to(DEV)
        d0 = (torch.rand([n, c], generator=g) + 0.5).to(DEV)
        b0 = torch.randn([c], generator=g).to(DEV)
        nz = torch.randn([h, w], generator=g).to(DEV)
        nzn = torch.randn([n, 1, h, w], generator=g).to(DEV)
        gy = torch.randn([n, c, h, w], generator=g).to(dtype).to(DEV)
        for noise in (nz, nzn, None):
            res = []
            for fmt in (torch.contiguous_format, torch.channels_last):
                x = x0.clone().contiguous(memory_format=fmt).requires_grad_(True)
                d, b = d0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
                st = torch.tensor(0.7, device=DEV, requires_grad=True)
                k0 = native.launch_count(native.K_FMA_BIAS_ACT), native.launch_count(native.K_BWD_FUSED)
                y = E.modconv_epilogue(x, d, noise, b, act='lrelu', clamp=1.5, noise_strength=st if noise is not None else None)
                grads = torch.autograd.grad([y], [x, d, b] + ([st] if noise is not None else []), [gy.contiguous(memory_format=fmt)])
                assert native.launch_count(native.K_FMA_BIAS_ACT) == k0[0] + 1 and native.launch_count(native.K_BWD_FUSED) == k0[1] + 1
                if fmt == torch.channels_last:
                    assert y.is_contiguous(memory_format=torch.channels_last) and grads[0].is_contiguous(memory_format=torch.channels_last)
                res.append([y.detach()] + [t.detach() for t in grads])
            assert torch.equal(res[0][0], res[1][0]), 'epilogue forward'
            assert torch.equal(res[0][1], res[1][1]), 'epilogue dx'
            for a, bb, what in zip(res[0][2:], res[1][2:], ['gd', 'gb', 'gs']):
                close_lowp(bb.float().cpu().numpy(), a.float().cpu().numpy(), torch.float32 if dtype == torch.float32 else dtype, what)
        # channel_scale
        res = []
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = x0.clone().contiguous(memory_format=fmt).requires_grad_(True)
            d = d0.clone().requires_grad_(True)
            y = E.channel_scale(x, d)
            gx, gd = torch.autograd.grad([y], [x, d], [gy.contiguous(memory_format=fmt)])
            res.append((y.detach(), gx, gd))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), 'channel_scale'
        close_lowp(res[1][2].float().cpu().numpy(), res[0][2].float().cpu().numpy(), torch.float32 if dtype == torch.float32 else dtype, 'channel_scale gd')
        # bias_act with its fused bias gradient
        res = []
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = x0.clone().contiguous(memory_format=fmt).requires_grad_(True)
            b = b0.to(dtype).clone().requires_grad_(True)
            y = B.bias_act(x, b, act='lrelu', clamp=1.5)
            gx, gb = torch.autograd.grad([y], [x, b], [gy.contiguous(memory_format=fmt)])
            res.append((y.detach(), gx, gb))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), 'bias_act nhwc'
        close_lowp(res[1][2].float().cpu().numpy(), res[0][2].float().cpu().numpy(), dtype, 'bias_act db')


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16, torch.float64])
def test_upfirdn2d_dtypes_vs_oracle(native, oracle, dtype):
    rng = np.random.RandomState(11)
    U = op('upfirdn2d')
    f = U.setup_filter([1, 3, 3, 1])
    for shape, kw in [([2, 3, 33, 33], dict(padding=[1, 1, 1, 1], gain=4)), ([2, 4, 16, 16], dict(up=2, padding=[2, 1, 2, 1], gain=4)),
                      ([2, 3, 32, 32], dict(down=2, padding=[1, 1, 1, 1])), ([1, 2, 9, 11], dict(up=3, down=2, padding=[2, 2, 2, 2]))]:
        x = T(rng.randn(*shape).astype(np.float32)).to(dtype)
        want = oracle.upfirdn2d_api(x.to(torch.float64 if dtype == torch.float64 else torch.float32).numpy(), f.numpy(), **kw)
        got = U.upfirdn2d(x.to(DEV), f.to(DEV), **kw)
        assert got.dtype == dtype
        close_lowp(got.double().cpu().numpy(), want, dtype, str((shape, kw)))


def test_upfirdn2d_strided_views_and_channels_last(native, oracle):
    rng = np.random.RandomState(3)
    U = op('upfirdn2d')
    f = U.setup_filter([1, 3, 3, 1])
    big = T(rng.randn(2, 6, 20, 24).astype(np.float32)).to(DEV)
    view = big[:, 1:5, 2:18, 3:21]                                  # non-dense view -> generic kernel
    want = oracle.upfirdn2d_api(view.cpu().numpy(), f.numpy(), up=2, padding=[2, 1, 2, 1], gain=4)
    np.testing.assert_allclose(U.upfirdn2d(view, f.to(DEV), up=2, padding=[2, 1, 2, 1], gain=4).cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    cl = big.contiguous(memory_format=torch.channels_last)
    y = U.upfirdn2d(cl, f.to(DEV), down=2, padding=[1, 1, 1, 1])
    assert y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(y.cpu().numpy(), oracle.upfirdn2d_api(big.cpu().numpy(), f.numpy(), down=2, padding=[1, 1, 1, 1]), rtol=1e-5, atol=1e-5)


def test_upfirdn2d_full_size_properties(native):
    """BASELINE-size tensors (too big for the oracle loop): linearity, DC preservation, and
    <upfirdn(x), y> == <x, upfirdn^T(y)> (the backward is the adjoint)."""
    U = op('upfirdn2d')
    f = U.setup_filter([1, 3, 3, 1]).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn([8, 64, 257, 257], device=DEV, generator=g)
    z = torch.randn([8, 64, 257, 257], device=DEV, generator=g)
    kw = dict(padding=[1, 1, 1, 1], gain=4)
    lin = U.upfirdn2d(2 * x - 3 * z, f, **kw) - (2 * U.upfirdn2d(x, f, **kw) - 3 * U.upfirdn2d(z, f, **kw))
    assert float(lin.abs().max()) < 2e-4
    ones = torch.ones([2, 4, 128, 128], device=DEV)
    up = U.upsample2d(ones, f)
    assert up.shape == (2, 4, 256, 256)
    assert float((up[:, :, 4:-4, 4:-4] - 1).abs().max()) < 1e-6
    xr = x[:2].clone().requires_grad_(True)
    y = U.upfirdn2d(xr, f, down=2, padding=[1, 1, 1, 1])
    w = torch.randn(y.shape, device=DEV, generator=g)
    gx, = torch.autograd.grad([y], [xr], [w])
    lhs, rhs = float((y.detach().double() * w.double()).sum()), float((xr.detach().double() * gx.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_upfirdn2d_errors(native):
    U = op('upfirdn2d')
    plugin = custom_ops().get_plugin('upfirdn2d_plugin')
    x = torch.zeros([1, 1, 4, 4], device=DEV)
    f = torch.ones([2, 2], device=DEV)
    with pytest.raises(RuntimeError, match='output must be at least 1x1'):
        plugin.upfirdn2d(x, torch.ones([8, 8], device=DEV), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match='f must be float32'):
        plugin.upfirdn2d(x, f.double(), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match='x must be rank 4'):
        plugin.upfirdn2d(x[0], f, 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match='x must reside on CUDA device'):
        plugin.upfirdn2d(x.cpu(), f.cpu(), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    # empty batch is legal
    assert U.upfirdn2d(torch.zeros([0, 3, 8, 8], device=DEV), f, padding=1).shape == (0, 3, 9, 9)


# --------------------------------------------------------------------------------------------
# bias_act

def test_bias_act_golden_all_orders(native):
    B = op('bias_act')
    cases, arr = load_golden('bias_act')
    before = native.launch_count(native.K_BIAS_ACT)
    for c in cases:
        i = c['id']
        out = run_bias_act_case(B, c, arr, device=DEV, impl='cuda')
        x = arr[f'c{i}_x']
        b = arr[f'c{i}_b'] if c['with_b'] else None
        dim = c['dim']
        pre = x + (0 if b is None else b.reshape([-1 if k == dim else 1 for k in range(x.ndim)]))
        mask = np.ones(x.shape, bool)
        if c['act'] in ('relu', 'lrelu', 'elu', 'selu'):
            mask &= np.abs(pre) > 1e-6           # one-sided derivative at the kink: kernel vs aten convention
        if c['clamp'] is not None:
            mask &= np.abs(np.abs(arr[f'c{i}_y']) - c['clamp']) > 1e-5
        tol = dict(rtol=2e-4, atol=2e-5) if c['dtype'] == 'float32' else dict(rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out['y'].detach().cpu().numpy(), arr[f'c{i}_y'], err_msg=str(c), **tol)
        np.testing.assert_allclose(out['dx'].detach().cpu().numpy()[mask], arr[f'c{i}_dx'][mask], err_msg=str(c), **tol)
        if out['db'] is not None and mask.all():
            np.testing.assert_allclose(out['db'].detach().cpu().numpy(), arr[f'c{i}_db'], err_msg=str(c), rtol=1e-3, atol=1e-4)
        if 'g2x' in out:
            np.testing.assert_allclose(out['g2dy'].cpu().numpy()[mask], arr[f'c{i}_g2dy'][mask], err_msg=str(c), **tol)
            np.testing.assert_allclose(out['g2x'].cpu().numpy()[mask], arr[f'c{i}_g2x'][mask], err_msg=str(c), rtol=1e-3, atol=1e-4)
    assert native.launch_count(native.K_BIAS_ACT) - before >= 2 * len(cases)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('memfmt', ['contiguous', 'channels_last'])
def test_bias_act_raw_vs_oracle(native, oracle, dtype, memfmt):
    """Plugin-level call on every vector/bias-index path: aligned + step_b % VEC == 0 (NCHW), bias per
    element (NHWC / odd planes), tails, all three gradient orders."""
    rng = np.random.RandomState(17)
    plugin = custom_ops().get_plugin('bias_act_plugin')
    null = torch.empty([0])
    for shape in ([2, 6, 8, 8], [3, 5, 7, 9], [2, 16, 4, 4], [4, 512], [1, 3, 1, 5]):
        if memfmt == 'channels_last' and len(shape) != 4:
            continue
        x = T(rng.randn(*shape).astype(np.float32) * 2).to(dtype).to(DEV)
        xr = T(rng.randn(*shape).astype(np.float32)).to(dtype).to(DEV)
        yr = T(rng.randn(*shape).astype(np.float32)).to(dtype).to(DEV)
        dy = T(rng.randn(*shape).astype(np.float32)).to(dtype).to(DEV)
        b = T(rng.randn(shape[1]).astype(np.float32)).to(dtype).to(DEV)
        if memfmt == 'channels_last':
            x, xr, yr, dy = (t.contiguous(memory_format=torch.channels_last) for t in (x, xr, yr, dy))

        def phys(t):   # dense physical order, as the kernel sees it
            t = t.float().cpu()
            return (t.permute(0, 2, 3, 1) if memfmt == 'channels_last' else t).contiguous().numpy()

        dim_phys = 3 if memfmt == 'channels_last' else 1
        for act in (1, 3, 5, 9):
            for grad in (0, 1, 2):
                if grad == 2 and act in (1, 3):
                    continue
                got = plugin.bias_act(x, b, xr if grad else null, yr if grad else null, dy if grad == 2 else null, grad, 1, act,
                                      0.2, 1.4142135, 1.5)
                assert got.stride() == x.stride()
                want = oracle.bias_act_raw(phys(x), b.float().cpu().numpy(), phys(xr) if grad else None, phys(yr) if grad else None,
                                           phys(dy) if grad == 2 else None, grad=grad, dim=dim_phys, act=act, alpha=0.2,
                                           gain=1.4142135, clamp=1.5)
                close_lowp(phys(got), want, dtype, f'{shape} act {act} grad {grad} {memfmt}')


def test_bias_act_unaligned_and_empty(native, oracle):
    plugin = custom_ops().get_plugin('bias_act_plugin')
    null = torch.empty([0])
    base = torch.randn([1003], device=DEV)
    x = base[1:].reshape(1, 2, 501)             # storage offset 4 bytes -> not 16-byte aligned
    b = torch.randn([2], device=DEV)
    got = plugin.bias_act(x, b, null, null, null, 0, 1, 3, 0.2, 1.0, -1.0)
    want = oracle.bias_act_raw(x.cpu().numpy(), b.cpu().numpy(), grad=0, dim=1, act=3, alpha=0.2, gain=1.0, clamp=-1.0)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    B = op('bias_act')
    assert B.bias_act(torch.zeros([0, 4, 2, 2], device=DEV), torch.zeros([4], device=DEV), act='lrelu').shape == (0, 4, 2, 2)
    with pytest.raises(RuntimeError, match='b has wrong number of elements'):
        plugin.bias_act(torch.zeros([2, 3], device=DEV), torch.zeros([4], device=DEV), null, null, null, 0, 1, 1, 0.0, 1.0, -1.0)


@pytest.mark.parametrize('act,gain,clamp', [('linear', 0.7, 0.5), ('linear', 1.0, 0.5), ('linear', 0.7, None), ('lrelu', 0.7, 0.5), ('relu', None, None)])
@pytest.mark.parametrize('memfmt', ['contiguous', 'channels_last'])
def test_bias_act_plain_backward_matches_ref_impl(native, act, gain, clamp, memfmt):
    """``.backward()`` WITHOUT create_graph (the one-pass fused backward kernel where it applies, the gradient kernel
    otherwise) against impl='ref' -- in particular act='linear' with gain != 1, which saves no output to evaluate the
    gain against and therefore must not take the fused path."""
    B = op('bias_act')
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn([3, 8, 10, 12], device=DEV, generator=g)
    if memfmt == 'channels_last':
        x = x.contiguous(memory_format=torch.channels_last)
    b = torch.randn([8], device=DEV, generator=g)
    dy = torch.randn([3, 8, 10, 12], device=DEV, generator=g)
    res = []
    for impl in ('cuda', 'ref'):
        xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = B.bias_act(xr, br, act=act, gain=gain, clamp=clamp, impl=impl)
        y.backward(dy)
        res.append((y.detach(), xr.grad, br.grad))
    (y0, dx0, db0), (y1, dx1, db1) = res
    np.testing.assert_allclose(y0.cpu().numpy(), y1.cpu().numpy(), rtol=1e-5, atol=1e-5)
    # 'linear' saves neither input nor output (spec.ref == ''), so its native gradient cannot zero the saturated elements
    # the way aten's clamp does -- upstream's CUDA path behaves the same (bias_act.cu:139-141 reads yref = 0): compare the
    # unsaturated elements there, everything otherwise
    free = torch.ones_like(y0, dtype=torch.bool) if (act != 'linear' or clamp is None) else (y0.abs() < clamp)
    np.testing.assert_allclose(dx0[free].cpu().numpy(), dx1[free].cpu().numpy(), rtol=1e-5, atol=1e-5)
    if bool(free.all()):
        np.testing.assert_allclose(db0.cpu().numpy(), db1.cpu().numpy(), rtol=1e-4, atol=1e-4)
    else:
        np.testing.assert_allclose(dx0[~free].cpu().numpy(), (dy * (gain or 1.0))[~free].cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_bias_act_full_size_properties(native):
    B = op('bias_act')
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn([16, 64, 256, 256], device=DEV, generator=g)
    b = torch.randn([64], device=DEV, generator=g)
    y = B.bias_act(x, b, act='lrelu', clamp=256)
    ref = torch.nn.functional.leaky_relu(x + b.reshape(1, -1, 1, 1), 0.2) * np.sqrt(2)
    assert float((y - ref.clamp(-256, 256)).abs().max()) < 1e-5
    # idempotence of the clamp stage and oddness of lrelu-with-alpha-1
    z = B.bias_act(x, None, act='lrelu', alpha=1.0, gain=1.0)
    assert torch.equal(z, x)


# --------------------------------------------------------------------------------------------
# conv2d_resample / modulated conv building blocks on the GPU (aten convs + native resampling)

@pytest.mark.parametrize('gradfix', [False, True])
def test_conv2d_resample_and_modulated_conv2d_golden_on_gpu(native, gradfix):
    """All 26 reference vectors of tests/golden/conv.npz on the device: the six conv2d_resample paths, the 16
    modulated_conv2d cases (fused / non-fused, demodulation, noise, up 1/2: y, dx, dw, dstyles) and fma."""
    cases, arr = load_golden('conv')
    before = native.launch_counts()
    for c in cases:
        out = run_conv_case(c, arr, device=DEV, gradfix=gradfix)
        assert out is not None, c
        for k, v in out.items():
            np.testing.assert_allclose(v.detach().cpu().numpy(), arr[f'c{c["id"]}_{k}'], rtol=1e-3, atol=1e-3, err_msg=f'{c} {k}')
    after = native.launch_counts()
    assert after['demod'] > before['demod']        # the native demodulation coefficients were part of it


# --------------------------------------------------------------------------------------------
# fused epilogue, filtered_lrelu

@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_modconv_epilogue_matches_fma_then_bias_act(native, dtype):
    E, B, F = op('modconv_epilogue'), op('bias_act'), op('fma')
    g = torch.Generator(device=DEV).manual_seed(2)
    for shape in ([2, 6, 16, 16], [3, 5, 7, 9]):
        n, c, h, w = shape
        x = torch.randn(shape, device=DEV, generator=g).to(dtype).requires_grad_(True)
        d = (torch.rand([n, c], device=DEV, generator=g) + 0.5).requires_grad_(True)
        nz = (torch.randn([h, w], device=DEV, generator=g) * 0.1).requires_grad_(True)
        b = torch.randn([c], device=DEV, generator=g).requires_grad_(True)
        before = native.launch_count(native.K_FMA_BIAS_ACT)
        y = E.modconv_epilogue(x, d, nz, b, act='lrelu', clamp=256)
        assert native.launch_count(native.K_FMA_BIAS_ACT) == before + 1
        t = F.fma(x, d.to(dtype).reshape(n, c, 1, 1), nz.to(dtype))
        y2 = B.bias_act(t, b.to(dtype), act='lrelu', clamp=256)
        close_lowp(y.float().detach().cpu().numpy(), y2.float().detach().cpu().numpy(), dtype, 'fwd')
        dy = torch.randn(shape, device=DEV, generator=g).to(dtype)
        g1 = torch.autograd.grad([y], [x, d, nz, b], [dy])
        g2 = torch.autograd.grad([y2], [x, d, nz, b], [dy])
        for u, v, nm in zip(g1, g2, 'x d noise b'.split()):
            close_lowp(u.float().cpu().numpy(), v.float().cpu().numpy(), dtype, f'grad {nm}')


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_blur_epilogue_one_pass_matches_blur_then_epilogue(native, dtype):
    """modconv_epilogue.blur_epilogue (mg_upfirdn2d_epilogue: the 4x4 blur that ends an up-sampling layer + the layer's epilogue in
    one pass; backward through mg_bwd_fused_nhwc_y with the blurred tensor rebuilt from the saved output) against upfirdn2d followed
    by modconv_epilogue.  Integer data: bit for bit forward (both FIR kernels: the block form and, from 192 output rows on, the
    row-streaming form); random data: forward within two ulps, gradients within the bounds of the epilogue test -- the demodulation
    gradient additionally carries the rounding of y (it is summed over the REBUILT filter output)."""
    E, U = op('modconv_epilogue'), op('upfirdn2d')
    g = torch.Generator(device=DEV).manual_seed(9)
    f = U.setup_filter([1, 3, 3, 1]).to(DEV)
    for (n, c, h, w, full) in [(2, 64, 33, 33, True), (3, 16, 17, 41, True), (1, 64, 257, 129, True), (2, 128, 65, 65, False), (1, 32, 200, 24, False)]:
        for exact in (True, False):
            oh, ow = h - 1, w - 1
            if exact:       # the filter [1,3,3,1]^2 / 64 * gain 4 = integers / 16: products and sums stay exact in both 16-bit types
                x = (torch.randint(-2, 3, [n, c, h, w], device=DEV, generator=g) * 16).to(dtype)
                d = torch.randint(1, 3, [n, c], device=DEV, generator=g).float()
                nz = torch.randint(-2, 3, [n, 1, oh, ow], device=DEV, generator=g).float()
                st = torch.tensor(2.0, device=DEV)
                b = torch.randint(-3, 4, [c], device=DEV, generator=g).float()
            else:
                x = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype)
                d = torch.rand([n, c], device=DEV, generator=g) + 0.5
                nz = torch.randn([n, 1, oh, ow], device=DEV, generator=g)
                st = torch.tensor(0.3, device=DEV)
                b = torch.randn([c], device=DEV, generator=g) * 0.3
            x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
            args = dict(dcoefs=d.requires_grad_(True), noise=nz, b=b.requires_grad_(True), noise_strength=st.requires_grad_(True)) if full else dict(b=b.requires_grad_(True))
            leaves = [x] + [v for v in args.values() if v.requires_grad]
            kw = dict(act='lrelu', gain=float(np.sqrt(2)) if not exact else 2.0, clamp=None if not exact else 40.0)
            before = native.launch_count(native.K_UPFIRDN2D_NHWC)
            assert E.blur_epilogue_applicable(x, f, [1, 1, 1, 1], 'lrelu')
            y = E.blur_epilogue(x, f, [1, 1, 1, 1], fgain=4.0, **args, **kw)
            assert native.launch_count(native.K_UPFIRDN2D_NHWC) == before + 1 and tuple(y.shape) == (n, c, oh, ow)
            ref = E.modconv_epilogue(U.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4.0), args.get('dcoefs'), args.get('noise'), args['b'],
                                     noise_strength=args.get('noise_strength'), **kw)
            what = f'{(n, c, h, w, full)} exact={exact}'
            if exact:
                assert torch.equal(y, ref), (what, float((y.float() - ref.float()).abs().max()))
            else:
                close_lowp(y.float().detach().cpu().numpy(), ref.float().detach().cpu().numpy(), dtype, what)
            dy = (torch.randint(-2, 3, y.shape, device=DEV, generator=g) if exact else torch.randn(y.shape, device=DEV, generator=g)).to(dtype)
            dy = dy.contiguous(memory_format=torch.channels_last)
            g1, g2 = torch.autograd.grad([y], leaves, [dy]), torch.autograd.grad([ref], leaves, [dy])
            for u, v, leaf in zip(g1, g2, leaves):
                nm = 'x' if leaf is x else 'd' if (full and leaf is args['dcoefs']) else 'st' if (full and leaf is args.get('noise_strength')) else 'b'
                if exact and nm != 'd':
                    assert torch.equal(u.float(), v.float()), f'{what}: grad {nm}'
                else:
                    # d: summed over the filter output rebuilt from y -- exact only where y was not rounded
                    close_lowp(u.float().cpu().numpy(), v.float().cpu().numpy(), dtype, f'{what}: grad {nm}', tol=(8 if nm == 'd' else 4) * LOWP_TOL[dtype])
    # second order (create_graph): the differentiable composition inside the same Function
    x = (torch.randn([2, 16, 17, 17], device=DEV, generator=g) * 0.25).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    d = (torch.rand([2, 16], device=DEV, generator=g) + 0.5).requires_grad_(True)
    b = torch.zeros([16], device=DEV, requires_grad=True)
    outs = []
    for fused in (True, False):
        E.blur_fusion = fused
        try:
            y = E.blur_epilogue(x, f, [1, 1, 1, 1], fgain=4.0, dcoefs=d, b=b, act='lrelu', clamp=256)
            gx, = torch.autograd.grad([y.float().square().mean()], [x], create_graph=True)       # (small magnitudes: float16 gradients)
            outs.append(torch.autograd.grad([gx.float().square().sum()], [x, d]))
        finally:
            E.blur_fusion = True
    for u, v in zip(*outs):
        close_lowp(u.float().cpu().numpy(), v.float().cpu().numpy(), dtype, 'second order', tol=8 * LOWP_TOL[dtype])


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_torgb_one_pass_matches_the_four_launches(native, dtype):
    """ops/torgb.py: ToRGB + the cast to the float32 image path as one native pass against the composition it replaces
    (channel_scale -> 1x1 convolution -> bias_act(clamp) -> .to(float32)): values, all four first-order gradients, and the
    gradient of a gradient (create_graph=True runs the differentiable composition inside the same Function).  Integer-valued data
    makes every product and sum exact in both, so those cases must agree BIT FOR BIT (rounding points included); random data
    within two ulps.  Ragged pixel counts, 3 and 4 image channels, a clamp that bites."""
    TG = op('torgb')
    g = torch.Generator(device=DEV).manual_seed(5)
    for (n, c, h, w, o, clamp) in [(2, 64, 24, 40, 4, 6.0), (3, 128, 9, 7, 3, None), (2, 256, 16, 16, 4, 2.5), (1, 512, 8, 8, 4, 4.0), (5, 64, 33, 31, 4, 3.0)]:
        for exact in (True, False):
            if exact:       # small integers: |x*s| <= 12, |sum| <= 12 * 3 * 512 < 2^15, every value representable in both 16-bit types
                x = torch.randint(-3, 4, [n, c, h, w], device=DEV, generator=g).to(dtype)
                wt = torch.randint(-3, 4, [o, c, 1, 1], device=DEV, generator=g).to(dtype)
                st = torch.randint(-2, 3, [n, c], device=DEV, generator=g).float()
                b = torch.randint(-3, 4, [o], device=DEV, generator=g).to(dtype)
                wt = wt * (torch.rand([o, c, 1, 1], device=DEV, generator=g) < 8.0 / c).to(dtype)      # sparse rows keep the sums small
            else:
                x = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype)
                wt = (torch.randn([o, c, 1, 1], device=DEV, generator=g) / c ** 0.5).to(dtype)
                st = torch.randn([n, c], device=DEV, generator=g) + 1
                b = torch.randn([o], device=DEV, generator=g).to(dtype)
            x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wt, st, b = wt.requires_grad_(True), st.requires_grad_(True), b.requires_grad_(True)
            gy = torch.randint(-4, 5, [n, o, h, w], device=DEV, generator=g).float() if exact else torch.randn([n, o, h, w], device=DEV, generator=g)
            before = native.launch_count(native.K_TORGB)
            y = TG.torgb(x, wt, st, b, clamp=clamp)
            g1 = torch.autograd.grad([y], [x, wt, st, b], [gy])
            assert native.launch_count(native.K_TORGB) == before + 2, 'the native forward and backward passes did not run'
            assert y.dtype == torch.float32 and y.is_contiguous() and g1[0].is_contiguous(memory_format=torch.channels_last)
            TG.enabled = False
            try:
                y2 = TG.torgb(x, wt, st, b, clamp=clamp)
                g2 = torch.autograd.grad([y2], [x, wt, st, b], [gy])
            finally:
                TG.enabled = True
            assert native.launch_count(native.K_TORGB) == before + 2
            what = f'{(n, c, h, w, o, clamp)} exact={exact}'
            if exact:
                assert torch.equal(y, y2), what
                for u, v, nm in zip(g1, g2, 'x w s b'.split()):
                    # dW / ds of the composition are rounded through a 16-bit dxs / xs; on integer data those are exact too
                    assert torch.equal(u.float(), v.float()), f'{what}: grad {nm}'
            else:
                close_lowp(y.detach().cpu().numpy(), y2.detach().cpu().numpy(), dtype, what + ' fwd')
                for u, v, nm in zip(g1, g2, 'x w s b'.split()):
                    # sums of ~h*w*n rounded terms: the two sides round different intermediates (documented in ops/torgb.py)
                    close_lowp(u.float().cpu().numpy(), v.float().cpu().numpy(), dtype, f'{what}: grad {nm}', tol=4 * LOWP_TOL[dtype])
    # second order (path-length regularisation shape): d/dx and d/ds of |dy/ds|^2 through both forms
    n, c, h, w, o = 2, 64, 8, 8, 4
    x = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn([o, c, 1, 1], device=DEV, generator=g) / 8).to(dtype).requires_grad_(True)
    st = (torch.randn([n, c], device=DEV, generator=g) + 1).requires_grad_(True)
    b = torch.zeros([o], device=DEV).to(dtype).requires_grad_(True)
    noise = torch.randn([n, o, h, w], device=DEV, generator=g)
    outs = []
    for flag in (True, False):
        TG.enabled = flag
        try:
            y = TG.torgb(x, wt, st, b, clamp=256)
            gs, gx = torch.autograd.grad([(y * noise).sum()], [st, x], create_graph=True)       # (y is linear in s and in x)
            outs.append(torch.autograd.grad([gs.square().sum() + gx.float().square().sum()], [x, wt, st]))
        finally:
            TG.enabled = True
    for u, v, nm in zip(outs[0], outs[1], 'x w s'.split()):
        close_lowp(u.float().cpu().numpy(), v.float().cpu().numpy(), dtype, f'second order grad {nm}', tol=4 * LOWP_TOL[dtype])


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_fromrgb_one_pass_matches_the_two_launches(native, dtype):
    """ops/fromrgb.py: the discriminator's FromRGB (1x1 convolution from <= 4 channels + bias_act) as one native pass each way against
    the two launches it replaces: values, the three first-order gradients (with and without a gradient for the image), and the R1 shape
    of a second differentiation (create_graph=True runs the composition's own backward inside the same Function).  Integer-valued
    data with a gain of 1 makes every product and sum exact on both sides, so those cases must agree BIT FOR BIT; random data within
    the tolerance of the 16-bit type.  Ragged pixel counts, 3 and 4 image channels, a clamp that bites, lrelu and linear."""
    FR = op('fromrgb')
    g = torch.Generator(device=DEV).manual_seed(6)
    for (n, i, c, h, w, act, clamp) in [(2, 4, 64, 24, 40, 'lrelu', 6.0), (3, 3, 128, 9, 7, 'lrelu', None), (2, 4, 256, 16, 16, 'linear', 2.5),
                                        (1, 4, 512, 8, 8, 'lrelu', 4.0), (5, 4, 64, 33, 31, 'lrelu', 3.0), (2, 1, 64, 5, 5, 'lrelu', None)]:
        for exact in (True, False):
            if exact:       # small integers and alpha = 0.5, gain = 1: every intermediate representable in both 16-bit types
                x = torch.randint(-3, 4, [n, i, h, w], device=DEV, generator=g).to(dtype)
                wt = torch.randint(-3, 4, [c, i, 1, 1], device=DEV, generator=g).to(dtype)
                b = torch.randint(-3, 4, [c], device=DEV, generator=g).to(dtype)
                gy = torch.randint(-4, 5, [n, c, h, w], device=DEV, generator=g).to(dtype)
                kw = dict(act=act, alpha=0.5, gain=1.0, clamp=clamp)
            else:
                x = torch.randn([n, i, h, w], device=DEV, generator=g).to(dtype)
                wt = (torch.randn([c, i, 1, 1], device=DEV, generator=g) / i ** 0.5).to(dtype)
                b = torch.randn([c], device=DEV, generator=g).to(dtype)
                gy = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype)
                kw = dict(act=act, clamp=clamp)
            gy = gy.contiguous(memory_format=torch.channels_last)
            for image_grad in (True, False):
                xx = x.contiguous(memory_format=torch.channels_last).requires_grad_(image_grad)
                ww, bb = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
                leaves = ([xx] if image_grad else []) + [ww, bb]
                before = native.launch_count(native.K_FROMRGB)
                y = FR.fromrgb(xx, ww, bb, **kw)
                g1 = torch.autograd.grad([y], leaves, [gy])
                assert native.launch_count(native.K_FROMRGB) == before + 2, 'the native forward and backward passes did not run'
                assert y.is_contiguous(memory_format=torch.channels_last) and y.dtype == dtype
                FR.enabled = False
                try:
                    y2 = FR.fromrgb(xx, ww, bb, **kw)
                    g2 = torch.autograd.grad([y2], leaves, [gy])
                finally:
                    FR.enabled = True
                assert native.launch_count(native.K_FROMRGB) == before + 2
                what = f'{(n, i, c, h, w, act, clamp)} exact={exact} image_grad={image_grad}'
                names = (['x'] if image_grad else []) + ['w', 'b']
                if exact:
                    assert torch.equal(y, y2), what
                    for u, v, nm in zip(g1, g2, names):
                        assert torch.equal(u.float(), v.float()), f'{what}: grad {nm}'
                else:
                    close_lowp(y.detach().float().cpu().numpy(), y2.detach().float().cpu().numpy(), dtype, what + ' fwd')
                    for u, v, nm in zip(g1, g2, names):
                        # dW / db: sums of n*h*w terms, rounded to the 16-bit type once here and once by the library / the reduction
                        close_lowp(u.float().cpu().numpy(), v.float().cpu().numpy(), dtype, f'{what}: grad {nm}', tol=4 * LOWP_TOL[dtype])
    # second order, the R1 shape: d/dw, d/db of |d sum(y) / dx|^2 through both forms
    n, i, c, h, w = 2, 4, 64, 8, 8
    x = torch.randn([n, i, h, w], device=DEV, generator=g).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn([c, i, 1, 1], device=DEV, generator=g) / 2).to(dtype).requires_grad_(True)
    b = (torch.randn([c], device=DEV, generator=g) / 4).to(dtype).requires_grad_(True)
    mix = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype)
    outs = []
    for flag in (True, False):
        FR.enabled = flag
        try:
            y = FR.fromrgb(x, wt, b, act='lrelu', clamp=256)
            gx, = torch.autograd.grad([(y.float() * mix.float()).sum()], [x], create_graph=True)
            outs.append(torch.autograd.grad([gx.float().square().sum()], [wt]))
        finally:
            FR.enabled = True
    for u, v, nm in zip(outs[0], outs[1], ['w']):
        close_lowp(u.float().cpu().numpy(), v.float().cpu().numpy(), dtype, f'second order grad {nm}', tol=4 * LOWP_TOL[dtype])


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_residual_tail_one_pass_matches_the_three_launches(native, dtype):
    """ops/residual_bias_act.py: bias_act(y_raw, linear, gain) + bias_act(x_raw, b, lrelu, gain, clamp) + add -- the tail of the
    discriminator's residual block -- as one native pass each way against the three launches.  The arithmetic is element-wise with the
    same rounding points, so values and dx / dy must agree BIT FOR BIT on any data; the bias gradient is a float32 sum in another order
    (exact on integer data).  Ragged pixel counts, a clamp that bites, no bias, the linear activation, and the R1 shape of a second
    differentiation."""
    RS = op('residual_bias_act')
    g = torch.Generator(device=DEV).manual_seed(9)
    cl = torch.channels_last
    for (n, c, h, w, act, clamp, has_b) in [(2, 64, 24, 40, 'lrelu', 1.5, True), (3, 128, 9, 7, 'lrelu', None, True), (2, 256, 5, 5, 'linear', 0.75, True),
                                            (1, 512, 3, 3, 'lrelu', 2.0, False), (5, 64, 33, 31, 'lrelu', 181.02, True)]:
        for exact in (True, False):
            if exact:
                x = torch.randint(-6, 7, [n, c, h, w], device=DEV, generator=g).to(dtype)
                y = torch.randint(-6, 7, [n, c, h, w], device=DEV, generator=g).to(dtype)
                b = torch.randint(-3, 4, [c], device=DEV, generator=g).to(dtype) if has_b else None
                gy = torch.randint(-4, 5, [n, c, h, w], device=DEV, generator=g).to(dtype)
                kw = dict(act=act, alpha=0.5, gain=2.0, clamp=clamp, r_gain=0.5)
            else:
                x = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype)
                y = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype)
                b = torch.randn([c], device=DEV, generator=g).to(dtype) if has_b else None
                gy = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype)
                kw = dict(act=act, gain=float(np.sqrt(2) * np.sqrt(0.5)), clamp=clamp, r_gain=float(np.sqrt(0.5)))
            x = x.contiguous(memory_format=cl).requires_grad_(True)
            y = y.contiguous(memory_format=cl).requires_grad_(True)
            gy = gy.contiguous(memory_format=cl)
            leaves = [x, y] + ([b.requires_grad_(True)] if has_b else [])
            before = native.launch_count(native.K_RESIDUAL)
            out = RS.residual_bias_act(x, b, y, **kw)
            g1 = torch.autograd.grad([out], leaves, [gy])
            assert native.launch_count(native.K_RESIDUAL) == before + 2, 'the native forward and backward passes did not run'
            assert out.is_contiguous(memory_format=cl) and g1[0].is_contiguous(memory_format=cl)
            RS.enabled = False
            try:
                out2 = RS.residual_bias_act(x, b, y, **kw)
                g2 = torch.autograd.grad([out2], leaves, [gy])
            finally:
                RS.enabled = True
            assert native.launch_count(native.K_RESIDUAL) == before + 2
            what = f'{(n, c, h, w, act, clamp, has_b)} exact={exact}'
            assert torch.equal(out, out2), what
            assert torch.equal(g1[0], g2[0]), what + ': grad x'
            assert torch.equal(g1[1], g2[1]), what + ': grad y'
            if has_b:
                if exact:
                    assert torch.equal(g1[2].float(), g2[2].float()), what + ': grad b'
                else:
                    close_lowp(g1[2].float().cpu().numpy(), g2[2].float().cpu().numpy(), dtype, what + ': grad b', tol=4 * LOWP_TOL[dtype])
    # second order, the R1 shape: the first gradients under create_graph=True, then the derivative of their squared norm w.r.t. a leaf the
    # incoming gradient depends on (the activation's slope is piecewise constant: what has to work is that the graph through dout is connected)
    n, c, h, w = 2, 64, 8, 8
    x = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype).contiguous(memory_format=cl).requires_grad_(True)
    y = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype).contiguous(memory_format=cl).requires_grad_(True)
    b = (torch.randn([c], device=DEV, generator=g) / 4).to(dtype).requires_grad_(True)
    mix = torch.randn([n, c, h, w], device=DEV, generator=g).to(dtype).contiguous(memory_format=cl)
    outs = []
    for flag in (True, False):
        RS.enabled = flag
        try:
            scale = torch.full([], 0.75, device=DEV, requires_grad=True)
            out = RS.residual_bias_act(x * 1.0, b, y * 1.0, act='lrelu', clamp=256, r_gain=0.7)
            gx, gy2 = torch.autograd.grad([(out.float() * mix.float() * scale).sum()], [x, y], create_graph=True)
            second, = torch.autograd.grad([gx.float().square().sum() + gy2.float().square().sum()], [scale])
            outs.append((gx.detach().clone(), gy2.detach().clone(), second.detach().clone()))
        finally:
            RS.enabled = True
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    np.testing.assert_allclose(float(outs[0][2]), float(outs[1][2]), rtol=2e-2)
    assert float(outs[0][2]) != 0.0


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_filtered_lrelu_matches_composition(native, dtype):
    FL, U = op('filtered_lrelu'), op('upfirdn2d')
    g = torch.Generator(device=DEV).manual_seed(3)
    f4 = U.setup_filter([1, 3, 3, 1]).to(DEV)
    f12 = U.setup_filter(list(np.hanning(14)[1:-1])).to(DEV)
    for shape, up, down, fu, fd, pad in [([2, 3, 32, 32], 2, 2, f4, f4, [2, 1, 2, 1]), ([1, 4, 40, 24], 2, 1, f12, None, 6),
                                         ([2, 2, 50, 70], 1, 2, None, f12, 0), ([2, 3, 33, 35], 2, 2, f12, f12, [8, 7, 8, 7])]:
        x = torch.randn(shape, device=DEV, generator=g).to(dtype)
        b = torch.randn([shape[1]], device=DEV, generator=g).to(dtype)
        before = native.launch_count(native.K_FILTERED_LRELU)
        with torch.no_grad():
            y = FL.filtered_lrelu(x, fu, fd, b, up=up, down=down, padding=pad, clamp=2.0)
        assert native.launch_count(native.K_FILTERED_LRELU) == before + 1
        ref = FL.filtered_lrelu(x, fu, fd, b, up=up, down=down, padding=pad, clamp=2.0, impl='ref')
        assert y.shape == ref.shape
        close_lowp(y.float().cpu().numpy(), ref.float().cpu().numpy(), dtype, str((shape, up, down)))


# --------------------------------------------------------------------------------------------
# alpha composite, translate sample

def test_alpha_composite_golden_and_oracle(native, oracle):
    A = op('alpha_composite')
    cases, arr = load_golden('montage')
    x = T(arr['png_layers_u8'].astype(np.float32) / 255.0).to(DEV)
    np.testing.assert_allclose(A.alpha_composite(x).cpu().numpy(), arr['png_out'], rtol=1e-5, atol=1e-6)
    for c in cases:
        if c['op'] != 'alpha_composite' or c['src'] != 'rand':
            continue
        i = c['id']
        x = T(arr[f'c{i}_x']).to(DEV).requires_grad_(True)
        y = A.alpha_composite(x)
        dx, = torch.autograd.grad([y], [x], [T(arr[f'c{i}_dy']).to(DEV)])
        np.testing.assert_allclose(y.detach().cpu().numpy(), arr[f'c{i}_y'], rtol=1e-5, atol=1e-6, err_msg=str(c))
        ref = arr[f'c{i}_dx']
        got = dx.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref)), c
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=2e-4, equal_nan=True, err_msg=str(c))
        y2 = A.alpha_composite(x, safe_grad=True)
        dx2, = torch.autograd.grad([y2], [x], [T(arr[f'c{i}_dy']).to(DEV)])
        assert bool(torch.isfinite(dx2).all())
    # odd pixel count -> scalar path; half precision
    rng = np.random.RandomState(4)
    x = rng.rand(2, 5, 4, 7, 9).astype(np.float32)
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        xt = T(x).to(dtype)
        want = oracle.alpha_composite(xt.float().numpy())
        close_lowp(A.alpha_composite(xt.to(DEV)).float().cpu().numpy(), want, dtype, 'composite')


def test_alpha_composite_full_size_properties(native):
    A = op('alpha_composite')
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand([4, 8, 4, 256, 256], device=DEV, generator=g)
    y = A.alpha_composite(x)
    assert y.shape == (4, 4, 256, 256)
    # an opaque top layer hides everything below; a transparent one changes nothing
    top = x.clone(); top[:, -1, 3] = 1.0
    assert torch.allclose(A.alpha_composite(top)[:, :3], top[:, -1, :3], atol=1e-6)
    clear = x.clone(); clear[:, -1, 3] = 0.0
    assert torch.allclose(A.alpha_composite(clear), A.alpha_composite(x[:, :-1]), atol=1e-6)
    assert torch.allclose(y, A.alpha_composite(x, impl='ref'), atol=1e-5)


def test_translate_sample_golden_and_oracle(native, oracle):
    TS = op('translate_sample')
    cases, arr = load_golden('montage')
    for c in cases:
        if c['op'] != 'translate_sample':
            continue
        i = c['id']
        b, l, ch, h, w = c['shape']
        x = T(arr[f'c{i}_x']).reshape(b * l, ch, h, w).to(DEV).requires_grad_(True)
        t = T(arr[f'c{i}_t']).reshape(b * l, 2).to(DEV).requires_grad_(True)
        y = TS.translate_sample(x, t, shift=c['shift'])
        dx, dt = torch.autograd.grad([y], [x, t], [T(arr[f'c{i}_dy']).to(DEV)])
        np.testing.assert_allclose(y.detach().cpu().numpy(), arr[f'c{i}_y'], rtol=1e-4, atol=1e-4, err_msg=str(c))
        np.testing.assert_allclose(dx.cpu().numpy(), arr[f'c{i}_dx'].reshape(x.shape), rtol=1e-4, atol=1e-4, err_msg=str(c))
        np.testing.assert_allclose(dt.cpu().numpy(), arr[f'c{i}_dt'].reshape(-1, 2), rtol=2e-3, atol=2e-3, err_msg=str(c))
    # integer-pixel and half-pixel shifts (exact texel boundaries), out-of-frame, generic
    rng = np.random.RandomState(6)
    x = rng.randn(8, 4, 32, 48).astype(np.float32)
    t = np.asarray([[0, 0], [2 / 48, -4 / 32], [1 / 48, 1 / 32], [-0.5, 0.25], [2.5, 0], [0.3333, -0.7777],
                    [-0.1234, 0.0577], [0.9, 0.41]], dtype=np.float32)
    # d/dt is discontinuous where a sample lands exactly on a texel (left and right slopes differ and
    # the float rounding of the coordinate picks one per pixel -- in the reference too), so it is
    # compared only for shifts that are not a whole number of pixels; y and dx are continuous.
    smooth = [2, 5, 6, 7]
    dy = rng.randn(*x.shape).astype(np.float32)
    for shift in (0.0, 1.0):
        xt, tt = T(x).to(DEV).requires_grad_(True), T(t).to(DEV).requires_grad_(True)
        y = TS.translate_sample(xt, tt, shift=shift)
        gx, gt = torch.autograd.grad([y], [xt, tt], [T(dy).to(DEV)])
        np.testing.assert_allclose(y.detach().cpu().numpy(), oracle.translate_sample(x, t, shift), rtol=1e-5, atol=1e-5)
        odx, odt = oracle.translate_sample_bwd(x, dy, t, shift)
        np.testing.assert_allclose(gx.cpu().numpy(), odx, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gt.cpu().numpy()[smooth], odt[smooth], rtol=1e-3, atol=1e-2)
        assert np.isfinite(gt.cpu().numpy()).all()


def test_translate_sample_full_size_properties(native):
    TS = op('translate_sample')
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn([32, 4, 256, 256], device=DEV, generator=g)
    zero = torch.zeros([32, 2], device=DEV)
    # identity up to the float32 rounding of the sampling coordinate (~2.5e-5 px at 256) times the local slope
    assert torch.allclose(TS.translate_sample(x, zero, shift=1.0), x, atol=1e-3)
    t = (torch.rand([32, 2], device=DEV, generator=g) - 0.5)
    y = TS.translate_sample(x, t, shift=1.0)
    assert torch.allclose(y, TS.translate_sample(x, t, shift=1.0, impl='ref'), atol=1e-4)
    # adjoint: <S x, w> == <x, S^T w>
    xr = x.clone().requires_grad_(True)
    w = torch.randn(x.shape, device=DEV, generator=g)
    gx, = torch.autograd.grad([TS.translate_sample(xr, t)], [xr], [w])
    lhs = float((TS.translate_sample(x, t).double() * w.double()).sum())
    rhs = float((x.double() * gx.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_affine_warp_golden_oracle_and_adjointness(native, oracle):
    """ADA's resampling step as one native pass (csrc/augment.hip): the reference's golden outputs and input
    gradients, the CPU oracle on rotations / anisotropic scales / far translations, the gather-form adjoint against
    the scatter-form oracle, second-order use (backward of backward = forward), and -- at the bench shape -- the
    size-independent adjointness property <W x, g> = <x, W^T g>."""
    W = op('affine_warp')
    cases, arr = load_golden('warp')
    k0 = native.launch_count(native.K_AFFINE_WARP_FWD), native.launch_count(native.K_AFFINE_WARP_BWD)
    nw = 0
    for c in cases:
        if c['kind'] != 'warp':
            continue
        i = c['id']
        x = T(arr[f'w{i}_x']).to(DEV).requires_grad_(True)
        y = W.affine_warp(x, T(arr[f'w{i}_theta']).to(DEV), *c['out'])
        dx, = torch.autograd.grad([y], [x], [T(arr[f'w{i}_gy']).to(DEV)])
        nw += 1
        np.testing.assert_allclose(y.detach().cpu().numpy(), arr[f'w{i}_y'], rtol=1e-4, atol=3e-5, err_msg=str(c))
        np.testing.assert_allclose(dx.cpu().numpy(), arr[f'w{i}_dx'], rtol=1e-4, atol=6e-5, err_msg=str(c))
    assert native.launch_count(native.K_AFFINE_WARP_FWD) == k0[0] + nw and native.launch_count(native.K_AFFINE_WARP_BWD) == k0[1] + nw
    rng = np.random.RandomState(4)
    for (n, ch, hi, wi, ho, wo) in [(3, 4, 61, 47, 50, 70), (2, 3, 33, 33, 33, 33), (2, 1, 8, 100, 40, 9)]:
        x = rng.randn(n, ch, hi, wi).astype(np.float32)
        gy = rng.randn(n, ch, ho, wo).astype(np.float32)
        ang = rng.uniform(-np.pi, np.pi, n)
        sx, sy = np.exp2(rng.uniform(-1.2, 1.2, n)), np.exp2(rng.uniform(-1.2, 1.2, n))
        th = np.stack([np.stack([sx * np.cos(ang), -sy * np.sin(ang), rng.uniform(-1.5, 1.5, n)], 1),
                       np.stack([sx * np.sin(ang), sy * np.cos(ang), rng.uniform(-1.5, 1.5, n)], 1)], 1).astype(np.float32)
        xg = T(x).to(DEV).requires_grad_(True)
        y = W.affine_warp(xg, T(th).to(DEV), ho, wo)
        dx, = torch.autograd.grad([y], [xg], [T(gy).to(DEV)], create_graph=True)
        np.testing.assert_allclose(y.detach().cpu().numpy(), oracle.affine_warp(x, th, ho, wo), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(dx.detach().cpu().numpy(), oracle.affine_warp_bwd(gy, th, hi, wi), rtol=1e-4, atol=2e-4)
        # second order: d/d(gy) of <dx, v> is W v
        v = torch.randn_like(dx)
        gyt = T(gy).to(DEV).requires_grad_(True)
        dx2, = torch.autograd.grad([W.affine_warp(xg, T(th).to(DEV), ho, wo)], [xg], [gyt], create_graph=True)
        back, = torch.autograd.grad([(dx2 * v).sum()], [gyt])
        np.testing.assert_allclose(back.cpu().numpy(), oracle.affine_warp(v.cpu().numpy(), th, ho, wo), rtol=1e-4, atol=1e-4)
        # half precision storage
        for dt in (torch.float16, torch.bfloat16):
            yl = W.affine_warp(T(x).to(dt).to(DEV), T(th).to(DEV), ho, wo)
            close_lowp(yl.float().cpu().numpy(), oracle.affine_warp(T(x).to(dt).float().numpy(), th, ho, wo), dt, 'warp lowp')
    # bench shape: <W x, g> == <x, W^T g>
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn([8, 4, 620, 620], device=DEV, generator=g)
    gy = torch.randn([8, 4, 548, 548], device=DEV, generator=g)
    a = torch.linspace(-0.6, 0.6, 8, device=DEV)
    th = torch.stack([torch.stack([1.1 * a.cos(), -0.9 * a.sin(), 0.05 * a], 1), torch.stack([1.1 * a.sin(), 0.9 * a.cos(), -0.1 * a], 1)], 1)
    xr = x.clone().requires_grad_(True)
    y = W.affine_warp(xr, th, 548, 548)
    dx, = torch.autograd.grad([y], [xr], [gy])
    lhs, rhs = float((y.detach().double() * gy.double()).sum()), float((x.double() * dx.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0) + 1e-2, (lhs, rhs)
    ref = W.affine_warp(x, th, 548, 548, impl='ref')          # aten on the device
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.cpu().numpy(), rtol=1e-3, atol=1e-3)


def test_channel_mix_golden_and_oracle(native, oracle):
    M = op('channel_mix')
    cases, arr = load_golden('warp')
    k0 = native.launch_count(native.K_CHANNEL_MIX)
    for c in cases:
        if c['kind'] != 'mix':
            continue
        i, ch = c['id'], c['channels']
        x, C = T(arr[f'm{i}_x']).to(DEV).requires_grad_(True), T(arr[f'm{i}_C']).to(DEV)
        A = torch.eye(ch, device=DEV).repeat(x.shape[0], 1, 1)
        A[:, :3, :3] = C[:, :3, :3]
        t = torch.nn.functional.pad(C[:, :3, 3], [0, ch - 3])
        y = M.channel_mix(x, A, t)
        dx, = torch.autograd.grad([y], [x], [T(arr[f'm{i}_gy']).to(DEV)])
        np.testing.assert_allclose(y.detach().cpu().numpy(), arr[f'm{i}_y'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dx.cpu().numpy(), arr[f'm{i}_dx'], rtol=1e-5, atol=1e-5)
    assert native.launch_count(native.K_CHANNEL_MIX) == k0 + 4
    rng = np.random.RandomState(6)
    for shape, dt in [([5, 4, 256, 256], torch.float32), ([2, 3, 7, 9], torch.float32), ([3, 1, 33], torch.float32),
                      ([2, 4, 64, 64], torch.float16), ([2, 2, 17], torch.bfloat16)]:
        x = T(rng.randn(*shape).astype(np.float32)).to(dt)
        A = rng.randn(shape[0], shape[1], shape[1]).astype(np.float32)
        t = rng.randn(shape[0], shape[1]).astype(np.float32)
        got = M.channel_mix(x.to(DEV), T(A).to(DEV), T(t).to(DEV))
        assert got.dtype == dt and got.shape == x.shape
        close_lowp(got.float().cpu().numpy(), oracle.channel_mix(x.float().numpy(), A, t), dt, str(shape))


def test_demod_coefs_native_vs_oracle(native, oracle):
    """csrc/demod.hip: demodulation coefficients and both gradients against the float64 oracle (the reference's
    definition on per-sample weights), the double backward against the composite expression."""
    D = op('demod')
    g = torch.Generator().manual_seed(9)
    k0 = native.launch_count(native.K_DEMOD)
    for ci, (oc, ic, k, n) in enumerate([(64, 128, 3, 32), (512, 512, 3, 32), (4, 64, 1, 7), (37, 100, 3, 5), (256, 512, 1, 1), (64, 300, 3, 9),
                                         (128, 64, 3, 32), (16, 8, 5, 3)]):
        w = torch.randn([oc, ic, k, k], generator=g).to(DEV)
        if ci >= 5:             # channels-last parameters (the reduced-precision blocks): read and written in place
            w = w.contiguous(memory_format=torch.channels_last)
        w.requires_grad_(True)
        s = (torch.randn([n, ic], generator=g) + 1).to(DEV).requires_grad_(True)
        d = D.demod_coefs(w, s)
        gy = torch.randn([n, oc], generator=g).to(DEV)
        dw, ds = torch.autograd.grad([d], [w, s], [gy])
        wn, sn = w.detach().cpu().numpy(), s.detach().cpu().numpy()
        np.testing.assert_allclose(d.detach().cpu().numpy(), oracle.demod_coefs(wn, sn), rtol=2e-5, atol=1e-7)
        ow, os_ = oracle.demod_coefs_bwd(wn, sn, gy.cpu().numpy())
        np.testing.assert_allclose(dw.cpu().numpy(), ow, rtol=1e-4, atol=1e-5 * np.abs(ow).max())
        np.testing.assert_allclose(ds.cpu().numpy(), os_, rtol=1e-4, atol=1e-5 * np.abs(os_).max())
        assert dw.stride() == w.stride()
    assert native.launch_count(native.K_DEMOD) == k0 + 16           # one forward and one backward call per case
    # second order (path-length regularisation): gradient of a function of d's gradient
    w = torch.randn([16, 24, 3, 3], generator=g).to(DEV).requires_grad_(True)
    s = (torch.randn([4, 24], generator=g) + 1).to(DEV).requires_grad_(True)
    res = []
    for impl in ('cuda', 'ref'):
        d = D.demod_coefs(w, s, impl=impl)
        gs, = torch.autograd.grad([d.square().sum()], [s], create_graph=True)
        gw2, gs2 = torch.autograd.grad([gs.square().sum()], [w, s])
        res.append((gs.detach(), gw2, gs2))
    for a, b in zip(*res):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(b.abs().max()))


def test_profiling_hooks_report_kernel_time(native):
    B = op('bias_act')
    x = torch.randn([8, 64, 128, 128], device=DEV)
    native.prof_reset()
    native.prof_enable(True)
    try:
        for _ in range(5):
            B.bias_act(x, None, act='lrelu')
        torch.cuda.synchronize()
        rec = native.prof_collect()
    finally:
        native.prof_enable(False)
        native.prof_reset()
    assert rec['bias_act']['launches'] == 5
    assert rec['bias_act']['ms'] > 0
    assert rec['bias_act']['bytes'] == 5 * 2 * x.numel() * 4


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_fused_backward_reductions_match_unfused(native, dtype):
    """mg_bwd_fused behind bias_act / channel_scale / modconv_epilogue: dx and the float32-accumulated
    reductions (bias, dcoefs/styles, noise_strength gradients) against the unfused autograd graph."""
    B, E = op('bias_act'), op('modconv_epilogue')
    g = torch.Generator(device=DEV).manual_seed(9)
    # the last two shapes have >= 1024 small planes: the one-wave-per-plane variant (vector and scalar paths)
    for shape in ([3, 6, 16, 16], [2, 5, 7, 9], [4, 8, 64, 64], [8, 160, 8, 8], [16, 128, 3, 5]):
        n, c, h, w = shape
        # (1) bias_act with bias gradient
        x = torch.randn(shape, device=DEV, generator=g).to(dtype).requires_grad_(True)
        b = torch.randn([c], device=DEV, generator=g).to(dtype).requires_grad_(True)
        dy = torch.randn(shape, device=DEV, generator=g).to(dtype)
        before = native.launch_count(native.K_BWD_FUSED)
        y = B.bias_act(x, b, act='lrelu', clamp=1.5)
        gx, gb = torch.autograd.grad([y], [x, b], [dy])
        assert native.launch_count(native.K_BWD_FUSED) == before + 1
        y2 = B.bias_act(x, b, act='lrelu', clamp=1.5, impl='ref')
        rx, rb = torch.autograd.grad([y2], [x, b], [dy])
        # where the rounded output sits exactly ON the clamp bound (common in 16-bit formats) the kernel zeroes the
        # gradient (bias_act.cu:139-141) and aten's clamp() passes it: compare dx away from those points, and db
        # against the sum of the native dx itself
        ok = ((y2.detach().abs().float() < 1.49) & (y.detach().abs().float() < 1.49) & (y2.detach().abs().float() > 0.02)).cpu().numpy()
        close_lowp(gx.float().cpu().numpy()[ok], rx.float().cpu().numpy()[ok], dtype, 'bias_act dx')
        close_lowp(gb.float().cpu().numpy(), gx.float().sum([0, 2, 3]).cpu().numpy(), dtype, 'bias_act db')
        # (2) channel scale
        s = (torch.rand([n, c], device=DEV, generator=g) + 0.5).requires_grad_(True)
        before = native.launch_count(native.K_BWD_FUSED)
        ys = E.channel_scale(x, s)
        gx, gs = torch.autograd.grad([ys], [x, s], [dy])
        assert native.launch_count(native.K_BWD_FUSED) == before + 1
        yr = x * s.to(dtype).reshape(n, c, 1, 1)
        rx, rs = torch.autograd.grad([yr], [x, s], [dy])
        close_lowp(ys.float().detach().cpu().numpy(), yr.float().detach().cpu().numpy(), dtype, 'scale fwd')
        close_lowp(gx.float().cpu().numpy(), rx.float().cpu().numpy(), dtype, 'scale dx')
        close_lowp(gs.float().cpu().numpy(), rs.float().cpu().numpy(), dtype, 'scale ds')
        # (3) epilogue with raw noise + device-side strength, per-sample and shared noise
        for per_sample in (False, True):
            d = (torch.rand([n, c], device=DEV, generator=g) + 0.5).requires_grad_(True)
            strength = torch.tensor(0.37, device=DEV, requires_grad=True)
            raw = torch.randn([n, 1, h, w] if per_sample else [h, w], device=DEV, generator=g)
            bb = torch.randn([c], device=DEV, generator=g).requires_grad_(True)
            before = native.launch_count(native.K_BWD_FUSED)
            y = E.modconv_epilogue(x, d, raw, bb, act='lrelu', clamp=256, noise_strength=strength)
            grads = torch.autograd.grad([y], [x, d, strength, bb], [dy])
            assert native.launch_count(native.K_BWD_FUSED) == before + 1
            y2 = E.modconv_epilogue(x, d, raw * strength, bb, act='lrelu', clamp=256, impl='ref')
            refs = torch.autograd.grad([y2], [x, d, strength, bb], [dy])
            close_lowp(y.float().detach().cpu().numpy(), y2.float().detach().cpu().numpy(), dtype, 'epilogue fwd')
            # 16-bit: the fused kernel adds the unrounded float32 noise, the composition rounds noise*strength to T first;
            # pre-activations within a rounding step of 0 may land on the other side of the lrelu kink, where the
            # derivative jumps from gain to 0.2*gain.  Compare dx away from the kink and the reductions loosely there.
            near = (y2.detach().abs().float() < (1e-6 if dtype == torch.float32 else 0.05)).cpu().numpy()
            close_lowp(grads[0].float().cpu().numpy()[~near], refs[0].float().cpu().numpy()[~near], dtype, f'epilogue dx per_sample={per_sample}')
            if dtype == torch.float32:
                for u, v, nm in zip(grads[1:], refs[1:], ('dcoefs', 'strength', 'bias')):
                    got, want = u.float().cpu().numpy(), v.float().cpu().numpy()
                    assert np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max()) * (1 + 10 * near.mean()), f'{nm} per_sample={per_sample} {dtype}'
                continue
            # 16-bit: (a) the two forwards may disagree on the side of the kink ONLY inside the rounding band counted above;
            flipped = ((y.detach().float() > 0) != (y2.detach().float() > 0)).cpu().numpy()
            assert not (flipped & ~near).any() and flipped.sum() <= near.sum(), f'{flipped.sum()} sign flips, {near.sum()} elements near the kink'
            # (b) given the side each element took in the kernel's OWN forward (the saved y, as bias_act.cu:119-133 decides it),
            # the backward is a float64 sum with no kink left in it: dx to one rounding of T, the float32-accumulated
            # reductions to 1e-3 -- no quantile.
            slope = torch.where(y.detach().double() > 0, 1.0, 0.2) * float(np.sqrt(2))         # yy > 0 (bias_act.cu:86)
            dpre = (dy.double() * slope).to(dtype).double()         # autograd materialises this tensor in T; the kernel rounds it the same way
            d_t = d.detach().to(dtype).double().reshape(n, c, 1, 1)              # the kernel multiplies by dcoefs.to(x.dtype), as the reference
            raw_b = raw.double().reshape([n, 1, h, w] if per_sample else [1, 1, h, w])
            want = [dpre * d_t, (dpre * x.detach().double()).sum([2, 3]), (dpre * raw_b).sum(), dpre.sum([0, 2, 3])]
            # dx = T(dpre * dcoefs) is one more rounding; a double-rounding flip of dpre itself (float32 vs float64 product) is one ulp: 2 ulps
            ulp = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[dtype]
            dx_err = (grads[0].double() - want[0]).abs()
            excess = dx_err - (ulp * want[0].abs() + 3e-7)             # 3e-7: float16 subnormal spacing (6e-8) through the same two roundings
            worst = int(excess.argmax())
            assert excess.flatten()[worst] <= 0, (f'epilogue dx per_sample={per_sample} {dtype}: got {grads[0].flatten()[worst].item():.6g} '
                                                   f'want {want[0].flatten()[worst].item():.6g}')
            for u, v, nm in zip(grads[1:], want[1:], ('dcoefs', 'strength', 'bias')):
                err, scale = (u.double() - v).abs().max().item(), max(1.0, v.abs().max().item())
                # the results are handed back in the parameter's dtype (float32 here); float32 accumulation over <= 131072 terms
                assert err <= 1e-3 * scale, f'{nm} per_sample={per_sample} {dtype}: err {err:.3g} scale {scale:.3g}'


# --------------------------------------------------------------------------------------------
# minibatch standard deviation

def test_minibatch_std_one_launch_each_way(native):
    """mg_mbstd_fwd / mg_mbstd_bwd against the reference composition (training/networks.py:677-690, restated in ops/mbstd._ref) in
    float64: output, first derivative (one native launch each), and the second-order path (create_graph=True goes through the
    composition) -- every compiled group size, several channel blocks, ragged pixel counts, a group of identical samples (sigma at
    its 1e-8 floor)."""
    M = op('mbstd')
    g = torch.Generator(device=DEV).manual_seed(23)
    for shape, group, f in (([32, 512, 4, 4], 4, 1), ([8, 16, 4, 4], 4, 2), ([6, 12, 3, 5], 3, 4), ([2, 8, 4, 4], 4, 1), ([16, 32, 2, 2], 16, 1),
                            ([8, 520, 4, 4], 8, 1), ([1, 8, 4, 4], 4, 1), ([5, 8, 4, 4], None, 2)):
        n = shape[0]
        gg = min(group, n) if group is not None else n
        x = torch.randn(shape, device=DEV, generator=g)
        if n >= 2:
            x[1] = x[0]                      # identical samples inside some group
        x.requires_grad_(True)
        native_path = gg in (1, 2, 3, 4, 8, 16)
        before = native.launch_count(native.K_MBSTD)
        y = M.minibatch_std(x, group, f)
        dy = torch.randn(y.shape, device=DEV, generator=g)
        dx, = torch.autograd.grad([y], [x], [dy])
        assert native.launch_count(native.K_MBSTD) == before + (2 if native_path else 0), (shape, group)
        xr = x.detach().double().requires_grad_(True)
        yr = M._ref(xr, gg, f)
        rx, = torch.autograd.grad([yr], [xr], [dy.double()])
        assert y.shape == yr.shape
        assert (y.double() - yr).abs().max().item() <= 1e-5 * max(1.0, yr.abs().max().item()), (shape, group)
        assert (dx.double() - rx).abs().max().item() <= 1e-5 * max(1.0, rx.abs().max().item()), (shape, group)
        # second order, gradient penalty style (fresh samples: next to sigma's floor the float32 composition itself is ill-conditioned)
        x2 = torch.randn(shape, device=DEV, generator=g).requires_grad_(True)
        y2 = M.minibatch_std(x2, group, f)
        g1, = torch.autograd.grad([(y2 * dy).sum()], [x2], create_graph=True)
        g2, = torch.autograd.grad([g1.square().sum()], [x2])
        xr2 = x2.detach().double().requires_grad_(True)
        r1, = torch.autograd.grad([(M._ref(xr2, gg, f) * dy.double()).sum()], [xr2], create_graph=True)
        r2, = torch.autograd.grad([r1.square().sum()], [xr2])
        assert (g2.double() - r2).abs().max().item() <= 1e-3 * max(1.0, r2.abs().max().item()), (shape, group)


# --------------------------------------------------------------------------------------------
# conv2d_gradfix: the plain backward's two gradient paths

@pytest.mark.parametrize('channels,res,pad', [(128, 32, 1), (256, 16, 1), (128, 20, 0)])
def test_conv2d_gradfix_plain_backward_paths(native, channels, res, pad):
    """Plain backward of the square stride-1 3x3 layers in bf16 channels-last: dx as a forward convolution with the flipped,
    transposed weights (the default) and dx from aten::convolution_backward, both against the float32 autograd gradient of
    the same convolution on the same (bf16-rounded) operands; dw from the single-gradient call.  bf16 operands, float32
    accumulation, ONE rounding of the result: 2^-7 of the gradient's magnitude (LOWP_TOL)."""
    G = op('conv2d_gradfix')
    g = torch.Generator(device=DEV).manual_seed(channels + res)
    x = torch.randn([4, channels, res, res], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn([channels, channels, 3, 3], device=DEV, generator=g) / (3 * channels ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xr, wr = x.float().requires_grad_(True), w.float().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, padding=pad)
    dy = torch.randn(yr.shape, device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    rx, rw = torch.autograd.grad([yr], [xr, wr], [dy.float()])
    old_enabled, old_path = G.enabled, G.data_gradient_as_forward
    G.enabled = True
    try:
        for as_forward in (True, False):
            G.data_gradient_as_forward = as_forward
            xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = G.conv2d(xx, ww, padding=pad)
            gx, gw = torch.autograd.grad([y], [xx, ww], [dy])
            assert gx.shape == x.shape and gw.shape == w.shape
            close_lowp(gx.float().cpu().numpy(), rx.cpu().numpy(), torch.bfloat16, f'dx as_forward={as_forward}')
            close_lowp(gw.float().cpu().numpy(), rw.cpu().numpy(), torch.bfloat16, f'dw as_forward={as_forward}')
            with G.no_weight_gradients():
                y = G.conv2d(xx, ww, padding=pad)
                gx2, = torch.autograd.grad([y], [xx], [dy])
            close_lowp(gx2.float().cpu().numpy(), rx.cpu().numpy(), torch.bfloat16, f'dx alone as_forward={as_forward}')
    finally:
        G.enabled, G.data_gradient_as_forward = old_enabled, old_path


# --------------------------------------------------------------------------------------------
# training_stats.report on the device

@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_training_stats_report_native_moments(native, dtype):
    """mg_stats_accumulate: [count, sum, sum of squares] into the float64 counter, against the aten expression of the
    reference (torch_utils/training_stats.py:58-101) -- small and large, contiguous and strided values, repeated reports."""
    TS = importlib.import_module(f'{PKG}.torch_utils.training_stats')
    TS._counters.clear()
    g = torch.Generator(device=DEV).manual_seed(5)
    before = native.launch_counts()['stats']
    want = torch.zeros([3], dtype=torch.float64)
    vals = [torch.randn([32, 1], device=DEV, generator=g), torch.randn([1], device=DEV, generator=g), torch.randn([3, 70001], device=DEV, generator=g),
            torch.randn([64, 8], device=DEV, generator=g)[:, ::2]]
    for v in vals:
        v = v.to(dtype)
        assert TS.report('t/x', v) is v
        f = v.detach().flatten().to(torch.float32).cpu().double()
        want += torch.stack([torch.tensor(float(f.numel()), dtype=torch.float64), f.sum(), f.square().sum()])
    got = TS._counters['t/x'][torch.device(DEV)].cpu()
    assert native.launch_counts()['stats'] == before + len(vals)
    assert float(got[0]) == float(want[0])
    np.testing.assert_allclose(got[1:].numpy(), want[1:].numpy(), rtol=2e-6, atol=1e-3)
    c = TS.Collector(regex='t/.*')
    TS.report('t/x', torch.ones([4], device=DEV))
    c.update()
    assert c.num('t/x') == 4 and c.mean('t/x') == 1.0
    TS._counters.clear()


# --------------------------------------------------------------------------------------------
# ADA transform parameters in one launch

@pytest.mark.parametrize('spec,ch', [('bgc', 4), ('bgc', 3), ('bg', 4), ('color', 1), ('blit', 4), ('geom', 3), ('color', 4)])
def test_ada_params_native_matches_python_formulation(native, spec, ch):
    """mg_ada_params against AugmentPipe.params_from_tables (the reference-ordered python code run from the same tables):
    G_inv, margin, theta = (left @ G_inv @ right)[:2], colour matrix and its RGBA form."""
    AU, AP = importlib.import_module(f'{PKG}.training.augment'), op('ada_params')
    g = torch.Generator(device=DEV).manual_seed(12)
    for p, n, h, w in ((0.7, 32, 256, 256), (1.0, 5, 24, 40), (0.0, 3, 16, 16)):
        pipe = AU.AugmentPipe(**AU.augpipe_specs[spec]).to(DEV)
        pipe.p.copy_(torch.as_tensor(p))
        U = torch.rand([n, AP.NU], device=DEV, generator=g)
        Z = torch.randn([n, AP.NZ], device=DEV, generator=g)
        left, right, shape = pipe._sampling_factors(n, ch, h, w, (w - 1, h - 1, w - 1, h - 1))
        before = native.launch_counts()['ada_params']
        out = AP.ada_params(pipe, U, Z, w, h, ch, left, right)
        assert native.launch_counts()['ada_params'] == before + 1
        G, m, C = pipe.params_from_tables(U, Z, n, ch, h, w)
        assert (G is None) == (out['G_inv'] is None) and (C is None) == (out['C'] is None)
        if G is not None:
            np.testing.assert_allclose(out['G_inv'].cpu().numpy(), G.cpu().numpy(), rtol=2e-5, atol=2e-5 * max(h, w))
            np.testing.assert_allclose(out['margin'].cpu().numpy(), m.cpu().numpy(), rtol=1e-5, atol=1e-3)
            lr = torch.from_numpy(np.stack([left, right])).to(DEV)
            want = (lr[0] @ G @ lr[1])[:, :2, :]
            np.testing.assert_allclose(out['theta'].cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5)
        if C is not None:
            np.testing.assert_allclose(out['C'].cpu().numpy(), C.cpu().numpy(), rtol=1e-5, atol=1e-5)
            A4 = torch.eye(4, device=DEV).repeat([n, 1, 1])
            A4[:, :3, :3] = C[:, :3, :3]
            np.testing.assert_allclose(out['A4'].cpu().numpy(), A4.cpu().numpy(), rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(out['t4'].cpu().numpy(), torch.nn.functional.pad(C[:, :3, 3], [0, 1]).cpu().numpy(), rtol=1e-5, atol=1e-5)
        if p == 0.0 and G is not None:
            assert torch.equal(out['G_inv'], torch.eye(3, device=DEV).repeat(n, 1, 1))


def test_augment_pipe_native_params_statistics(native):
    """The pipe with native parameters is the same random transformation family: with p = 0 it is the identity (up to the
    resampling filters' float rounding), and the launch counters show ONE parameter kernel per call."""
    AU = importlib.import_module(f'{PKG}.training.augment')
    pipe = AU.AugmentPipe(**AU.augpipe_specs['bgc']).to(DEV)
    x = torch.rand([8, 4, 64, 64], device=DEV) * 2 - 1
    for fixed in (True, False):
        AU.fixed_margin = fixed
        try:
            pipe.p.copy_(torch.as_tensor(0.0))
            before = native.launch_counts()['ada_params']
            y = pipe(x)
            assert native.launch_counts()['ada_params'] == before + 1
            np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), atol=2e-3)
            pipe.p.copy_(torch.as_tensor(1.0))
            y = pipe(x)
            assert y.shape == x.shape and bool(torch.isfinite(y).all()) and float((y - x).abs().mean()) > 1e-2
        finally:
            AU.fixed_margin = False


def test_fixed_margin_device_window_computes_what_is_read(native):
    """augment.fixed_margin pads by the clamp bound (a recorded HIP graph cannot read the data-dependent margin back); with
    `device_window` the pad + upsample pass computes only the tiles the margin COMPUTED ON THE DEVICE asks for and leaves the rest of
    its (9x larger) output unwritten.  Same images (bit for bit) and image gradients as computing everything -- with the allocator's
    free memory poisoned with NaN first, so that a read of an unwritten tile would show."""
    AU = importlib.import_module(f'{PKG}.training.augment')
    pipe = AU.AugmentPipe(**AU.augpipe_specs['bgc']).to(DEV)
    pipe.p.copy_(torch.as_tensor(1.0))
    x = (torch.rand([6, 4, 64, 64], device=DEV, generator=torch.Generator(device=DEV).manual_seed(4)) * 2 - 1)
    AU.fixed_margin = True
    try:
        res = []
        for window in (True, False):
            AU.device_window = window
            poison = torch.full([64 * 1024 * 1024], float('nan'), device=DEV)        # 256 MB of NaN back into the caching allocator
            del poison
            torch.manual_seed(21)
            xx = x.clone().requires_grad_(True)
            y = pipe(xx)
            gx, = torch.autograd.grad([(y * torch.linspace(-1, 1, y.numel(), device=DEV).reshape(y.shape)).sum()], [xx])
            res.append((y.detach(), gx))
        assert bool(torch.isfinite(res[0][0]).all()) and bool(torch.isfinite(res[0][1]).all())
        assert torch.equal(res[0][0], res[1][0])
        # (the image gradient passes through float atomics of the warp's adjoint: equal up to their order)
        np.testing.assert_allclose(res[0][1].cpu().numpy(), res[1][1].cpu().numpy(), rtol=0, atol=2e-6)
        assert float((res[0][0] - x).abs().mean()) > 1e-2                            # the pipe did transform
    finally:
        AU.fixed_margin, AU.device_window = False, True


# --------------------------------------------------------------------------------------------
# fused modulated 3x3 convolution on the matrix cores (csrc/modconv.hip)

@pytest.fixture
def modconv_on():
    """The matrix-core kernel is opt-in (MG_MODCONV=1, see ops/modconv3x3.py): switched on for the duration of a test."""
    M = op('modconv3x3')
    old, M.enabled = M.enabled, True
    yield M
    M.enabled = old


def _modconv_inputs(n, h, w, integer, seed, with_mod=True):
    g = torch.Generator(device=DEV).manual_seed(seed)
    cl = torch.channels_last
    if integer:     # small integers: every product and partial sum is exact in bf16 / float32 -> bit-exact comparison
        x = torch.randint(-2, 3, [n, 64, h, w], device=DEV, generator=g).to(torch.bfloat16)
        wt = torch.randint(-1, 2, [64, 64, 3, 3], device=DEV, generator=g).to(torch.bfloat16)
        s = torch.randint(1, 3, [n, 64], device=DEV, generator=g).float()
        d = torch.ones([n, 64], device=DEV)
        nz = torch.randint(-2, 3, [n, 1, h, w], device=DEV, generator=g).float()
        st = torch.tensor(2.0, device=DEV)
        b = torch.randint(-3, 4, [64], device=DEV, generator=g).float()
    else:
        x = torch.randn([n, 64, h, w], device=DEV, generator=g).to(torch.bfloat16)
        wt = (torch.randn([64, 64, 3, 3], device=DEV, generator=g) / 24).to(torch.bfloat16)
        s = torch.randn([n, 64], device=DEV, generator=g) * 0.5 + 1
        d = torch.rand([n, 64], device=DEV, generator=g) + 0.5
        nz = torch.randn([n, 1, h, w], device=DEV, generator=g)
        st = torch.tensor(0.3, device=DEV)
        b = torch.randn([64], device=DEV, generator=g) * 0.2
    x, wt = x.contiguous(memory_format=cl), wt.contiguous(memory_format=cl)
    if not with_mod:
        s = d = nz = st = None
    return x, wt, s, d, nz, st, b


@pytest.mark.parametrize('shape', [(2, 8, 32), (3, 16, 64), (2, 64, 32), (1, 256, 256)])
@pytest.mark.parametrize('with_mod', [True, False])
def test_fused_modconv3x3_integer_data_is_bit_exact(native, modconv_on, shape, with_mod):
    """mg_modconv3x3_c64 on integer-valued data (exact in bf16 and float32) against the three-launch path: tile seams, image
    borders, the MFMA operand / accumulator lane maps and the channels-last weight layout all show up as bit differences."""
    M = op('modconv3x3')
    n, h, w = shape
    x, wt, s, d, nz, st, b = _modconv_inputs(n, h, w, True, 3, with_mod)
    bb = b if with_mod else b.to(torch.bfloat16)
    before = native.launch_counts()['modconv']
    assert M.applicable(x, wt)
    for act, gain, clamp in (('lrelu', 1.0, None), ('linear', 1.0, None), ('lrelu', 2.0, 16.0)):
        y = M.fused_modconv3x3(x, wt, s, d, nz, st, bb, act=act, gain=gain, clamp=clamp)
        ref = M._composition(x, wt, s, d, nz, st, bb, act, 0.2 if act == 'lrelu' else 0.0, gain, -1.0 if clamp is None else clamp)
        assert y.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.bfloat16
        assert torch.equal(y, ref), (act, float((y.float() - ref.float()).abs().max()))
    assert native.launch_counts()['modconv'] == before + 3


@pytest.mark.parametrize('with_mod', [True, False])
def test_fused_modconv3x3_values_and_gradients(native, modconv_on, with_mod):
    """Random data: forward and every gradient (x, weight, styles, dcoefs, noise strength, bias) against the three-launch path,
    both with a plain backward and under create_graph=True (second-order term of the path-length regulariser).
    Bounds (bf16): the output within 3e-2 of the tensor maximum (two ulps of bf16 at that magnitude plus the different rounding of
    the modulation -- the kernel rounds w * s, the three launches x * s; both are what the reference computes, fused_modconv
    True / False).  Gradients: mean error within 1.5e-2 of the maximum (sums over thousands of pixels of bf16 products whose
    operands differ in their last bit); single elements may differ by a whole leaky-ReLU slope (0.8 * |dy| * gain) where the two
    roundings put a pre-activation on different sides of zero, hence 0.2 of the maximum."""
    M = op('modconv3x3')
    n, h, w = 3, 32, 64
    x, wt, s, d, nz, st, b = _modconv_inputs(n, h, w, False, 11, with_mod)
    bb = b if with_mod else b.to(torch.bfloat16)
    names = ['x', 'w'] + (['s', 'd', 'st', 'b'] if with_mod else ['b'])
    dy = torch.randn([n, 64, h, w], device=DEV, generator=torch.Generator(device=DEV).manual_seed(12)).to(torch.bfloat16) \
        .contiguous(memory_format=torch.channels_last)            # (its own generator: the same values whichever tests ran before)
    # the natural size of the noise-strength gradient, ONE number sum_px dpre * noise of zero-mean terms: (sum of squared terms)^0.5
    st_scale = float(dy.float().square().sum().sqrt()) * float(np.sqrt(2))

    def run(fused, second):
        leaves = [t.clone().requires_grad_(True) for t in ([x, wt] + ([s, d, st, bb] if with_mod else [bb]))]
        xx, ww = leaves[0], leaves[1]
        ss, dd, stt, b_ = (leaves[2], leaves[3], leaves[4], leaves[5]) if with_mod else (None, None, None, leaves[2])
        if fused:
            y = M.fused_modconv3x3(xx, ww, ss, dd, nz, stt, b_, act='lrelu', gain=np.sqrt(2), clamp=256)
        else:
            y = M._composition(xx, ww, ss, dd, nz, stt, b_, 'lrelu', 0.2, float(np.sqrt(2)), 256.0)
        if not second:
            grads = torch.autograd.grad([y], leaves, [dy])
            return [y.detach()] + list(grads)
        g1 = torch.autograd.grad([y], [xx] + ([ss] if with_mod else []), [dy], create_graph=True)
        tot = sum(t.float().square().sum() for t in g1)
        g2 = torch.autograd.grad([tot], [ww] + ([ss] if with_mod else []), allow_unused=True)
        return [t.detach() for t in g1] + [t for t in g2 if t is not None]
    for second in (False, True):
        got, want = run(True, second), run(False, second)
        assert len(got) == len(want)
        for i, (a, r) in enumerate(zip(got, want)):
            scale = float(r.float().abs().max()) + 1e-6
            err = float((a.float() - r.float()).abs().max())
            name = names[i - 1] if (not second and i) else ''
            # the noise-strength gradient: ~0.5 % of its N terms flip by a leaky-ReLU slope, a random sum of ~(0.005 N)^0.5 of them;
            # its own value can be anything down to 0, so the error is held against the natural size of the sum instead
            # (measured: 0.02-0.05 of it)
            if name == 'st':
                scale = st_scale
            bound = 3e-2 if (i == 0 and not second) else 0.1 if name == 'st' else 0.2
            assert err <= bound * scale, (second, i, names[i - 1] if (not second and i) else i, err, scale)
            mean = float((a.float() - r.float()).abs().mean())
            assert mean <= (4e-3 if (i == 0 and not second) else 0.1 if name == 'st' else 1.5e-2) * scale, (second, i, name, mean, scale)


@pytest.mark.parametrize('shape', [(2, 4, 32), (3, 16, 64), (1, 64, 96), (2, 256, 256)])
def test_wgrad3x3_c64_integer_data_is_exact(native, shape):
    """mg_wgrad3x3_c64 (csrc/wgrad.hip: transposed LDS reads feeding v_mfma_f32_32x32x16_bf16, per-sample style scaling, the style
    gradient from the same accumulators) on integer-valued data against float32 library convolutions of the same tensors: every
    product and sum is exact in both, so dW (float32, before its cast) and d(styles) must agree EXACTLY -- tile seams, image borders,
    the transposed-read lane maps, the swizzled LDS image and the accumulator-to-(co, tap, ci) map all show up as differences."""
    M = op('modconv3x3')
    n, h, w = shape
    g = torch.Generator(device=DEV).manual_seed(21)
    cl = torch.channels_last
    x = torch.randint(-2, 3, [n, 64, h, w], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
    dy = torch.randint(-2, 3, [n, 64, h, w], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
    # (sparse gradients keep |sum| < 2^24 at the largest shape)
    dy = (dy * (torch.rand([n, 64, h, w], device=DEV, generator=g) < 0.25)).contiguous(memory_format=cl)
    wt = torch.randint(-1, 2, [64, 64, 3, 3], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
    s = torch.randint(1, 3, [n, 64], device=DEV, generator=g).float()
    before = native.launch_count(native.K_WGRAD)
    for styles in (s, None):
        dw, ds = M.wgrad3x3_c64(dy, x, styles, wt if styles is not None else None, raw=True)
        xs32 = x.float() * (styles if styles is not None else torch.ones_like(s)).reshape(n, 64, 1, 1)
        dxs_ref, dw_ref, _ = torch.ops.aten.convolution_backward(dy.float().contiguous(), xs32.contiguous(), wt.float().contiguous(), None, [1, 1], [1, 1], [1, 1],
                                                               False, [0, 0], 1, [True, True, False])
        assert tuple(dw.shape) == (64, 64, 3, 3) and dw.dtype == torch.float32
        assert torch.equal(dw, dw_ref), (styles is not None, float((dw - dw_ref).abs().max()), float(dw_ref.abs().max()))
        if styles is not None:
            # d(x * s)/ds = sum_p dxs * x with dxs the data gradient of the convolution
            ds_ref = (dxs_ref * x.float()).sum(dim=[2, 3])
            assert torch.equal(ds, ds_ref), float((ds - ds_ref).abs().max())
        else:
            assert ds is None
    assert native.launch_count(native.K_WGRAD) == before + 2
    # and the data gradient with the modulation's x-gradient folded in: conv(dy, flipped w^T) * styles, rounded as the two launches round
    dx = M.conv3x3_c64(dy, M.flipped_transposed(wt), scale=s)
    dxs = torch.ops.aten.convolution_backward(dy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    assert torch.equal(dx, (dxs * s.to(torch.bfloat16).reshape(n, 64, 1, 1)))


def test_synthesis_and_discriminator_take_the_fused_kernel(native, modconv_on):
    """The 256x256 layers of the benchmark configuration (64 channels, bf16, channels-last) run mg_modconv3x3_c64 inside
    SynthesisNetwork and Discriminator; outputs agree with the three-launch path."""
    NW, M = importlib.import_module(f'{PKG}.training.networks'), op('modconv3x3')
    torch.manual_seed(0)
    S = NW.SynthesisNetwork(w_dim=64, img_resolution=256, img_channels=4, init_res=[8, 8], conv_config_index=3, channel_base=16384,
                            channel_max=64, num_fp16_res=4, conv_clamp=256, fp16_dtype=torch.bfloat16, fp16_channels_last=True).to(DEV).train()
    D = NW.Discriminator(img_resolution=256, img_channels=4, init_res=[8, 8], conv_config_index=3, channel_base=16384, channel_max=64,
                         num_fp16_res=4, conv_clamp=256, block_kwargs=dict(fp16_dtype=torch.bfloat16, fp16_channels_last=True)).to(DEV).train()
    ws = torch.randn([2, S.num_ws, 64], device=DEV)
    res = []
    for on in (True, False):
        M.enabled = on
        try:
            before = native.launch_counts()['modconv']
            img = S(ws, noise_mode='const')
            logits = D(img)
            used = native.launch_counts()['modconv'] - before
            res.append((img.detach(), logits.detach(), used))
        finally:
            M.enabled = True        # (the fixture restores the default afterwards)
    assert res[0][2] >= 2 and res[1][2] == 0            # at least G b256.conv1 and D b256.conv0 (here every 64-channel block qualifies)
    assert float((res[0][0] - res[1][0]).abs().max()) <= 3e-2 * float(res[1][0].abs().max())
    assert float((res[0][1] - res[1][1]).abs().max()) <= 3e-2 * float(res[1][1].abs().max()) + 1e-3
