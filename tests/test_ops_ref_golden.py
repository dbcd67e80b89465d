"""The package's own aten (`impl='ref'`, CPU) implementations of the plugin API against the golden
vectors produced by the reference (tests/golden/make_golden.py).  These are the semantics the CPU
configuration of BASELINE.json runs and the `cpu_baseline` leg of bench.py times.  CPU only."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden, load_pkg

PKG = 'docker-montage-gan_amd'


def op(name):
    load_pkg()
    return importlib.import_module(f'{PKG}.torch_utils.ops.{name}')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def run_upfirdn2d_case(U, c, arr, device='cpu', impl='ref', dtype=None):
    i = c['id']
    x = T(arr[f'c{i}_x'])
    if dtype is not None:
        x = x.to(dtype)
    x = x.to(device)
    if c['memfmt'] == 'channels_last':
        x = x.contiguous(memory_format=torch.channels_last)
    f = T(arr[f'c{i}_f']).to(device) if f'c{i}_f' in arr else None
    x.requires_grad_(True)
    fn = dict(upfirdn2d=U.upfirdn2d, upsample2d=U.upsample2d, downsample2d=U.downsample2d, filter2d=U.filter2d)[c['kind']]
    kw = dict(c['kwargs'])
    y = fn(x, f, impl=impl, **kw)
    dy = T(arr[f'c{i}_dy']).to(device=device, dtype=y.dtype)
    dx, = torch.autograd.grad([y], [x], [dy])
    return y, dx


def test_upfirdn2d_ref_matches_reference():
    U = op('upfirdn2d')
    cases, arr = load_golden('upfirdn2d')
    for c in cases:
        y, dx = run_upfirdn2d_case(U, c, arr)
        i = c['id']
        assert list(y.shape) == c['out_shape']
        np.testing.assert_allclose(y.detach().numpy(), arr[f'c{i}_y'], rtol=1e-5, atol=1e-5, err_msg=str(c))
        np.testing.assert_allclose(dx.numpy(), arr[f'c{i}_dx'], rtol=1e-5, atol=1e-5, err_msg=str(c))


def test_setup_filter_matches_reference():
    U = op('upfirdn2d')
    cases, arr = load_golden('upfirdn2d')
    f = U.setup_filter([1, 3, 3, 1])
    np.testing.assert_allclose(f.numpy(), arr['c0_f'], rtol=1e-7)
    assert f.shape == (4, 4) and abs(float(f.sum()) - 1) < 1e-6
    f12 = U.setup_filter(list(range(1, 13)))
    assert f12.ndim == 1                       # >= 8 taps -> separable
    assert U.setup_filter(None).shape == (1, 1)
    g = U.setup_filter([1, 2, 1], gain=4, flip_filter=True)
    assert abs(float(g.sum()) - 4) < 1e-6


def run_bias_act_case(B, c, arr, device='cpu', impl='ref'):
    i = c['id']
    x = T(arr[f'c{i}_x']).to(device)
    if c['memfmt'] == 'channels_last':
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    b = T(arr[f'c{i}_b']).to(device).requires_grad_(True) if c['with_b'] else None
    y = B.bias_act(x, b, dim=c['dim'], act=c['act'], alpha=c['alpha'], gain=c['gain'], clamp=c['clamp'], impl=impl)
    dy = T(arr[f'c{i}_dy']).to(device).requires_grad_(True)
    ins = [x] + ([b] if b is not None else [])
    grads = torch.autograd.grad([y], ins, [dy], create_graph=True)
    out = dict(y=y, dx=grads[0], db=grads[1] if b is not None else None)
    if grads[0].requires_grad and f'c{i}_g2x' in arr:
        ddx = T(arr[f'c{i}_ddx']).to(device)
        g2 = torch.autograd.grad([grads[0]], [x, dy], [ddx], allow_unused=True)
        out['g2x'] = g2[0] if g2[0] is not None else torch.zeros_like(x)
        out['g2dy'] = g2[1] if g2[1] is not None else torch.zeros_like(x)
    return out


def test_bias_act_ref_matches_reference():
    B = op('bias_act')
    cases, arr = load_golden('bias_act')
    for c in cases:
        i = c['id']
        out = run_bias_act_case(B, c, arr)
        tol = dict(rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out['y'].detach().numpy(), arr[f'c{i}_y'], err_msg=str(c), **tol)
        np.testing.assert_allclose(out['dx'].detach().numpy(), arr[f'c{i}_dx'], err_msg=str(c), **tol)
        if out['db'] is not None:
            np.testing.assert_allclose(out['db'].detach().numpy(), arr[f'c{i}_db'], err_msg=str(c), rtol=1e-5, atol=1e-5)
        if 'g2x' in out:
            np.testing.assert_allclose(out['g2x'].numpy(), arr[f'c{i}_g2x'], err_msg=str(c), rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(out['g2dy'].numpy(), arr[f'c{i}_g2dy'], err_msg=str(c), **tol)
    assert set(B.activation_funcs) == {'linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'}
    assert [B.activation_funcs[k].cuda_idx for k in ('linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish')] == list(range(1, 10))


def run_conv_case(c, arr, device='cpu', gradfix=False):
    CR, U, FMA, G = op('conv2d_resample'), op('upfirdn2d'), op('fma'), op('conv2d_gradfix')
    i = c['id']
    f = U.setup_filter([1, 3, 3, 1]).to(device)
    old = G.enabled
    G.enabled = gradfix
    try:
        if c['op'] == 'conv2d_resample':
            x = T(arr[f'c{i}_x']).to(device).requires_grad_(True)
            w = T(arr[f'c{i}_w']).to(device).requires_grad_(True)
            y = CR.conv2d_resample(x, w, f=f, **c['kwargs'])
            dx, dw = torch.autograd.grad([y], [x, w], [T(arr[f'c{i}_dy']).to(device)])
            return dict(y=y, dx=dx, dw=dw)
        if c['op'] == 'fma':
            a, b, cc = (T(arr[f'c{i}_{k}']).to(device).requires_grad_(True) for k in 'abc')
            y = FMA.fma(a, b, cc)
            da, db, dc = torch.autograd.grad([y], [a, b, cc], [T(arr[f'c{i}_dy']).to(device)])
            return dict(y=y, da=da, db=db, dc=dc)
        if c['op'] == 'modulated_conv2d':
            # training/networks.py:37-98 of the reference: fused (grouped conv, per-sample weights) and non-fused
            # (scale -> shared conv -> fma) forms, demodulation on/off, noise on/off, up 1/2
            NW = importlib.import_module(f'{PKG}.training.networks')
            x, w, s = (T(arr[f'c{i}_{k}']).to(device).requires_grad_(True) for k in 'xws')
            nz = T(arr[f'c{i}_noise']).to(device) if c['noise'] else None
            y = NW.modulated_conv2d(x=x, weight=w, styles=s, noise=nz, up=c['up'], padding=1, resample_filter=f,
                                    demodulate=c['demodulate'], flip_weight=(c['up'] == 1), fused_modconv=c['fused'])
            dx, dw, ds = torch.autograd.grad([y], [x, w, s], [T(arr[f'c{i}_dy']).to(device)])
            return dict(y=y, dx=dx, dw=dw, ds=ds)
    finally:
        G.enabled = old
    return None


@pytest.mark.parametrize('gradfix', [False, True])
def test_conv2d_resample_modulated_conv2d_and_fma_match_reference(gradfix):
    cases, arr = load_golden('conv')
    n = 0
    for c in cases:
        out = run_conv_case(c, arr, gradfix=gradfix)
        assert out is not None, c               # every committed vector is in force
        n += 1
        for k, v in out.items():
            np.testing.assert_allclose(v.detach().numpy(), arr[f'c{c["id"]}_{k}'], rtol=2e-4, atol=2e-4, err_msg=f'{c} {k}')
    assert n == 26          # 9 conv2d_resample + 16 modulated_conv2d + 1 fma


def test_conv2d_gradfix_double_backward_and_no_weight_gradients():
    """R1 / path-length style second-order graph through the hand-written conv autograd (CPU, aten)."""
    G = op('conv2d_gradfix')
    torch.manual_seed(0)
    for transpose in (False, True):
        x = torch.randn(2, 3, 6, 6, dtype=torch.float64, requires_grad=True)
        w = torch.randn(3 if transpose else 4, 4 if transpose else 3, 3, 3, dtype=torch.float64, requires_grad=True)
        kw = dict(stride=2, padding=1)

        def run(custom):
            G.enabled = custom
            try:
                fn = G.conv_transpose2d if transpose else G.conv2d
                y = fn(x, w, **kw)
                gx, = torch.autograd.grad([y.square().sum()], [x], create_graph=True)
                return torch.autograd.grad([gx.square().sum()], [x, w])
            finally:
                G.enabled = False
        a, b = run(True), run(False)
        for u, v in zip(a, b):
            torch.testing.assert_close(u, v, rtol=1e-9, atol=1e-9)
        # inside no_weight_gradients() the weight gets no gradient at all
        G.enabled = True
        try:
            y = (G.conv_transpose2d if transpose else G.conv2d)(x, w, **kw)
            with G.no_weight_gradients():
                gx, gw = torch.autograd.grad([y.sum()], [x, w], allow_unused=True)
            assert gw is None and gx is not None
        finally:
            G.enabled = False


def test_grid_sample_gradfix_matches_functional():
    GS = op('grid_sample_gradfix')
    torch.manual_seed(1)
    x = torch.randn(2, 3, 7, 6, dtype=torch.float64, requires_grad=True)
    grid = (torch.rand(2, 5, 4, 2, dtype=torch.float64) * 2.4 - 1.2).requires_grad_(True)

    def run(custom):
        GS.enabled = custom
        try:
            y = GS.grid_sample(x, grid)
            gx, gg = torch.autograd.grad([y.square().sum()], [x, grid])
            return y, gx, gg
        finally:
            GS.enabled = False
    for u, v in zip(run(True), run(False)):
        torch.testing.assert_close(u, v, rtol=1e-9, atol=1e-9)
    # second order w.r.t. the input: aten has no derivative for grid_sampler_2d_backward (the reason this
    # op exists), so the custom graph is checked numerically instead
    GS.enabled = True
    try:
        g = grid.detach()
        xs = torch.randn(1, 2, 4, 4, dtype=torch.float64, requires_grad=True)
        assert torch.autograd.gradcheck(lambda t: GS.grid_sample(t, g[:1]), [xs])
        assert torch.autograd.gradgradcheck(lambda t: GS.grid_sample(t, g[:1]), [xs])
    finally:
        GS.enabled = False


def test_alpha_composite_ref_matches_reference():
    A = op('alpha_composite')
    cases, arr = load_golden('montage')
    x = T(arr['png_layers_u8'].astype(np.float32) / 255.0)
    np.testing.assert_allclose(A.alpha_composite(x, impl='ref').numpy(), arr['png_out'], rtol=1e-6, atol=1e-6)
    for c in cases:
        if c['op'] != 'alpha_composite' or c['src'] != 'rand':
            continue
        i = c['id']
        x = T(arr[f'c{i}_x']).requires_grad_(True)
        y = A.alpha_composite(x, impl='ref')
        dx, = torch.autograd.grad([y], [x], [T(arr[f'c{i}_dy'])])
        np.testing.assert_allclose(y.detach().numpy(), arr[f'c{i}_y'], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(dx.numpy(), arr[f'c{i}_dx'], rtol=1e-4, atol=1e-5, equal_nan=True)


def test_translate_sample_ref_matches_reference():
    TS = op('translate_sample')
    cases, arr = load_golden('montage')
    for c in cases:
        if c['op'] != 'translate_sample':
            continue
        i = c['id']
        b, l, ch, h, w = c['shape']
        x = T(arr[f'c{i}_x']).reshape(b * l, ch, h, w).requires_grad_(True)
        t = T(arr[f'c{i}_t']).reshape(b * l, 2).requires_grad_(True)
        np.testing.assert_allclose(TS.translation_to_theta(t.detach()).numpy(), arr[f'c{i}_theta'], rtol=0, atol=0)
        y = TS.translate_sample(x, t, shift=c['shift'], impl='ref')
        dx, dt = torch.autograd.grad([y], [x, t], [T(arr[f'c{i}_dy'])])
        np.testing.assert_allclose(y.detach().numpy(), arr[f'c{i}_y'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dx.numpy(), arr[f'c{i}_dx'].reshape(x.shape), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dt.numpy(), arr[f'c{i}_dt'].reshape(-1, 2), rtol=1e-4, atol=1e-4)


def test_affine_warp_and_channel_mix_ref_paths_match_reference():
    """CPU tensors take the reference composition inside affine_warp / channel_mix (no native code without a GPU)."""
    W, M = op('affine_warp'), op('channel_mix')
    cases, arr = load_golden('warp')
    for c in cases:
        i = c['id']
        if c['kind'] == 'warp':
            x = T(arr[f'w{i}_x']).requires_grad_(True)
            y = W.affine_warp(x, T(arr[f'w{i}_theta']), *c['out'])
            dx, = torch.autograd.grad([y], [x], [T(arr[f'w{i}_gy'])])
            np.testing.assert_allclose(y.detach().numpy(), arr[f'w{i}_y'], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(dx.numpy(), arr[f'w{i}_dx'], rtol=1e-5, atol=1e-6)
        else:
            x, C = T(arr[f'm{i}_x']).requires_grad_(True), T(arr[f'm{i}_C'])
            ch = c['channels']
            A = torch.eye(ch).repeat(x.shape[0], 1, 1)
            A[:, :3, :3] = C[:, :3, :3]
            t = torch.nn.functional.pad(C[:, :3, 3], [0, ch - 3])
            y = M.channel_mix(x, A, t)
            dx, = torch.autograd.grad([y], [x], [T(arr[f'm{i}_gy'])])
            np.testing.assert_allclose(y.detach().numpy(), arr[f'm{i}_y'], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(dx.numpy(), arr[f'm{i}_dx'], rtol=1e-5, atol=1e-5)


def test_demod_coefs_composite_matches_definition(oracle):
    """The [N,I] x [I,O] form used by the networks equals the reference's definition on per-sample weights
    (training/networks.py:76-77), values and gradients (CPU: the composite path)."""
    D = op('demod')
    g = torch.Generator().manual_seed(5)
    for oc, ic, k, n in [(6, 5, 3, 4), (8, 16, 1, 3)]:
        w = torch.randn([oc, ic, k, k], generator=g, dtype=torch.float64, requires_grad=True)
        s = (torch.randn([n, ic], generator=g, dtype=torch.float64) + 1).requires_grad_(True)
        d = D.demod_coefs(w, s)
        ref = (w.unsqueeze(0) * s.reshape(n, 1, ic, 1, 1)).square().sum(dim=[2, 3, 4]).add(1e-8).rsqrt()       # upstream's expression
        np.testing.assert_allclose(d.detach().numpy(), ref.detach().numpy(), rtol=1e-12)
        np.testing.assert_allclose(d.detach().numpy(), oracle.demod_coefs(w.detach().numpy(), s.detach().numpy()), rtol=1e-12)
        gy = torch.randn(d.shape, generator=g, dtype=torch.float64)
        dw, ds = torch.autograd.grad([d], [w, s], [gy])
        ow, os_ = oracle.demod_coefs_bwd(w.detach().numpy(), s.detach().numpy(), gy.numpy())
        np.testing.assert_allclose(dw.numpy(), ow, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(ds.numpy(), os_, rtol=1e-10, atol=1e-12)
