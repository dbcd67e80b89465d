"""The oracle (oracle/montage_oracle.c) against every golden vector produced by the
reference's own CPU implementations (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden

F32_TOL = dict(rtol=2e-5, atol=2e-5)


def _kw_to_plugin(case, f):
    """Resolve the reference's convenience wrappers (upfirdn2d.py:290-402) to
    (up, down, padding4, flip, gain) exactly as they do; returns python ints."""
    kw = dict(case['kwargs'])
    kind = case['kind']

    def two(v):
        return (v, v) if isinstance(v, int) else tuple(v)

    def pad4(p):
        if isinstance(p, int):
            p = [p, p]
        p = list(p)
        if len(p) == 2:
            p = [p[0], p[0], p[1], p[1]]
        return p

    if f is None:
        fw = fh = 1
    elif f.ndim == 1:
        fw = fh = f.shape[0]
    else:
        fh, fw = f.shape
    up, down = two(kw.get('up', 1)), two(kw.get('down', 1))
    px0, px1, py0, py1 = pad4(kw.get('padding', 0))
    gain = kw.get('gain', 1)
    if kind == 'filter2d':
        up = down = (1, 1)
        p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    elif kind == 'upsample2d':
        up = two(kw.get('up', 2)); down = (1, 1)
        p = [px0 + (fw + up[0] - 1) // 2, px1 + (fw - up[0]) // 2, py0 + (fh + up[1] - 1) // 2, py1 + (fh - up[1]) // 2]
        gain = gain * up[0] * up[1]
    elif kind == 'downsample2d':
        down = two(kw.get('down', 2)); up = (1, 1)
        p = [px0 + (fw - down[0] + 1) // 2, px1 + (fw - down[0]) // 2, py0 + (fh - down[1] + 1) // 2, py1 + (fh - down[1]) // 2]
    else:
        p = [px0, px1, py0, py1]
    return list(up), list(down), p, bool(kw.get('flip_filter', False)), float(gain)


def test_upfirdn2d_forward_and_backward(oracle):
    cases, arr = load_golden('upfirdn2d')
    assert len(cases) >= 70
    for c in cases:
        i = c['id']
        x = arr[f'c{i}_x']
        f = arr[f'c{i}_f'] if f'c{i}_f' in arr else None
        up, down, p, flip, gain = _kw_to_plugin(c, f)
        y = oracle.upfirdn2d_api(x, f, up=up, down=down, padding=p, flip_filter=flip, gain=gain)
        assert list(y.shape) == c['out_shape'], c
        np.testing.assert_allclose(y, arr[f'c{i}_y'], err_msg=str(c), **F32_TOL)
        # backward = the same op with up<->down, flipped filter, recomputed pad (upfirdn2d.py:263-278)
        ih, iw = x.shape[2:]
        oh, ow = y.shape[2:]
        fw, fh = (1, 1) if f is None else ((f.shape[0],) * 2 if f.ndim == 1 else (f.shape[1], f.shape[0]))
        pb = [fw - p[0] - 1, iw * up[0] - ow * down[0] + p[0] - up[0] + 1,
              fh - p[2] - 1, ih * up[1] - oh * down[1] + p[2] - up[1] + 1]
        dx = oracle.upfirdn2d_api(arr[f'c{i}_dy'], f, up=down, down=up, padding=pb, flip_filter=not flip, gain=gain)
        np.testing.assert_allclose(dx, arr[f'c{i}_dx'], err_msg=str(c), **F32_TOL)


def test_upfirdn2d_index_table_matches_definition(oracle):
    """Bit-exact integer path: the oracle's (in0, count, f0) against a brute-force
    enumeration of the reference DEFINITION (zero-insert, pad/crop, valid FIR,
    decimate -- upfirdn2d.py:197-220), for every parameter combination incl.
    negative pads."""
    for up in (1, 2, 3, 4):
        for down in (1, 2, 3):
            for fs in (1, 2, 3, 4, 7, 12):
                for pad0 in (-3, -1, 0, 1, 2, 5, 11):
                    for pad1 in (-2, 0, 3):
                        for in_size in (1, 5, 8, 17):
                            out = oracle.out_size(in_size, up, down, pad0, pad1, fs)
                            if out < 1:
                                continue
                            for flip in (0, 1):
                                t = oracle.index_table(out, in_size, up, down, pad0, fs, flip)
                                for o in range(out):
                                    taps = []          # (input index, filter index) pairs by definition
                                    for k in range(fs):
                                        j = o * down + k - pad0      # position in the zero-inserted signal
                                        if j % up == 0 and 0 <= j // up < in_size:
                                            taps.append((j // up, k if flip else fs - 1 - k))
                                    in0, cnt, f0 = (int(v) for v in t[o])
                                    step = up if flip else -up
                                    got = [(in0 + q, f0 + q * step) for q in range(cnt)]
                                    assert got == taps, (up, down, fs, pad0, pad1, in_size, flip, o, got, taps)


def _phys(a, memfmt):
    """Physical (dense) view of an NCHW array stored channels_last."""
    return np.ascontiguousarray(a.transpose(0, 2, 3, 1)) if memfmt == 'channels_last' else a


def test_bias_act_all_orders(oracle):
    cases, arr = load_golden('bias_act')
    for c in cases:
        i = c['id']
        idx, da, dg, ref, has2 = oracle.ACTS[c['act']]
        alpha = da if c['alpha'] is None else c['alpha']
        gain = dg if c['gain'] is None else c['gain']
        clamp = -1.0 if c['clamp'] is None else c['clamp']
        x, y_ref = arr[f'c{i}_x'], arr[f'c{i}_y']
        b = arr[f'c{i}_b'] if c['with_b'] else None
        dim = c['dim']
        # gain/alpha/clamp cross the plugin boundary as C float (bias_act.cpp:32): ~6e-8 relative in float64
        tol = dict(rtol=2e-7, atol=2e-7) if c['dtype'] == 'float64' else dict(rtol=3e-5, atol=3e-5)
        y = oracle.bias_act_raw(x, b, grad=0, dim=dim, act=idx, alpha=alpha, gain=gain, clamp=clamp)
        np.testing.assert_allclose(y, y_ref, err_msg=str(c), **tol)
        # first order (bias_act.py:198-206): dx = plugin(dy, b, x, y, null, grad=1)
        dy = arr[f'c{i}_dy']
        dx = oracle.bias_act_raw(dy, b, xref=x, yref=y, grad=1, dim=dim, act=idx, alpha=alpha, gain=gain, clamp=clamp)
        mask = np.ones(x.shape, bool)
        if c['act'] in ('relu', 'lrelu', 'elu', 'selu'):
            # x + b == 0 exactly: the native kernel (bias_act.cu:62-106, which the oracle follows) and aten's
            # autograd pick different one-sided derivatives; a measure-zero set, excluded.
            mask &= np.abs(x + (0 if b is None else b.reshape([-1 if k == dim else 1 for k in range(x.ndim)]))) > 1e-6
        if clamp >= 0:   # autograd of clamp() passes gradient AT the boundary, the kernel does not: measure-zero set
            mask &= np.abs(np.abs(y_ref) - clamp) > 1e-6
        np.testing.assert_allclose(dx[mask], arr[f'c{i}_dx'][mask], err_msg=str(c), **tol)
        if b is not None:
            db = dx.sum(axis=tuple(k for k in range(x.ndim) if k != dim))
            if mask.all():
                np.testing.assert_allclose(db, arr[f'c{i}_db'], err_msg=str(c), rtol=1e-4, atol=1e-4)
        # second order
        if f'c{i}_g2x' in arr:
            ddx = arr[f'c{i}_ddx']
            g2dy = oracle.bias_act_raw(ddx, b, xref=x, yref=y, grad=1, dim=dim, act=idx, alpha=alpha, gain=gain, clamp=clamp)
            np.testing.assert_allclose(g2dy[mask], arr[f'c{i}_g2dy'][mask], err_msg=str(c), **tol)
            if has2:
                g2x = oracle.bias_act_raw(ddx, b, xref=x, yref=y, dy=dy, grad=2, dim=dim, act=idx, alpha=alpha, gain=gain, clamp=clamp)
                tol2 = dict(rtol=1e-6, atol=1e-6) if c['dtype'] == 'float64' else dict(rtol=2e-4, atol=2e-4)
                np.testing.assert_allclose(g2x[mask], arr[f'c{i}_g2x'][mask], err_msg=str(c), **tol2)


def test_alpha_composite(oracle):
    cases, arr = load_golden('montage')
    x = arr['png_layers_u8'].astype(np.float32) / 255.0
    y = oracle.alpha_composite(x[None])[0]
    np.testing.assert_allclose(y, arr['png_out'], rtol=1e-6, atol=1e-6)
    for c in cases:
        if c['op'] != 'alpha_composite' or c['src'] != 'rand':
            continue
        i = c['id']
        x, dy = arr[f'c{i}_x'], arr[f'c{i}_dy']
        np.testing.assert_allclose(oracle.alpha_composite(x), arr[f'c{i}_y'], rtol=1e-5, atol=1e-6, err_msg=str(c))
        dx = oracle.alpha_composite_bwd(x, dy)
        ref = arr[f'c{i}_dx']
        assert np.array_equal(np.isnan(dx), np.isnan(ref)), c      # NaN exactly where torch's autograd has it
        np.testing.assert_allclose(dx, ref, rtol=2e-4, atol=2e-4, equal_nan=True, err_msg=str(c))
        safe = oracle.alpha_composite_bwd(x, dy, safe_grad=True)
        assert np.isfinite(safe).all()
        ok = ~np.isnan(ref).reshape(ref.shape[0], -1, ref.shape[3], ref.shape[4]).any(axis=1)
        for bi in range(ref.shape[0]):
            np.testing.assert_allclose(safe[bi][:, :, ok[bi]], ref[bi][:, :, ok[bi]], rtol=2e-4, atol=2e-4)


def test_translate_sample(oracle):
    cases, arr = load_golden('montage')
    n = 0
    for c in cases:
        if c['op'] != 'translate_sample':
            continue
        n += 1
        i = c['id']
        b, l, ch, h, w = c['shape']
        x = arr[f'c{i}_x'].reshape(b * l, ch, h, w)
        t = arr[f'c{i}_t'].reshape(b * l, 2)
        y = oracle.translate_sample(x, t, shift=c['shift'])
        np.testing.assert_allclose(y, arr[f'c{i}_y'], rtol=1e-4, atol=1e-4, err_msg=str(c))
        dx, dt = oracle.translate_sample_bwd(x, arr[f'c{i}_dy'], t, shift=c['shift'])
        np.testing.assert_allclose(dx, arr[f'c{i}_dx'].reshape(x.shape), rtol=1e-4, atol=1e-4, err_msg=str(c))
        np.testing.assert_allclose(dt, arr[f'c{i}_dt'].reshape(-1, 2), rtol=2e-3, atol=2e-3, err_msg=str(c))
    assert n == 4


def test_affine_warp_and_channel_mix(oracle):
    """ADA's resampling step (affine_grid + grid_sample, training/augment.py:376-377) and colour matrix (:403-406):
    the oracle against the reference's own outputs and first-order input gradients."""
    cases, arr = load_golden('warp')
    for c in cases:
        i = c['id']
        if c['kind'] == 'warp':
            x, th = arr[f'w{i}_x'], arr[f'w{i}_theta']
            oh, ow = c['out']
            np.testing.assert_allclose(oracle.affine_warp(x, th, oh, ow), arr[f'w{i}_y'], rtol=1e-4, atol=2e-5, err_msg=str(c))
            np.testing.assert_allclose(oracle.affine_warp_bwd(arr[f'w{i}_gy'], th, x.shape[2], x.shape[3]), arr[f'w{i}_dx'],
                                       rtol=1e-4, atol=5e-5, err_msg=str(c))
            # float64 instantiation: same coordinates (float32 by construction), wider arithmetic
            np.testing.assert_allclose(oracle.affine_warp(x.astype(np.float64), th, oh, ow), arr[f'w{i}_y'], rtol=1e-4, atol=2e-5)
        else:
            x, C = arr[f'm{i}_x'], arr[f'm{i}_C']
            ch = c['channels']
            A = np.tile(np.eye(ch, dtype=np.float32), [x.shape[0], 1, 1])
            A[:, :3, :3] = C[:, :3, :3]
            t = np.zeros([x.shape[0], ch], dtype=np.float32)
            t[:, :3] = C[:, :3, 3]
            np.testing.assert_allclose(oracle.channel_mix(x, A, t), arr[f'm{i}_y'], **F32_TOL)
            np.testing.assert_allclose(oracle.channel_mix(arr[f'm{i}_gy'], np.ascontiguousarray(A.transpose(0, 2, 1))), arr[f'm{i}_dx'], **F32_TOL)
