"""AugmentPipe and the aio layout glue against the reference's outputs (tests/golden/augment.*):
deterministic debug_percentile images, seeded random images + input gradients, pseudo-fake centring
and per-layer crops.  CPU only (the GPU variant reuses these helpers)."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden, load_pkg

PKG = 'docker-montage-gan_amd'
SPECS = None


def mod(name):
    load_pkg()
    return importlib.import_module(f'{PKG}.{name}')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def run_augment_case(c, arr, device='cpu'):
    AU = mod('training.augment')
    spec = dict(AU.augpipe_specs['bgc'])
    if c['spec'] == 'all':
        spec.update(imgfilter=1, noise=1, cutout=1)
    pipe = AU.AugmentPipe(**spec).to(device)
    pipe.p.copy_(torch.as_tensor(c['p']))
    x = T(arr[c['x']]).to(device)
    if c['debug_percentile'] is not None:
        return pipe(x.clone(), debug_percentile=c['debug_percentile']), None
    torch.manual_seed(c['seed'])
    xr = x.clone().requires_grad_(True)
    y = pipe(xr)
    dx, = torch.autograd.grad([y], [xr], [T(arr[f'c{c["id"]}_dy']).to(device)])
    return y, dx


def test_augment_pipe_matches_reference():
    cases, arr = load_golden('augment')
    GS = mod('torch_utils.ops.grid_sample_gradfix')
    n = 0
    for enabled in (False, True):
        GS.enabled = enabled
        try:
            for c in cases:
                if c['op'] != 'augment':
                    continue
                y, dx = run_augment_case(c, arr)
                np.testing.assert_allclose(y.detach().numpy(), arr[f'c{c["id"]}_y'], rtol=1e-4, atol=2e-5, err_msg=str(c))
                if dx is not None:
                    np.testing.assert_allclose(dx.numpy(), arr[f'c{c["id"]}_dx'], rtol=1e-4, atol=2e-5, err_msg=str(c))
                n += 1
        finally:
            GS.enabled = False
    assert n == 30


def _layout_glue(device):
    IU = mod('custom_utils.image_utils')
    cases, arr = load_golden('augment')
    c = [c for c in cases if c['op'] == 'layout'][0]
    lay = (T(arr['lay']).float() / 255).to(device)
    pf = IU.generate_pseudo_fake(lay)
    assert np.array_equal((pf * 255).round().to(torch.uint8).cpu().numpy(), arr['pseudo_fake_u8'])
    outs = IU.make_batch_for_local_d(lay, c['sizes'], to_minus11=True)
    for k, o in enumerate(outs):
        np.testing.assert_allclose(o.cpu().numpy(), arr[f'local_d_{k}'], rtol=0, atol=1e-7)
    # single-image helpers agree with the batched path
    crop = IU.crop_to_content(lay[0, 0])
    assert crop.shape == (4, 29, 59)            # the max row/column is excluded, as upstream
    np.testing.assert_allclose(IU.pad_256(crop).cpu().numpy(), pf[0, 0].cpu().numpy(), atol=0)
    assert IU.crop_to_content(lay[1, 1]).shape == (4, 0, 0)        # empty layer
    # make_batch_for_pos_estimator / pad_256 (image_utils.py:216-243) against the reference's stack
    _, marr = load_golden('montage')
    pe = IU.make_batch_for_pos_estimator([T(marr['pe_in0']).to(device), T(marr['pe_in1']).to(device)], pad_value=-1)
    assert list(pe.shape) == list(marr['pe_out_shape'])
    np.testing.assert_allclose(pe[:, :, :, 112:144, 112:144].cpu().numpy(), marr['pe_out_corner'], atol=0)
    np.testing.assert_allclose(float(pe.double().sum()), float(marr['pe_out_sum']), rtol=1e-12)


def test_layout_glue_matches_reference():
    _layout_glue('cpu')


@pytest.mark.gpu
def test_layout_glue_matches_reference_on_gpu():
    """The batched device gathers of generate_pseudo_fake / make_batch_for_local_d / make_batch_for_pos_estimator on the
    MI355X, bit-for-bit against the reference's per-sample python loops (no host synchronisation in the batched path)."""
    _layout_glue('cuda:0')


def test_fixed_margin_mode_gives_the_same_images():
    """augment.fixed_margin pads by the clamp bound instead of reading the margin back: same output."""
    AU = mod('training.augment')
    cases, arr = load_golden('augment')
    n = 0
    for c in cases:
        if c['op'] != 'augment' or c['spec'] != 'bgc':
            continue
        AU.fixed_margin = True
        try:
            y, dx = run_augment_case(c, arr)
        finally:
            AU.fixed_margin = False
        np.testing.assert_allclose(y.detach().numpy(), arr[f'c{c["id"]}_y'], rtol=1e-4, atol=5e-5, err_msg=str(c))
        if dx is not None:
            np.testing.assert_allclose(dx.numpy(), arr[f'c{c["id"]}_dx'], rtol=1e-4, atol=5e-5, err_msg=str(c))
        n += 1
    assert n == 10


def test_table_driven_draws_reproduce_the_draw_by_draw_path():
    """AugmentPipe.params_from_tables (the formulation ops.ada_params mirrors in one native launch) consumes its tables in
    the order the reference-ordered code draws: record the fresh draws of one call, lay them out as tables, run again."""
    AU = mod('training.augment')

    class Recorder(AU._Draws):
        def __init__(self, dev):
            super().__init__(dev)
            self.u, self.z = [], []

        def rand(self, *shape):
            v = super().rand(*shape)
            self.u.append(v.reshape(shape[0], -1))
            return v

        def randn(self, *shape):
            v = super().randn(*shape)
            self.z.append(v.reshape(shape[0], -1))
            return v
    for spec, ch in (('bgc', 4), ('bgc', 3), ('bg', 4), ('color', 1), ('blit', 4), ('geom', 3)):
        pipe = AU.AugmentPipe(**AU.augpipe_specs[spec])
        pipe.p.copy_(torch.as_tensor(0.7))
        torch.manual_seed(3)
        n, h, w = 6, 32, 48
        rec = Recorder(torch.device('cpu'))
        G_inv, I_3, margin = pipe._geometric_params(n, h, w, torch.device('cpu'), None, rec)
        I_4 = torch.eye(4)
        C = pipe._color_matrix(n, ch, torch.device('cpu'), None, I_4, rec)
        U = torch.cat(rec.u, dim=1) if rec.u else torch.zeros([n, 0])
        Z = torch.cat(rec.z, dim=1) if rec.z else torch.zeros([n, 0])
        assert U.shape[1] <= 21 and Z.shape[1] <= 7
        if spec == 'bgc' and ch == 4:
            assert U.shape[1] == 21 and Z.shape[1] == 7        # ops.ada_params.NU / NZ
        G2, m2, C2 = pipe.params_from_tables(U, Z, n, ch, h, w)
        assert (G2 is None) == (G_inv is I_3) and (C2 is None) == (C is I_4)
        if G2 is not None:
            assert torch.equal(G2, G_inv) and torch.equal(m2, margin)
        if C2 is not None:
            assert torch.equal(C2, C)
