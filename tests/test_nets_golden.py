"""The callers (networks + loss phases) of this package against golden values produced by the
reference's own networks/loss on the CPU with the reference's weights (tests/golden/nets.*).
CPU only; the GPU variants live in test_train_step_gpu.py."""
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden, load_pkg

PKG = 'docker-montage-gan_amd'


def mod(name):
    load_pkg()
    return importlib.import_module(f'{PKG}.{name}')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_sd(module, arr, prefix):
    sd = {k[len(prefix) + 1:]: T(arr[k]) for k in arr.files if k.startswith(prefix + '.')}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def build_gd(arr, cases, device='cpu', **over):
    NW = mod('training.networks')
    c0 = cases[0]
    gk = dict(c0['G_kwargs']); dk = dict(c0['D_kwargs'])
    gk['synthesis_kwargs'] = dict(gk['synthesis_kwargs'], **over)
    G, D = NW.Generator(**gk), NW.Discriminator(**dk)
    load_sd(G, arr, 'G'); load_sd(D, arr, 'D')        # same parameter/buffer names as the reference => strict load
    return G.to(device), D.to(device)


def test_generator_discriminator_forward_match_reference():
    cases, arr = load_golden('nets')
    G, D = build_gd(arr, cases)
    z = T(arr['z'])
    G.train(); D.train()
    img = G(z, noise_mode='const')
    np.testing.assert_allclose(img.detach().numpy(), arr['img_train'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(D(img.detach()).detach().numpy(), arr['logits'], rtol=1e-4, atol=1e-4)
    G.eval()
    np.testing.assert_allclose(G(z, noise_mode='const').detach().numpy(), arr['img_eval'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(G(z, truncation_psi=0.5, noise_mode='const').detach().numpy(), arr['img_trunc'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(G.mapping.w_avg.numpy(), arr['w_avg_after'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('pi', range(6))
def test_loss_phase_gradients_match_reference(pi):
    """Same seed => same random draws (style mixing, per-layer noise, pl noise) => same gradients."""
    LS, G_ = mod('training.loss'), mod('torch_utils.ops.conv2d_gradfix')
    cases, arr = load_golden('nets')
    c = cases[1 + pi]
    G, D = build_gd(arr, cases)
    G.train(); D.train()
    phase = c['phase']
    G.requires_grad_(phase[0] == 'G'); D.requires_grad_(phase[0] == 'D')
    loss = LS.StyleGAN2Loss(device=torch.device('cpu'), G_mapping=G.mapping, G_synthesis=G.synthesis, D=D)
    G_.enabled = True          # exercise the hand-written conv autograd, incl. no_weight_gradients()
    try:
        torch.manual_seed(c['seed'])
        gen_z = torch.randn([4, 32])
        assert np.array_equal(gen_z.numpy(), arr[f'p{pi}_gen_z'])
        loss.accumulate_gradients(phase=phase, real_img=T(arr['real']), gen_z=gen_z, sync=True, gain=c['gain'],
                                  real_c=torch.zeros([4, 0]), gen_c=torch.zeros([4, 0]))
    finally:
        G_.enabled = False
    m = G if phase[0] == 'G' else D
    params = dict(m.named_parameters())
    got_names = [n for n, p in m.named_parameters() if p.grad is not None]
    assert got_names == c['grad_names']
    sums = np.asarray([float(params[n].grad.double().sum()) for n in got_names])
    abss = np.asarray([float(params[n].grad.double().abs().sum()) for n in got_names])
    np.testing.assert_allclose(abss, arr[f'p{pi}_grad_abs'], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(sums, arr[f'p{pi}_grad_sum'], rtol=2e-3, atol=2e-3 * np.abs(arr[f'p{pi}_grad_abs']).max())
    np.testing.assert_allclose(params[c['sample_name']].grad.numpy(), arr[f'p{pi}_grad_sample'], rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(float(loss.pl_mean), float(arr[f'p{pi}_pl_mean']), rtol=1e-4, atol=1e-7)


def test_aio_variants_match_reference():
    NA = mod('custom.networks_aio')
    cases, arr = load_golden('nets')
    c = cases[7]
    S = NA.SynthesisNetwork(**c['S_kwargs'])
    M = NA.GlobalMappingNetwork(z_dim=32, w_dim=32, num_ws=c['num_ws'], num_splits=3, num_layers=2)
    DA = NA.Discriminator(**c['D_kwargs'])
    load_sd(S, arr, 'S'); load_sd(M, arr, 'M'); load_sd(DA, arr, 'DA')
    assert S.num_ws == c['num_ws'] and S.block_resolutions == [8, 16, 32, 64]
    M.train(); S.train(); DA.train()
    ws = M(T(arr['aio_z']))
    np.testing.assert_allclose(ws.detach().numpy(), arr['aio_ws'], rtol=1e-5, atol=1e-5)
    imgs = torch.stack([S(ws[:, li], noise_mode='const') for li in range(3)])
    np.testing.assert_allclose(imgs.detach().numpy(), arr['aio_imgs'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(DA(imgs[1].detach()).detach().numpy(), arr['aio_logits'], rtol=1e-4, atol=1e-4)
