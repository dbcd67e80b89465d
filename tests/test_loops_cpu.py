"""The loops themselves on the CPU: BASELINE config 1 through ``training.training_loop.Trainer``, the phase machinery's
arithmetic (lazy-regularisation rescale, EMA, ADA update) as known-answer tests of the formulas the reference states inline
(training/training_loop.py:211-222,319-326,333-337; custom/training_loop_aio.py:672-682 -- the loops cannot be imported
here: they hard-code torch.device('cuda')), snapshot -> resume round trips under the reference's key names
(custom/training_loop_aio.py:345-366,809-833) and the on-disk layer dataset."""
import math
import os

import numpy as np
import pytest
import torch

from test_nets_golden import mod


def test_config1_single_gan_64px_batch4_cpu():
    """BASELINE configs[0]: one StyleGAN2 at 64x64 RGBA, batch 4, on the CPU through the aten ``impl='ref'`` ops --
    full channel counts (channel_base 16384), every phase of iteration 0, then the main phases of iteration 1."""
    TL = mod('training.training_loop')
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=4, resolution=64, num_fp16_res=0)
    kw.training_set_kwargs.num_samples = 16
    kw.data_loader_kwargs = dict(num_workers=0)
    t = TL.Trainer(device='cpu', random_seed=1, **kw)
    assert [p.name for p in t.phases] == ['Gmain', 'Greg', 'Dmain', 'Dreg']
    assert t.G.synthesis.block_resolutions == [4, 8, 16, 32, 64] and t.G.img_channels == 4
    w0 = {n: p.detach().clone() for n, p in list(t.G.named_parameters()) + list(t.D.named_parameters())}
    ema0 = t.G_ema.synthesis.b64.conv1.weight.detach().clone()
    assert t.run_iteration() == 4
    assert t.run_iteration() == 2 and t.cur_nimg == 8 and t.batch_idx == 2
    moved = [n for n, p in list(t.G.named_parameters()) + list(t.D.named_parameters()) if not torch.equal(w0[n], p.detach())]
    assert len(moved) == len(w0)                        # every parameter of G and D took an Adam step
    assert all(torch.isfinite(p).all() for p in list(t.G.parameters()) + list(t.D.parameters()))
    assert not torch.equal(ema0, t.G_ema.synthesis.b64.conv1.weight)
    assert float(t.loss.pl_mean) > 0                   # the path-length phase ran and updated its running mean


def test_lazy_regularisation_rescale_known_answers():
    """mb_ratio = i / (i + 1): lr * mb_ratio, beta ** mb_ratio, ONE optimizer for the main and reg phase (:211-222)."""
    PE, dnnlib = mod('training.phase_engine'), mod('dnnlib')
    net_g, net_d = torch.nn.Linear(3, 3), torch.nn.Linear(3, 3)
    kw = dnnlib.EasyDict(class_name='torch.optim.Adam', lr=0.0025, betas=[0, 0.99], eps=1e-8)
    phases = PE.build_phases([('G{}', net_g, net_g.parameters(), kw, 4, 1), ('D{}', net_d, net_d.parameters(), kw, 16, 1),
                              ('X{}', net_d, net_d.parameters(), kw, None, 1)], 'cpu')
    assert [(p.name, p.interval) for p in phases] == [('Gmain', 1), ('Greg', 4), ('Dmain', 1), ('Dreg', 16), ('Xboth', 1)]
    g, d, x = phases[0].opt, phases[2].opt, phases[4].opt
    assert phases[1].opt is g and phases[3].opt is d
    assert math.isclose(g.param_groups[0]['lr'], 0.0025 * 4 / 5, rel_tol=1e-12) and g.param_groups[0]['lr'] == 0.002
    assert g.param_groups[0]['betas'][0] == 0 and math.isclose(g.param_groups[0]['betas'][1], 0.99 ** 0.8, rel_tol=1e-12)
    assert math.isclose(d.param_groups[0]['lr'], 0.0025 * 16 / 17, rel_tol=1e-12)
    assert math.isclose(d.param_groups[0]['betas'][1], 0.99 ** (16 / 17), rel_tol=1e-12)
    assert x.param_groups[0]['lr'] == 0.0025 and list(x.param_groups[0]['betas']) == [0.0, 0.99]
    assert kw.lr == 0.0025 and kw.betas == [0, 0.99]           # the caller's kwargs are not modified


def test_ema_and_ada_known_answers():
    PE = mod('training.phase_engine')
    # beta = 0.5 ** (batch / max(ema_nimg, 1e-8)), ema_nimg = min(ema_kimg * 1000, cur_nimg * rampup)  (:319-322)
    assert math.isclose(PE.ema_beta_for(32, 123456, 10, None), 0.5 ** (32 / 10000), rel_tol=1e-15)
    assert math.isclose(PE.ema_beta_for(32, 1000, 10, 0.05), 0.5 ** (32 / 50), rel_tol=1e-15)           # ramp-up: 1000 * 0.05 = 50 images
    assert math.isclose(PE.ema_beta_for(32, 10 ** 7, 10, 0.05), 0.5 ** (32 / 10000), rel_tol=1e-15)     # past the ramp
    assert PE.ema_beta_for(32, 0, 10, 0.05) == 0.0                                                       # cur_nimg 0 -> copy
    # p_ema <- p_ema.lerp(p, 1 - beta) == p.lerp(p_ema, beta); buffers copied (:323-326)
    src, dst = torch.nn.BatchNorm1d(3), torch.nn.BatchNorm1d(3)
    with torch.no_grad():
        src.weight.copy_(torch.tensor([1.0, 2.0, 3.0])); dst.weight.copy_(torch.tensor([3.0, 2.0, -1.0]))
        src.running_mean.copy_(torch.tensor([5.0, 6.0, 7.0]))
    PE.update_ema(dst, src, 0.75)
    np.testing.assert_allclose(dst.weight.detach().numpy(), [1 + (3 - 1) * 0.75, 2.0, 3 + (-1 - 3) * 0.75], rtol=1e-6)
    assert torch.equal(dst.running_mean, src.running_mean)
    # ADA: p += sign(r_t - target) * batch * interval / (ada_kimg * 1000)  (:333-337); aio clamps to [0, aug_p_max] (:681-682)
    assert math.isclose(PE.ada_adjust(0.9, 0.6, 32, 4, 500), 32 * 4 / 500000, rel_tol=1e-15)
    assert math.isclose(PE.ada_adjust(0.1, 0.6, 32, 4, 500), -32 * 4 / 500000, rel_tol=1e-15)
    assert PE.ada_adjust(0.6, 0.6, 32, 4, 500) == 0.0


def _tiny_aio(seed, **extra):
    TL = mod('custom.training_loop_aio')
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=2, num_fp16_res=0)
    for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
        kw[k].channel_base, kw[k].channel_max = 1024, 8
    kw.mapping_kwargs.update(z_dim=16, w_dim=16, num_layers=1)
    kw.local_G_kwargs.w_dim = 16
    kw.training_set_kwargs.update(num_layers=2, num_samples=8)
    kw.data_loader_kwargs = dict(num_workers=0)
    kw.update(extra)
    return TL, TL.AioTrainer(device='cpu', random_seed=seed, cfg=dict(TL.config, bypass_renderer=True), **kw)


def test_aio_ada_update_is_clamped_to_aug_p_max():
    """The aio loop's ADA step end to end: statistics -> Collector -> p, with the [0, aug_p_max] clamp (:672-682)."""
    AU, TS = mod('training.augment'), mod('torch_utils.training_stats')
    TS._counters.clear()
    TL, t = _tiny_aio(0, augment_kwargs=dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc']),
                      augment_p=0.59995, ada_target=0.6, ada_interval=1, ada_kimg=0.1)
    assert t.cfg['aug_p_max'] == 0.6 and all(s is not None for s in t.ada_stats_list)
    t.batch_idx = 1                                     # skip the lazy-reg phases (keeps the test short); ADA fires every iteration
    # make D(real) overwhelmingly positive for layer 0 and negative for layer 1 through the statistics themselves
    real_report = TS.report

    def biased(name, value):
        if name.endswith('Loss/signs/real'):
            value = torch.ones_like(torch.as_tensor(value)) * (1.0 if name.startswith(t.layer_names[0]) or name.startswith('global') else -1.0)
        return real_report(name, value)
    for m in (mod('training.loss'), mod('custom.loss_aio')):
        m.training_stats.report = biased
    try:
        t.run_iteration()
    finally:
        for m in (mod('training.loss'), mod('custom.loss_aio')):
            m.training_stats.report = real_report
    step = 2 * 1 / (0.1 * 1000)                         # batch 2, interval 1, ada_kimg 0.1 -> 0.02 per update
    assert math.isclose(float(t.augment_pipe_list[0].p), 0.6, rel_tol=1e-6)            # 0.59995 + 0.02 clamped to aug_p_max
    assert math.isclose(float(t.augment_pipe_list[1].p), 0.59995 - step, rel_tol=1e-5)
    assert math.isclose(float(t.global_augment_pipe.p), 0.6, rel_tol=1e-6)


def test_aio_snapshot_resume_round_trip(tmp_path):
    """state_dicts() -> torch.save -> a differently initialised trainer -> load_state_dicts(): every module the reference
    resumes (:349-359) comes back, under the reference's snapshot names (:811-826)."""
    TL, a = _tiny_aio(1)
    snap = a.state_dicts()
    assert list(snap.keys()) == ['training_set_kwargs', 'mapping_network', 'local_G_list', 'local_D_list', 'mapping_network_ema',
                                 'local_G_ema', 'augment_pipe_list', 'pos_estimator', 'pos_estimator_ema', 'global_augment_pipe',
                                 'global_D']
    path = os.path.join(tmp_path, 'network-snapshot-000000.pth')
    torch.save(snap, path)
    _, b = _tiny_aio(2)
    assert not torch.equal(a.global_D.b4.out.weight, b.global_D.b4.out.weight)
    restored = b.load_state_dicts(torch.load(path, map_location='cpu', weights_only=True))
    assert restored == ['mapping_network', 'local_G_list', 'local_D_list', 'mapping_network_ema', 'local_G_ema', 'pos_estimator',
                        'pos_estimator_ema', 'global_D']
    pairs = [(a.mapping_network, b.mapping_network), (a.global_D, b.global_D), (a.pos_estimator, b.pos_estimator),
             (a.global_G_ema['mapping_network'], b.global_G_ema['mapping_network']),
             (a.global_G_ema['pos_estimator'], b.global_G_ema['pos_estimator'])]
    pairs += list(zip(a.local_G_list, b.local_G_list)) + list(zip(a.local_D_list, b.local_D_list))
    pairs += list(zip(a.global_G_ema['local_G_list'], b.global_G_ema['local_G_list']))
    for ma, mb in pairs:
        for (ka, va), (kb, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), ka
    # a missing entry is skipped, as upstream ("Skip resuming ... as no data is found")
    part = {k: v for k, v in snap.items() if k in ('local_D_list', 'training_set_kwargs')}
    assert b.load_state_dicts(part) == ['local_D_list']
    # a plain MappingNetwork state is tiled into the GlobalMappingNetwork where the sizes differ (:319-331)
    NA = mod('custom.networks_aio')
    plain = NA.MappingNetwork(z_dim=16, w_dim=16, c_dim=0, num_ws=a.mapping_network.num_ws, num_layers=1)
    assert b.load_state_dicts(dict(mapping_network=plain.state_dict())) == ['mapping_network']
    assert torch.equal(b.mapping_network.fc0.weight, plain.fc0.weight.repeat(2, 1))
    assert torch.equal(b.mapping_network.w_avg, plain.w_avg.repeat(2))


def test_base_loop_snapshot_resume(tmp_path):
    """training_loop(resume_pkl=...) restores G, D and G_ema from the state_dict snapshot the loop itself writes."""
    TL = mod('training.training_loop')
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=2, resolution=32, num_fp16_res=0)
    kw.G_kwargs.synthesis_kwargs.update(channel_base=256, channel_max=8)
    kw.D_kwargs.update(channel_base=256, channel_max=8)
    kw.G_kwargs.update(z_dim=16, w_dim=16)
    kw.training_set_kwargs.num_samples = 8
    kw.data_loader_kwargs = dict(num_workers=0)
    common = dict(device='cpu', total_kimg=0.002, kimg_per_tick=0.002, **kw)
    t1 = TL.training_loop(run_dir=str(tmp_path / 'a'), random_seed=3, network_snapshot_ticks=1, **common)
    snaps = sorted(f for f in os.listdir(tmp_path / 'a') if f.endswith('.pth'))
    assert snaps and os.path.exists(tmp_path / 'a' / 'stats.jsonl')
    t2 = TL.Trainer(device='cpu', random_seed=4, **kw)
    saved = torch.load(os.path.join(tmp_path / 'a', snaps[-1]), map_location='cpu', weights_only=True)
    assert set(saved) == {'G', 'D', 'G_ema'}
    t2.G.load_state_dict(saved['G'])
    assert all(torch.equal(x, y) for x, y in zip(t1.G.state_dict().values(), t2.G.state_dict().values()))


def test_on_disk_layer_dataset(tmp_path):
    """DatasetAIO: <root>/<sample>/<layer>.png, RGBA, sorted by file name (reference custom/dataset_aio.py:143-254) -- the
    attributes training_loop_aio reads, the [L,C,H,W] float sample in [0,1], and the x-flip doubling."""
    from PIL import Image
    DS = mod('custom.dataset_aio')
    rs = np.random.RandomState(0)
    names = ['00_body', '01_face', '02_hair']
    imgs = {}
    for s in ('s0', 's1'):
        os.makedirs(tmp_path / s)
        for n in names:
            a = rs.randint(0, 256, size=[64, 64, 4]).astype(np.uint8)
            Image.fromarray(a, 'RGBA').save(tmp_path / s / f'{n}.png')
            imgs[(s, n)] = a
    ds = DS.DatasetAIO(str(tmp_path), xflip=True, conv_config_index=3)
    assert len(ds) == 4 and ds.layer_names == names and ds.num_layers == 3 and ds.num_channels == 4
    assert ds.resolution == 64 and ds.res_log2 == 6 and list(ds.init_res) == [8, 8]
    assert ds.target_res_layer('01_face') == [64, 64] and ds.resolution_layer('02_hair') == 64 and list(ds.init_res_layer('00_body')) == [8, 8]
    x = ds[1]
    assert x.shape == (3, 4, 64, 64) and x.dtype == torch.float32 and 0 <= float(x.min()) and float(x.max()) <= 1
    np.testing.assert_array_equal((x[2] * 255).round().byte().permute(1, 2, 0).numpy(), imgs[('s1', '02_hair')])
    assert torch.equal(ds[3], torch.flip(x, [-1]))                # second half of the index range = mirrored copies
    smaller = DS.DatasetAIO(str(tmp_path), xflip=False, conv_config_index=3, layer_sizes=[[64, 64], [32, 32], [64, 32]])
    assert smaller.target_res_layer('01_face') == [32, 32] and len(smaller) == 2
