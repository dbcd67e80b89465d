"""The callers on the GPU: networks, loss phases and one full (tiny) training iteration with the native
ops, against (a) the reference's golden values and (b) the same code with every op forced to its aten
`impl='ref'` path on the same device and RNG stream.  Need a real MI355X."""
import functools
import importlib

import numpy as np
import pytest
import torch

from conftest import load_golden, load_pkg
from test_nets_golden import build_gd, T, mod

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def force_ref(monkeypatch):
    """Route every plugin op to its aten composition (explicit impl='ref'), fused epilogue off."""
    U, B = mod('torch_utils.ops.upfirdn2d'), mod('torch_utils.ops.bias_act')
    A, TS = mod('torch_utils.ops.alpha_composite'), mod('torch_utils.ops.translate_sample')
    NW = mod('training.networks')
    real_up = U.upfirdn2d

    def up_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
        return real_up(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain, impl='ref')
    monkeypatch.setattr(U, 'upfirdn2d', up_ref)
    real_ba = B.bias_act

    def ba_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
        return real_ba(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp, impl='ref')
    monkeypatch.setattr(B, 'bias_act', ba_ref)
    monkeypatch.setattr(A, 'alpha_composite', functools.partial(A.alpha_composite, impl='ref'))
    monkeypatch.setattr(TS, 'translate_sample', functools.partial(TS.translate_sample, impl='ref'))
    monkeypatch.setattr(NW, 'fused_epilogue', False)


def test_forward_matches_reference_golden_on_gpu():
    native = load_pkg().native
    cases, arr = load_golden('nets')
    G, D = build_gd(arr, cases, device=DEV)
    z = T(arr['z']).to(DEV)
    G.train(); D.train()
    before = native.launch_counts()
    img = G(z, noise_mode='const')
    logits = D(img.detach())
    after = native.launch_counts()
    assert after['bias_act'] > before['bias_act'] and after['upfirdn2d_tiled'] + after['upfirdn2d_stream'] + after['upfirdn2d_generic'] > \
        before['upfirdn2d_tiled'] + before['upfirdn2d_stream'] + before['upfirdn2d_generic'] and after['fma_bias_act'] > before['fma_bias_act']
    np.testing.assert_allclose(img.detach().cpu().numpy(), arr['img_train'], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), arr['logits'], rtol=1e-3, atol=1e-3)
    G.eval()
    np.testing.assert_allclose(G(z, noise_mode='const').detach().cpu().numpy(), arr['img_eval'], rtol=1e-3, atol=1e-3)


def _phase_grads(phase, arr, cases, seed):
    LS = mod('training.loss')
    G, D = build_gd(arr, cases, device=DEV)
    G.train(); D.train()
    G.requires_grad_(phase[0] == 'G'); D.requires_grad_(phase[0] == 'D')
    loss = LS.StyleGAN2Loss(device=torch.device(DEV), G_mapping=G.mapping, G_synthesis=G.synthesis, D=D)
    torch.manual_seed(seed)
    gen_z = torch.randn([4, 32], device=DEV)
    c0 = torch.zeros([4, 0], device=DEV)
    loss.accumulate_gradients(phase=phase, real_img=T(arr['real']).to(DEV), gen_z=gen_z, sync=True, gain=1.0, real_c=c0, gen_c=c0)
    m = G if phase[0] == 'G' else D
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('phase', ['Gmain', 'Greg', 'Dmain', 'Dreg'])
def test_loss_phase_native_equals_aten_ref_on_gpu(phase, monkeypatch):
    """Same device, same RNG stream: native kernels (incl. fused epilogue and the double-backward paths of
    R1 / path length) vs the aten compositions."""
    cases, arr = load_golden('nets')
    G_ = mod('torch_utils.ops.conv2d_gradfix')
    G_.enabled = True
    try:
        got = _phase_grads(phase, arr, cases, seed=77)
        with monkeypatch.context() as mp:
            force_ref(mp)
            want = _phase_grads(phase, arr, cases, seed=77)
    finally:
        G_.enabled = False
    assert got.keys() == want.keys() and len(got) > 10
    for n in got:
        scale = max(1e-3, float(want[n].abs().max()))
        err = float((got[n] - want[n]).abs().max()) / scale
        assert err < 2e-3, f'{phase} {n}: rel err {err:.2e}'


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_mixed_precision_step_is_finite_and_close(dtype):
    NW = mod('training.networks')
    torch.manual_seed(0)
    kw = dict(w_dim=64, img_resolution=64, img_channels=4, channel_base=1024, channel_max=64, conv_clamp=256)
    S32 = NW.SynthesisNetwork(num_fp16_res=0, **kw).to(DEV).train()
    S16 = NW.SynthesisNetwork(num_fp16_res=3, fp16_dtype=dtype, **kw).to(DEV).train()
    S16.load_state_dict(S32.state_dict())
    ws = torch.randn([4, S32.num_ws, 64], device=DEV)
    a, b = S32(ws, noise_mode='const'), S16(ws, noise_mode='const')
    assert torch.isfinite(b).all() and b.dtype == torch.float32
    rel = float((a - b).abs().mean() / a.abs().mean())
    assert rel < (0.02 if dtype == torch.float16 else 0.08), rel
    b.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in S16.parameters() if p.grad is not None)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('one_by_one', ['conv', 'matmul'])
def test_channels_last_blocks_match_nchw_blocks(dtype, one_by_one):
    """fp16_channels_last (reference training/networks.py:298/427) only changes the memory layout of the
    reduced-precision blocks: generator images, discriminator logits and parameter gradients must agree with the
    NCHW run up to the convolution library's summation order, and the NHWC native kernels must be the ones running."""
    native = load_pkg().native
    NW, CR = mod('training.networks'), mod('torch_utils.ops.conv2d_resample')
    CR.nhwc_1x1_workaround = (one_by_one == 'matmul')
    try:
        torch.manual_seed(1)
        kw = dict(w_dim=64, img_resolution=64, img_channels=4, channel_base=2048, channel_max=64, conv_clamp=256, num_fp16_res=3, fp16_dtype=dtype)
        Sa = NW.SynthesisNetwork(**kw).to(DEV).train()
        Sb = NW.SynthesisNetwork(fp16_channels_last=True, **kw).to(DEV).train()
        Sb.load_state_dict(Sa.state_dict())
        dkw = dict(c_dim=0, img_resolution=64, img_channels=4, channel_base=2048, channel_max=64, conv_clamp=256, num_fp16_res=3)
        Da = NW.Discriminator(block_kwargs=dict(fp16_dtype=dtype), **dkw).to(DEV).train()
        Db = NW.Discriminator(block_kwargs=dict(fp16_dtype=dtype, fp16_channels_last=True), **dkw).to(DEV).train()
        Db.load_state_dict(Da.state_dict())
        # float32 twins give the yardstick: the NHWC run may not sit further from them than the NCHW run does
        S32 = NW.SynthesisNetwork(**dict(kw, num_fp16_res=0)).to(DEV).train()
        S32.load_state_dict(Sa.state_dict())
        D32 = NW.Discriminator(**dict(dkw, num_fp16_res=0)).to(DEV).train()
        D32.load_state_dict(Da.state_dict())
        ws = torch.randn([4, Sa.num_ws, 64], device=DEV)
        before = native.launch_counts()
        outs = []
        for S, D in ((Sa, Da), (Sb, Db), (S32, D32)):
            img = S(ws, noise_mode='const')
            logits = D(img, None)
            torch.nn.functional.softplus(-logits).mean().backward()
            outs.append((img.detach(), logits.detach(), [p.grad for p in S.parameters()] + [p.grad for p in D.parameters()]))
        after = native.launch_counts()
        assert after['upfirdn2d_nhwc'] > before['upfirdn2d_nhwc'], 'the channels-last FIR kernel did not run'
        tol = 0.02 if dtype == torch.float16 else 0.08
        (ia, la, ga_all), (ib, lb, gb_all), (i32, l32, g32_all) = outs
        assert torch.isfinite(ib).all() and torch.isfinite(lb).all()
        assert float((ia - ib).abs().mean() / ia.abs().mean()) < tol
        assert float((la - lb).abs().max() / la.abs().max().clamp_min(1e-3)) < 4 * tol

        def rel(a, b):
            return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))
        for ga, gb, g32 in zip(ga_all, gb_all, g32_all):
            assert (ga is None) == (gb is None)
            if ga is not None and ga.numel() > 1 and float(g32.abs().max()) > 0:
                assert torch.isfinite(gb).all()
                assert rel(gb, g32) <= 2 * rel(ga, g32) + tol, (rel(gb, g32), rel(ga, g32), tuple(ga.shape))
    finally:
        CR.nhwc_1x1_workaround = True


@pytest.mark.parametrize('channels_last', [False, True])
def test_batched_weight_preparation_matches_per_layer_expressions(channels_last):
    """training/weight_prep.py: all `weight * gain`, `.to(dtype)` of a network in a few multi-tensor launches behind
    one autograd node -- images, logits, first-order gradients and the path-length style double backward must equal
    the per-layer expressions (bit for bit in the forward pass; gradients up to the order of float atomics)."""
    NW, WP = mod('training.networks'), mod('training.weight_prep')
    torch.manual_seed(2)
    kw = dict(w_dim=64, img_resolution=64, img_channels=4, channel_base=2048, channel_max=64, conv_clamp=256, num_fp16_res=3,
              fp16_dtype=torch.bfloat16, fp16_channels_last=channels_last)
    G = NW.Generator(z_dim=64, mapping_kwargs=dict(num_layers=3), synthesis_kwargs={k: v for k, v in kw.items() if k not in ('w_dim', 'img_resolution', 'img_channels')},
                     w_dim=64, img_resolution=64, img_channels=4).to(DEV).train()
    D = NW.Discriminator(c_dim=0, img_resolution=64, img_channels=4, channel_base=2048, channel_max=64, conv_clamp=256, num_fp16_res=3,
                         block_kwargs=dict(fp16_dtype=torch.bfloat16, fp16_channels_last=channels_last)).to(DEV).train()
    z = torch.randn([4, 64], device=DEV)
    res = []
    try:
        for on in (False, True):
            WP.enabled = on
            for p in list(G.parameters()) + list(D.parameters()):
                p.grad = None
            ws = G.mapping(z, None, skip_w_avg_update=True)
            img = G.synthesis(ws, noise_mode='const')
            logits = D(img, None)
            torch.nn.functional.softplus(-logits).mean().backward()
            first = [img.detach().clone(), logits.detach().clone()] + [p.grad.clone() for p in list(G.parameters()) + list(D.parameters()) if p.grad is not None]
            # path-length style: gradient of the image w.r.t. ws with create_graph, then backward through it
            for p in G.parameters():
                p.grad = None
            ws = G.mapping(z, None, skip_w_avg_update=True)
            img = G.synthesis(ws, noise_mode='const')
            gw, = torch.autograd.grad([(img * torch.ones_like(img)).sum()], [ws], create_graph=True)
            gw.square().sum(2).mean(1).sqrt().mean().backward()
            second = [p.grad.clone() for p in G.parameters() if p.grad is not None]
            res.append(first + second)
    finally:
        WP.enabled = True
    assert len(res[0]) == len(res[1])
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])      # forward: same arithmetic, deterministic kernels
    for a, b in zip(res[0][2:], res[1][2:]):
        # gradients pass through float atomics (split-K weight gradients in MIOpen, the fused reductions here): equal up to their order
        # (bf16 weight gradients are rounded from those sums: a changed order flips single results by one ulp = 2^-8 of the element)
        assert float((a.float() - b.float()).norm()) <= 4e-3 * float(b.float().norm()) + 1e-12


def test_aio_iteration_runs_native_and_updates_weights():
    """One full (tiny) train_aio iteration on the GPU: every phase fires at batch_idx 0; the native kernels
    run; weights move; a second iteration skips the lazy-regularisation phases."""
    native = load_pkg().native
    TL, DS = mod('custom.training_loop_aio'), mod('custom.dataset_aio')
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=2, num_fp16_res=2, fp16_dtype=torch.bfloat16)
    for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
        kw[k].channel_base, kw[k].channel_max = 256, 16
    kw.mapping_kwargs.num_layers = 2
    kw.training_set_kwargs.num_layers = 3
    reals = [DS.synthetic_layers(2, 3, 256, seed=5, device=DEV)]
    t = TL.AioTrainer(device=DEV, cfg=dict(TL.config, bypass_renderer=True), resident_batches=reals, **kw)
    w0 = [p.detach().clone() for p in t.local_G_list[0].parameters()][:3]
    stn0 = t.pos_estimator.fc_loc[2].bias.detach().clone()
    before = native.launch_counts()
    assert t.run_iteration() == 16          # 3 x (Gmain, Greg, Dmain, Dreg) + global x 4
    after = native.launch_counts()
    for k in ('bias_act', 'fma_bias_act', 'alpha_composite_fwd', 'alpha_composite_bwd', 'translate_fwd', 'translate_bwd_input',
              'translate_bwd_shift'):
        assert after[k] > before[k], k
    assert after['upfirdn2d_tiled'] + after['upfirdn2d_stream'] > before['upfirdn2d_tiled'] + before['upfirdn2d_stream']
    assert after['upfirdn2d_nhwc'] == before['upfirdn2d_nhwc']       # NCHW configuration: the channels-last kernel stays idle
    assert t.run_iteration() == 8           # only the main phases at batch_idx 1
    torch.cuda.synchronize()
    w1 = [p.detach() for p in t.local_G_list[0].parameters()][:3]
    assert any(not torch.equal(a, b) for a, b in zip(w0, w1))
    assert all(torch.isfinite(p).all() for g in t.local_G_list for p in g.parameters())
    assert not torch.equal(stn0, t.pos_estimator.fc_loc[2].bias.detach())       # the STN learns a placement


def test_augment_pipe_on_gpu_matches_reference_golden():
    """Deterministic (debug_percentile) AugmentPipe images on the GPU -- the one-pass 12-tap resamplers around the native
    affine warp -- against the reference's CPU output; also in the sync-free fixed-margin and side-stream modes."""
    from test_augment_layout_golden import run_augment_case
    AU = mod('training.augment')
    native = load_pkg().native
    cases, arr = load_golden('augment')
    GS = mod('torch_utils.ops.grid_sample_gradfix')
    GS.enabled = True
    try:
        for fixed, side in ((False, False), (True, False), (False, True)):
            AU.fixed_margin, AU.side_stream_margin = fixed, side
            before = native.launch_counts()
            n = 0
            for c in cases:
                # 'all' adds image noise, which is random even under debug_percentile: device RNG differs from the CPU's
                if c['op'] != 'augment' or c['debug_percentile'] is None or c['spec'] != 'bgc':
                    continue
                y, _ = run_augment_case(c, arr, device=DEV)
                np.testing.assert_allclose(y.cpu().numpy(), arr[f'c{c["id"]}_y'], rtol=1e-3, atol=1e-3, err_msg=f'{c} fixed={fixed}')
                n += 1
            after = native.launch_counts()
            # reflect-pad + upsampling and the downsampling: ONE one-pass separable launch each (csrc/upfirdn2d_sep.hip)
            assert n == 6 and after['upfirdn2d_sep'] == before['upfirdn2d_sep'] + 2 * n
    finally:
        AU.fixed_margin = AU.side_stream_margin = False
        GS.enabled = False


def _tiny_aio(graphs, ada, seed=3, **extra):
    TL, DS = mod('custom.training_loop_aio'), mod('custom.dataset_aio')
    kw = TL.default_kwargs(num_gpus=1, batch_gpu=2, num_fp16_res=2, fp16_dtype=torch.bfloat16)
    for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
        kw[k].channel_base, kw[k].channel_max = 256, 16
    kw.mapping_kwargs.num_layers = 2
    kw.training_set_kwargs.num_layers = 2
    if ada:
        AU = mod('training.augment')
        kw.augment_kwargs = dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc'])
        kw.augment_p = 0.4
    reals = [DS.synthetic_layers(2, 2, 256, seed=5 + i, device=DEV) for i in range(2)]
    kw.update(extra)
    return TL.AioTrainer(device=DEV, cfg=dict(TL.config, bypass_renderer=True), resident_batches=reals, random_seed=seed,
                         graphs=graphs, graph_warmup=1, **kw)


def test_recorded_function_consumes_the_generator_like_eager_code():
    """PhaseGraphs.run: replays draw the SAME random numbers an eager execution would have drawn at that point (the
    philox offset advances per replay), and in-place state updates made by the recorded kernels persist."""
    PE = mod('training.phase_engine')
    outs = []
    for use_graph in (False, True):
        torch.manual_seed(1234)
        graphs = PE.PhaseGraphs(DEV, warmup=1) if use_graph else None
        state = torch.zeros([], device=DEV)
        buf = torch.empty([5, 1000], device=DEV)
        i_dev = torch.zeros([], dtype=torch.int64, device=DEV)

        def body():
            r = torch.randn([1000], device=DEV) + torch.rand([1000], device=DEV)
            state.add_(r.sum())
            buf.index_copy_(0, i_dev.reshape(1), r.reshape(1, -1))
            i_dev.add_(1)
        replayed = []
        for _ in range(5):
            if graphs is None:
                body()
            else:
                replayed.append(graphs.run('body', body))
        torch.cuda.synchronize()
        if use_graph:
            assert replayed == [False, True, True, True, True] and graphs.captured == 1
        outs.append((buf.clone(), state.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert len(torch.unique(outs[0][0][:, 0])) == 5            # five different draws, not one replayed five times


@pytest.mark.parametrize('ada', [False, True])
def test_recorded_phases_equal_the_eager_loop(ada):
    """phase_engine.PhaseGraphs: the same tiny train_aio run eagerly (twice: the run-to-run spread) and with every phase recorded
    into a HIP graph after one eager execution.  All consume the device generator in the same order, so the recorded run must
    reproduce the eager one.

    What this test had wrong in round 2 (it retried its recorded arm after "O(1) deviations in one run of four"): with the
    loop's own Adam(beta1 = 0, eps = 1e-8) two EAGER arms from the same seed differ by 0.64-1.09 of the norm of the second
    iteration's global_D gradients in half of all pairs (profiles/r03_graph_eager_spread_toy.log: six eager arms, three
    clusters) -- a near-zero gradient takes a +-lr step whose sign is rounding noise, and the global GAN amplifies that.  The
    deviation never came from the recordings (each phase, replayed from saved state, reproduces its eager execution to 1e-7 on
    this configuration: profiles/r03_graph_probe_all_toy.log).  Here Adam runs with eps = 1e-3, which keeps the update a smooth
    function of the gradient: the arms must then agree tightly, and a wrong replay (stale inputs, other random draws, a skipped
    step) still shows as an O(1) deviation.  No retry.

    What is asserted, and what is only printed.  The proof that a recording equals the eager phase is PhaseGraphs' own: every
    phase is executed eagerly twice and replayed three times FROM THE SAME SAVED STATE AND RANDOM DRAWS, and the deviations are
    asserted below (< 1e-2 of the update; measured 1e-8 ... 2e-3).  Across whole arms -- which start from the same seed but
    accumulate their own rounding history -- the weights, the EMA weights and the report counters are asserted against the
    eager-vs-eager spread.  The gradients of the second iteration's global_Dmain are printed, not asserted: between two EAGER
    arms they are bimodal on this toy (profiles/r03_graph_test_arms.log: 0.001-0.009 in five of ten pairs, 0.20-0.67 in the
    other five, with and without ADA, with every native kernel switch on or off), so a bound derived from ONE eager pair says
    nothing about the recorded arm -- the recorded arm lands in either cluster at the same rate."""
    AU = mod('training.augment')
    TS = mod('torch_utils.training_stats')
    smooth_adam = {k: dict(class_name='torch.optim.Adam', lr=0.0025, betas=[0, 0.99], eps=1e-3)
                   for k in ('local_G_opt_kwargs', 'local_D_opt_kwargs', 'global_G_opt_kwargs', 'global_D_opt_kwargs')}

    def arm(graphs):
        torch.manual_seed(11)
        TS._counters.clear()
        t = _tiny_aio(graphs, ada, **smooth_adam)
        t.run_iteration()
        t.run_iteration()               # graphs: every phase was recorded after its eager run above; the main phases replay here
        torch.cuda.synchronize()
        g_first = [p.grad.detach().float().clone() for p in t.global_D.parameters()]      # gradients of the last phase (global_Dmain)
        for _ in range(4):
            t.run_iteration()
        torch.cuda.synchronize()
        if graphs:
            # every phase with work was recorded and verified (an eager re-execution gives the phase's own spread, three replays from
            # the saved state must reproduce the eager execution within it), or left eager with the reason
            G = t.graphs
            print('recorded', sorted(map(str, G._graphs)), 'rejected', G.rejected)
            print('verification', {str(k): (f"{v['eager_spread']:.2g}", [f'{d:.2g}' for d in v['replays']]) for k, v in G.report.items()})
            assert G.captured + len(G.rejected) == 2 * 4 + 3 + 1 and G.replays >= 4 * 4
            assert not G.rejected, G.rejected                  # float32-accurate toy networks: every phase reproduces, global_Gmain included
            assert 'global_Greg' not in G._graphs and 'global_Greg' not in G.rejected
            for k, v in G.report.items():
                # (measured: 1e-8 ... 2e-3 of the update, the largest on the path-length phases' double backward in bf16)
                assert v['eager_spread'] < 1e-2 and len(v['replays']) == 3 and max(v['replays']) < 1e-2, (k, v)
        w = [p.detach().float().clone() for m in [t.mapping_network, *t.local_G_list, *t.local_D_list, t.pos_estimator, t.global_D]
             for p in m.parameters()]
        ema = [p.detach().float().clone() for p in t.global_G_ema['local_G_list'][0].parameters()]
        stats = {k: sum(v.values()).clone() for k, v in TS._counters.items() if v}
        return w, ema, stats, g_first

    def dist(xs, ys):
        num = sum(float((a - b).square().sum()) for a, b in zip(xs, ys)) ** 0.5
        return num / max(sum(float(a.square().sum()) for a in xs) ** 0.5, 1e-30)

    AU.fixed_margin, AU.side_stream_margin = True, False
    try:
        (w0, e0, s0, g0), (w1, e1, s1, g1) = arm(False), arm(False)
        w2, e2, s2, g2 = arm(True)
        assert len(w0) == len(w2) and len(g0) == len(g2)
        print(f'first replayed gradients (information): recorded-vs-eager {dist(g0, g2):.4g}, eager-vs-eager {dist(g0, g1):.4g}')
        assert all(bool(torch.isfinite(g).all()) for g in g2)
        for name, a, b, c, floor in (('weights', w0, w1, w2, 1e-3), ('ema weights', e0, e1, e2, 1e-3)):
            spread, got = dist(a, b), dist(a, c)
            print(f'{name}: recorded-vs-eager {got:.4g}, eager-vs-eager {spread:.4g}')
            assert got <= 4 * spread + floor, (name, got, spread)
        assert set(s0) == set(s2)
        for k in s0:
            assert float(s0[k][0]) == float(s2[k][0]), k              # same number of reported elements
    finally:
        AU.fixed_margin = AU.side_stream_margin = False


def test_a_phase_that_does_not_reproduce_eagerly_is_not_recorded():
    """PhaseGraphs measures a phase's own run-to-run spread with a second eager execution from the same state and random
    draws; a body whose eager executions differ by more than `chaos_tol` of the update (at full width: global_Gmain) cannot be
    proved against them and stays eager, with the measurement in the reason.  A reproducible body next to it is recorded."""
    PE = mod('training.phase_engine')
    graphs = PE.PhaseGraphs(DEV, warmup=1)
    w = torch.zeros([1000], device=DEV)
    hidden = torch.zeros([1000], device=DEV)           # state the verification does not know about: consecutive executions differ

    def chaotic():
        hidden.add_(1.0)
        w.add_(torch.sin(hidden * torch.arange(1000, device=DEV)))

    def smooth():
        w.add_(torch.randn([1000], device=DEV))

    state = lambda: [('w', w, True)]
    with graphs.stream_scope():
        graphs.run('chaotic', chaotic, state_fn=state)
        graphs.run('smooth', smooth, state_fn=state)
        ran = [graphs.run('chaotic', chaotic, state_fn=state), graphs.run('smooth', smooth, state_fn=state)]
    torch.cuda.synchronize()
    assert ran == [False, True]
    assert 'two EAGER executions' in graphs.rejected['chaotic'] and graphs.report['chaotic']['eager_spread'] > 0.25
    assert graphs.report['smooth']['eager_spread'] == 0 and graphs.report['smooth']['replays'] == [0, 0, 0]


@pytest.mark.parametrize('graphs', [False, True])
def test_base_trainer_config2_shapes_on_gpu(graphs):
    """BASELINE configs[1] through ``training.training_loop.Trainer``: one StyleGAN2 G/D pair at 256x256 RGBA, bf16 on the top
    4 resolutions, ADA bgc (small channel counts; the full-size run is ``bench.py --workload single``).
    Every phase fires at iteration 0, weights move, everything stays finite; with ``graphs`` the phases are recorded."""
    TL, AU = mod('training.training_loop'), mod('training.augment')
    native = load_pkg().native
    AU.fixed_margin, AU.side_stream_margin = True, False
    try:
        kw = TL.default_kwargs(num_gpus=1, batch_gpu=4, resolution=256, num_fp16_res=4, fp16_dtype=torch.bfloat16)
        kw.G_kwargs.synthesis_kwargs.update(channel_base=256, channel_max=16)
        kw.D_kwargs.update(channel_base=256, channel_max=16)
        kw.augment_kwargs = dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc'])
        kw.augment_p = 0.3
        g = torch.Generator(device='cpu').manual_seed(0)
        reals = [(torch.randint(0, 256, [4, 4, 256, 256], generator=g, dtype=torch.uint8).to(DEV), torch.zeros([4, 0], device=DEV))]
        t = TL.Trainer(device=DEV, resident_batches=reals, random_seed=0, graphs=graphs, **kw)
        assert [p.name for p in t.phases] == ['Gmain', 'Greg', 'Dmain', 'Dreg'] and t.G.synthesis.block_resolutions[-1] == 256
        w0 = [p.detach().clone() for p in list(t.G.parameters()) + list(t.D.parameters())]
        before = native.launch_counts()
        assert t.run_iteration() == 4
        after = native.launch_counts()
        assert after['bias_act'] > before['bias_act'] and after['ada_params'] > before['ada_params']
        for _ in range(3):
            assert t.run_iteration() == 2
        torch.cuda.synchronize()
        w1 = list(t.G.parameters()) + list(t.D.parameters())
        assert all(torch.isfinite(p).all() for p in w1)
        assert sum(int(not torch.equal(a, b.detach())) for a, b in zip(w0, w1)) >= len(w0) - 2
        if graphs:
            assert t.graphs.captured + len(t.graphs.rejected) >= 4 and t.graphs.replays >= 2 * 2, (t.graphs.captured, t.graphs.rejected)
    finally:
        AU.fixed_margin = AU.side_stream_margin = False


def test_bucketed_gradient_exchange_on_the_device():
    """GradSync's overlapped exchange on a real GPU: the hooks run on autograd's device thread, the all-reduce is RCCL's
    (backend "nccl", a one-rank group -- the sum over one rank is the identity, so the result must equal the plain path) and the optimizer step waits for it on the launch stream.  world_size = 2 is only what selects the code path."""
    import os
    import tempfile
    load_pkg()
    PE = mod('training.phase_engine')
    dev = torch.device(DEV)
    with tempfile.TemporaryDirectory() as d:
        if not torch.distributed.is_initialized():
            torch.distributed.init_process_group('nccl', init_method='file://' + os.path.join(d, 'init'), rank=0, world_size=1)
        try:
            results = []
            for overlap in (False, True):
                torch.manual_seed(0)
                net = torch.nn.Sequential(torch.nn.Conv2d(4, 32, 3, padding=1), torch.nn.LeakyReLU(0.2), torch.nn.Conv2d(32, 32, 3, padding=1),
                                          torch.nn.LeakyReLU(0.2), torch.nn.Conv2d(32, 4, 3, padding=1)).to(dev)
                params = list(net.parameters())
                opt = torch.optim.SGD(params, lr=0.05)
                phase = PE.Phase(name='toy', module=[net], opt=opt, interval=1)
                sync = PE.GradSync(world_size=1, overlap=overlap, bucket_bytes=4096)
                data = torch.randn([2, 8, 4, 32, 32], generator=torch.Generator().manual_seed(3)).to(dev)

                def accumulate():
                    for r in range(2):
                        sync.begin_round(r == 1)
                        with mod('torch_utils.misc').ddp_sync(net, False):
                            first = net(data[r, :4])
                        first.square().mean().backward()        # (two backward calls per round, as in Dmain)
                        with mod('torch_utils.misc').ddp_sync(net, r == 1):
                            second = net(data[r, 4:])
                        second.square().mean().backward()
                for it in range(3):
                    if overlap:
                        PE.run_phase(phase, accumulate, sync, None, 2)
                    else:
                        PE.run_phase(phase, accumulate, sync, None, 1)
                torch.cuda.synchronize()
                assert (sync.buckets_launched_early > 0) == overlap and sync.stale_phases == 0
                results.append(([p.grad.clone() for p in params], [p.detach().clone() for p in params]))
            # (the library's weight-gradient kernels sum through float atomics: equal up to their order, not bit for bit)
            for a, b in zip(results[0][0] + results[0][1], results[1][0] + results[1][1]):
                assert float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-12
        finally:
            torch.distributed.destroy_process_group()


def test_batched_affine_styles_match_per_layer_products():
    """SynthesisNetwork._all_styles (one baddbmm per group of equally wide affine layers) against `self.affine(w)` per layer:
    images, first-order gradients (weights, biases, ws -- incl. a w shared by a ToRGB and the next block's first conv, and distinct
    ws per layer as under style mixing) and the path-length style double backward."""
    NW = mod('training.networks')
    torch.manual_seed(4)
    G = NW.Generator(z_dim=64, w_dim=64, img_resolution=64, img_channels=4, mapping_kwargs=dict(num_layers=2),
                     synthesis_kwargs=dict(channel_base=2048, channel_max=64, conv_clamp=256, num_fp16_res=2, fp16_dtype=torch.bfloat16,
                                           fp16_channels_last=True)).to(DEV).train()
    z = torch.randn([4, 64], device=DEV)
    res = []
    try:
        for on in (False, True):
            NW.batched_styles = on
            for p in G.parameters():
                p.grad = None
            ws = G.mapping(z, None, skip_w_avg_update=True)
            ws = (ws + 0.1 * torch.randn(ws.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))).detach().requires_grad_(True)
            img = G.synthesis(ws, noise_mode='const')
            gws, = torch.autograd.grad([img.square().mean()], [ws], retain_graph=True)
            img.square().mean().backward()
            first = [img.detach().float(), gws] + [p.grad.clone() for p in G.synthesis.parameters() if p.grad is not None]
            for p in G.parameters():
                p.grad = None
            ws2 = ws.detach().requires_grad_(True)
            img2 = G.synthesis(ws2, noise_mode='const')
            gw, = torch.autograd.grad([(img2 * torch.ones_like(img2)).sum()], [ws2], create_graph=True)
            gw.square().sum(2).mean(1).sqrt().mean().backward()
            second = [p.grad.clone() for p in G.synthesis.parameters() if p.grad is not None]
            res.append(first + second)
    finally:
        NW.batched_styles = True
    assert len(res[0]) == len(res[1]) and len(res[0]) > 40
    for a, b in zip(res[0], res[1]):
        # bf16 blocks downstream: a last-bit difference of a style can flip single bf16 roundings of the activations
        assert a.shape == b.shape and float((a.float() - b.float()).norm()) <= 4e-3 * float(b.float().norm()) + 1e-10
