#!/usr/bin/env python3
"""Diagnostic: eager execution vs HIP-graph replay of pieces of the training step, from identical state."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'


def mod(n):
    importlib.import_module(PKG)
    return importlib.import_module(f'{PKG}.{n}')


DEV = torch.device('cuda', 0)


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-30))


def compare(name, fn, outs, n_replays=2):
    """fn writes its results into the tensors `outs` (list).  Eager from RNG state S vs graph replay from S."""
    torch.cuda.synchronize()
    fn()                                            # warm-up (lazy state, constants)
    torch.cuda.synchronize()
    S = torch.cuda.get_rng_state(DEV)
    fn()
    torch.cuda.synchronize()
    eager = [o.clone() for o in outs]
    torch.cuda.set_rng_state(S, DEV)
    fn()
    torch.cuda.synchronize()
    eager2 = [o.clone() for o in outs]
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream(device=DEV)
    with torch.cuda.graph(g, stream=st):
        fn()
    res = []
    for r in range(n_replays):
        for o in outs:
            o.zero_()
        torch.cuda.set_rng_state(S, DEV)
        g.replay()
        torch.cuda.synchronize()
        res.append(max(rel(o, e) for o, e in zip(outs, eager)))
    print(f'{name:50s} eager-vs-eager {max(rel(a, b) for a, b in zip(eager2, eager)):.3e}   graph-vs-eager ' + ' '.join(f'{v:.3e}' for v in res), flush=True)


def main():
    AU, NW, U = mod('training.augment'), mod('training.networks'), mod('torch_utils.ops.upfirdn2d')
    LS, PE = mod('training.loss'), mod('training.phase_engine')
    AU.fixed_margin = True
    torch.manual_seed(0)
    x = torch.rand([4, 4, 64, 64], device=DEV) * 2 - 1
    which = sys.argv[1:] or ['d', 'dbench', 'dopt', 'loss', 'lossopt']

    def make_d():
        return NW.Discriminator(img_resolution=64, img_channels=4, channel_base=512, channel_max=16, num_fp16_res=2, conv_clamp=256,
                                block_kwargs=dict(fp16_dtype=torch.bfloat16), epilogue_kwargs=dict(mbstd_group_size=2)).to(DEV).train()

    def d_case(name, D, with_opt):
        params = list(D.parameters())
        od = [torch.empty_like(p) for p in params] + [torch.empty([4, 1], device=DEV)]
        opt = torch.optim.Adam(params, lr=0.002, betas=(0.0, 0.99), eps=1e-8, fused=True, capturable=True) if with_opt else None
        sync = PE.GradSync()
        w0 = [p.detach().clone() for p in params]

        def step():
            with torch.no_grad():
                for p, w in zip(params, w0):
                    p.copy_(w)                                  # same starting weights every call
            for p in params:
                p.grad = None
            D.requires_grad_(True)
            logits = D(x)
            torch.nn.functional.softplus(logits).mean().backward()
            D.requires_grad_(False)
            if opt is not None:
                flat = sync.pack(params, 'k')
                sync.unpack(params, flat, 'k')
                opt.step()
                for dst, p in zip(od, params):
                    dst.copy_(p.detach())
            else:
                for dst, p in zip(od, params):
                    dst.copy_(p.grad)
            od[-1].copy_(logits.detach())
        compare(name, step, od)

    if 'd' in which:
        d_case('D fwd+bwd', make_d(), False)
    if 'dbench' in which:
        torch.backends.cudnn.benchmark = True
        d_case('D fwd+bwd, cudnn.benchmark=True', make_d(), False)
        torch.backends.cudnn.benchmark = False
    if 'dopt' in which:
        d_case('D fwd+bwd + pack/unpack + fused capturable Adam', make_d(), True)
    if 'loss' in which or 'lossopt' in which:
        G = NW.Generator(z_dim=32, w_dim=32, img_resolution=64, img_channels=4, mapping_kwargs=dict(num_layers=2),
                         synthesis_kwargs=dict(channel_base=512, channel_max=16, num_fp16_res=2, conv_clamp=256, fp16_dtype=torch.bfloat16)).to(DEV).train().requires_grad_(False)
        D = make_d().requires_grad_(False)
        for ada in (False, True):
            pipe = None
            if ada:
                pipe = AU.AugmentPipe(**AU.augpipe_specs['bgc']).to(DEV).requires_grad_(False)
                pipe.p.copy_(torch.as_tensor(0.5))
            loss = LS.StyleGAN2Loss(device=DEV, G_mapping=G.mapping, G_synthesis=G.synthesis, D=D, augment_pipe=pipe)
            z = torch.randn([4, 32], device=DEV)
            c0 = torch.zeros([4, 0], device=DEV)
            for phase in ('Dmain', 'Gmain', 'Dreg', 'Greg'):
                m = D if phase[0] == 'D' else G
                params = list(m.parameters())
                od = [torch.zeros_like(p) for p in params]

                def step(phase=phase, m=m, params=params, od=od):
                    for p in params:
                        p.grad = None
                    m.requires_grad_(True)
                    loss.accumulate_gradients(phase=phase, real_img=x, gen_z=z, sync=True, gain=1, real_c=c0, gen_c=c0)
                    m.requires_grad_(False)
                    for dst, p in zip(od, params):
                        if p.grad is not None:
                            dst.copy_(p.grad)
                compare(f'StyleGAN2Loss {phase} ada={ada}', step, od)


def _old_main():

    AU, NW, U = mod('training.augment'), mod('training.networks'), mod('torch_utils.ops.upfirdn2d')
    AU.fixed_margin = True
    torch.manual_seed(0)
    x = torch.rand([4, 4, 64, 64], device=DEV) * 2 - 1

    # 0. plain RNG
    o = [torch.empty([1000], device=DEV)]
    compare('randn', lambda: o[0].copy_(torch.randn([1000], device=DEV)), o)

    # 1. ADA pipe pieces
    pipe = AU.AugmentPipe(**AU.augpipe_specs['bgc']).to(DEV)
    pipe.p.copy_(torch.as_tensor(0.6))
    o = [torch.empty_like(x)]
    compare('AugmentPipe bgc forward', lambda: o[0].copy_(pipe(x)), o)
    pipe_b = AU.AugmentPipe(**AU.augpipe_specs['blit']).to(DEV); pipe_b.p.copy_(torch.as_tensor(0.6))
    compare('AugmentPipe blit forward', lambda: o[0].copy_(pipe_b(x)), o)
    pipe_c = AU.AugmentPipe(**AU.augpipe_specs['color']).to(DEV); pipe_c.p.copy_(torch.as_tensor(0.6))
    compare('AugmentPipe color forward', lambda: o[0].copy_(pipe_c(x)), o)
    o3 = [torch.empty([4, 3, 3], device=DEV)]
    compare('AugmentPipe geometric params', lambda: o3[0].copy_(pipe._geometric_params(4, 64, 64, DEV, None)[0]), o3)
    xg = x.clone().requires_grad_(True)
    og = [torch.empty_like(x)]

    def ada_bwd():
        y = pipe(xg)
        g, = torch.autograd.grad([y.square().sum()], [xg])
        og[0].copy_(g)
    compare('AugmentPipe bgc fwd+bwd', ada_bwd, og)

    # 2. upfirdn2d separable
    f = U.setup_filter(AU.wavelets['sym6']).to(DEV)
    ou = [torch.empty([4, 4, 128, 128], device=DEV)]
    compare('upsample2d sym6', lambda: ou[0].copy_(U.upsample2d(x, f, up=2)), ou)

    # 3. tiny D fwd+bwd with weight grads
    D = NW.Discriminator(img_resolution=64, img_channels=4, channel_base=512, channel_max=16, num_fp16_res=2, conv_clamp=256,
                         block_kwargs=dict(fp16_dtype=torch.bfloat16), epilogue_kwargs=dict(mbstd_group_size=2)).to(DEV).train()
    params = list(D.parameters())
    od = [torch.empty_like(p) for p in params] + [torch.empty([4, 1], device=DEV)]

    def d_step():
        for p in params:
            p.grad = None
        logits = D(x)
        torch.nn.functional.softplus(logits).mean().backward()
        for dst, p in zip(od, params):
            dst.copy_(p.grad)
        od[-1].copy_(logits.detach())
    compare('Discriminator fwd+bwd (bf16 top-2)', d_step, od)

    def d_aug_step():
        for p in params:
            p.grad = None
        logits = D(pipe(x))
        torch.nn.functional.softplus(logits).mean().backward()
        for dst, p in zip(od, params):
            dst.copy_(p.grad)
        od[-1].copy_(logits.detach())
    compare('ADA + Discriminator fwd+bwd', d_aug_step, od)

    # 4. tiny G fwd+bwd
    S = NW.SynthesisNetwork(w_dim=32, img_resolution=64, img_channels=4, channel_base=512, channel_max=16, num_fp16_res=2,
                            conv_clamp=256, fp16_dtype=torch.bfloat16).to(DEV).train()
    ws = torch.randn([4, S.num_ws, 32], device=DEV)
    sp = list(S.parameters())
    osn = [torch.empty_like(p) for p in sp] + [torch.empty([4, 4, 64, 64], device=DEV)]

    def g_step():
        for p in sp:
            p.grad = None
        img = S(ws)
        img.square().mean().backward()
        for dst, p in zip(osn, sp):
            dst.copy_(p.grad if p.grad is not None else torch.zeros_like(p))
        osn[-1].copy_(img.detach())
    compare('SynthesisNetwork fwd+bwd (random noise)', g_step, osn)


if __name__ == '__main__':
    main()
