#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own CPU code.

Runs ONLY in the authoring container (needs /root/reference); the GPU box never
sees the reference.  It imports the reference's Python (`impl='ref'` paths run
because tensors are on the CPU), feeds it seeded inputs, and stores inputs +
outputs as small .npz files plus a JSON manifest per family.  Nothing of the
reference's source text is written anywhere: fixtures are arrays and scalars.

    python tests/golden/make_golden.py            # (re)writes tests/golden/*.npz

`torchvision` is not installed here; `custom_utils.image_utils` imports four
names from it at module import time (image_utils.py:7-8) that none of the
functions exercised below use.  An empty placeholder module is registered so
that the import statement passes; it implements nothing.
"""
import io
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/montage_gan'
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    for name, attrs in (('torchvision', ()), ('torchvision.transforms', ('ToTensor', 'ToPILImage')),
                        ('torchvision.utils', ('make_grid', 'save_image'))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
    import warnings
    warnings.filterwarnings('ignore')


def save(name, arrays, manifest):
    arrays = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    buf = io.BytesIO()
    np.savez_compressed(buf, **arrays)
    with open(os.path.join(OUT, name + '.npz'), 'wb') as fh:
        fh.write(buf.getvalue())
    with open(os.path.join(OUT, name + '.json'), 'w') as fh:
        json.dump(manifest, fh, indent=1)
    print(f'{name}: {len(arrays)} arrays, {len(buf.getvalue()) / 1024:.0f} KiB, {len(manifest["cases"])} cases')


SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
        0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
        -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]


# ----------------------------------------------------------------------------
def gen_upfirdn2d():
    from torch_utils.ops import upfirdn2d as U
    g = torch.Generator().manual_seed(1234)
    filters = {
        'k1331': [1, 3, 3, 1],            # 4 taps -> 4x4 outer product (upfirdn2d.py:114-118)
        'sym6': SYM6,                     # 12 taps -> separable
        'k3': [1, 2, 1], 'k5': [1, 4, 6, 4, 1], 'k7': [1, 2, 3, 4, 3, 2, 1],
        'k2': [1, 1], 'rect': None, 'none': None,
    }
    cases, arrays = [], {}

    def add(kind, shape, fname, dtype='float32', memfmt='contiguous', **kw):
        i = len(cases)
        x = torch.randn(shape, generator=g, dtype=torch.float64).to(getattr(torch, dtype))
        if fname == 'rect':      # a non-square, non-symmetric 2-D filter
            f = torch.randn([3, 5], generator=g)
        elif fname == 'none':
            f = None
        else:
            f = U.setup_filter(filters[fname], gain=kw.pop('filter_gain', 1))
        if memfmt == 'channels_last':
            x = x.contiguous(memory_format=torch.channels_last)
        xr = x.clone().requires_grad_(True)
        fn = dict(upfirdn2d=U.upfirdn2d, upsample2d=U.upsample2d, downsample2d=U.downsample2d, filter2d=U.filter2d)[kind]
        y = fn(xr, f, impl='ref', **kw)
        dy = torch.randn(y.shape, generator=g, dtype=torch.float64).to(y.dtype)
        dx, = torch.autograd.grad([y], [xr], [dy])
        arrays[f'c{i}_x'] = x.float() if dtype == 'float16' else x
        if f is not None:
            arrays[f'c{i}_f'] = f
        arrays[f'c{i}_y'] = y.float() if dtype == 'float16' else y
        arrays[f'c{i}_dy'] = dy.float() if dtype == 'float16' else dy
        arrays[f'c{i}_dx'] = dx.float() if dtype == 'float16' else dx
        cases.append(dict(id=i, kind=kind, shape=list(shape), filter=fname, dtype=dtype, memfmt=memfmt, kwargs=kw,
                          out_shape=list(y.shape)))

    # The hot-path configurations.
    add('upfirdn2d', [2, 3, 17, 17], 'k1331', padding=[1, 1, 1, 1], gain=4)            # G conv0 post blur
    add('upfirdn2d', [2, 3, 16, 16], 'k1331', padding=[2, 2, 2, 2])                    # D conv1 pre blur
    add('upsample2d', [2, 4, 8, 8], 'k1331')                                           # ToRGB image upsample
    add('upfirdn2d', [2, 3, 16, 16], 'k1331', down=2, padding=[1, 1, 1, 1])            # D skip downsample
    add('upsample2d', [2, 3, 9, 8], 'sym6')                                            # ADA separable up
    add('downsample2d', [2, 3, 18, 16], 'sym6')                                        # ADA separable down
    # Grid: scaling per axis, paddings incl. negative, flips, gains, odd sizes, filters.
    for up in (1, 2, [2, 1], [1, 2], 3):
        for down in (1, 2, [1, 2], [2, 1]):
            for fname in ('k1331', 'k3'):
                add('upfirdn2d', [1, 2, 9, 8], fname, up=up, down=down, padding=[2, 1, 0, 3], gain=0.5)
    for pad in (0, [-1, 2], [3, -2, -1, 4], [5, 5, 5, 5]):
        for flip in (False, True):
            add('upfirdn2d', [1, 2, 16, 17], 'k5', padding=pad, flip_filter=flip)
            add('upfirdn2d', [1, 2, 16, 17], 'rect', up=2, padding=pad, flip_filter=flip, gain=4)
    for fname in ('k2', 'k7', 'sym6', 'none'):
        add('filter2d', [2, 2, 8, 9], fname)
        add('upsample2d', [2, 2, 8, 9], fname, up=2)
        add('downsample2d', [2, 2, 16, 18], fname, down=2)
    add('upfirdn2d', [2, 8, 16, 16], 'k1331', memfmt='channels_last', up=2, padding=[2, 1, 2, 1], gain=4)
    add('upfirdn2d', [2, 8, 16, 16], 'k1331', memfmt='channels_last', down=2, padding=[1, 1, 1, 1])
    add('upfirdn2d', [2, 3, 16, 16], 'k1331', dtype='float64', up=2, padding=[2, 1, 2, 1], gain=4)
    add('upsample2d', [1, 2, 8, 8], 'k1331', filter_gain=2.0)

    # Pure-integer table for the bit-exact index claim: for every case the per-axis
    # (first input sample, tap count) implied by the reference's own definition
    # (zero-insert, pad/crop, valid FIR, decimate -- upfirdn2d.py:197-220): output o
    # reads up-sampled/padded positions o*down .. o*down+fsize-1; position j maps to
    # input (j - pad0)/up when divisible and inside the image.
    save('upfirdn2d', arrays, dict(cases=cases))


# ----------------------------------------------------------------------------
def gen_bias_act():
    from torch_utils.ops import bias_act as B
    g = torch.Generator().manual_seed(4321)
    cases, arrays = [], {}
    special = torch.tensor([0.0, -0.0, 1e-4, -1e-4, 0.5, -0.5, 0.49999, 0.50001, 79.0, -79.0, 81.0, -81.0, 41.0, -41.0])

    def add(act, shape, dim, with_b, alpha=None, gain=None, clamp=None, dtype='float32', memfmt='contiguous'):
        i = len(cases)
        tdt = getattr(torch, dtype)
        x = (torch.randn(shape, generator=g, dtype=torch.float64) * 2).to(tdt)
        flat = x.reshape(-1)
        flat[:special.numel()] = special.to(tdt)          # boundary values
        if memfmt == 'channels_last':
            x = x.contiguous(memory_format=torch.channels_last)
        b = torch.randn([shape[dim]], generator=g, dtype=torch.float64).to(tdt) * 0.5 if with_b else None
        xr = x.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True) if b is not None else None
        y = B.bias_act(xr, br, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp, impl='ref')
        dy = torch.randn(y.shape, generator=g, dtype=torch.float64).to(tdt).requires_grad_(True)
        ins = [xr] + ([br] if br is not None else [])
        grads = torch.autograd.grad([y], ins, [dy], create_graph=True)
        dx = grads[0]
        arrays[f'c{i}_x'], arrays[f'c{i}_y'], arrays[f'c{i}_dy'], arrays[f'c{i}_dx'] = x, y, dy, dx
        if b is not None:
            arrays[f'c{i}_b'], arrays[f'c{i}_db'] = b, grads[1]
        # second order: d/dx and d/d(dy) of sum(dx * ddx)
        ddx = torch.randn(dx.shape, generator=g, dtype=torch.float64).to(tdt)
        arrays[f'c{i}_ddx'] = ddx
        if dx.requires_grad:
            g2 = torch.autograd.grad([dx], [xr, dy], [ddx], allow_unused=True)
            arrays[f'c{i}_g2x'] = g2[0] if g2[0] is not None else torch.zeros_like(x)
            arrays[f'c{i}_g2dy'] = g2[1] if g2[1] is not None else torch.zeros_like(x)
        cases.append(dict(id=i, act=act, shape=list(shape), dim=dim, with_b=with_b, alpha=alpha, gain=gain,
                          clamp=clamp, dtype=dtype, memfmt=memfmt))

    for act in B.activation_funcs.keys():
        add(act, [2, 5, 4, 4], 1, True)
        add(act, [2, 5, 4, 4], 1, True, clamp=0.5, gain=0.7)
        add(act, [3, 7], 1, False, alpha=0.1 if act == 'lrelu' else None)
    add('lrelu', [2, 6, 3, 5], 3, True, clamp=256)
    add('lrelu', [2, 8, 4, 4], 1, True, clamp=256, memfmt='channels_last')
    add('linear', [2, 4, 8, 8], 1, True, clamp=256)
    add('linear', [2, 4, 8, 8], 1, False, gain=float(np.sqrt(0.5)))
    add('lrelu', [2, 5, 4, 4], 1, True, dtype='float64', clamp=1.0)
    add('swish', [2, 5, 4, 4], 0, True, dtype='float64')
    save('bias_act', arrays, dict(cases=cases))


# ----------------------------------------------------------------------------
def gen_conv():
    from torch_utils.ops import conv2d_resample as CR, upfirdn2d as U, fma as FMA
    from training import networks as NW
    g = torch.Generator().manual_seed(777)
    cases, arrays = [], {}
    f = U.setup_filter([1, 3, 3, 1])

    def rnd(*s):
        return torch.randn(*s, generator=g)

    def add_cr(xs, ws, **kw):
        i = len(cases)
        x, w = rnd(xs).requires_grad_(True), rnd(ws).requires_grad_(True)
        y = CR.conv2d_resample(x, w, f=f if kw.pop('use_f', True) else None, **kw)
        dy = rnd(y.shape)
        dx, dw = torch.autograd.grad([y], [x, w], [dy])
        arrays.update({f'c{i}_x': x, f'c{i}_w': w, f'c{i}_y': y, f'c{i}_dy': dy, f'c{i}_dx': dx, f'c{i}_dw': dw})
        cases.append(dict(id=i, op='conv2d_resample', kwargs=kw, use_f=True))

    add_cr([2, 4, 8, 8], [3, 4, 1, 1], down=2)                       # 1x1 + down
    add_cr([2, 4, 8, 8], [3, 4, 1, 1], up=2)                         # 1x1 + up
    add_cr([2, 4, 8, 8], [3, 4, 3, 3], down=2, padding=1)            # blur + strided conv
    add_cr([2, 4, 8, 8], [3, 4, 3, 3], up=2, padding=1)              # transpose conv + blur
    add_cr([2, 4, 8, 8], [3, 4, 3, 3], up=2, padding=1, flip_weight=False)
    add_cr([2, 4, 8, 8], [4, 2, 3, 3], up=2, padding=1, groups=2)
    add_cr([2, 4, 8, 8], [3, 4, 3, 3], padding=1)                    # plain conv
    add_cr([2, 4, 8, 8], [3, 4, 3, 3], padding=[1, 0, 2, 1])         # generic fallback
    add_cr([2, 4, 8, 8], [3, 4, 3, 3], up=2, down=2, padding=1)      # up then down

    def add_mod(up, demod, noise, fused):
        i = len(cases)
        x, w, s = rnd([2, 4, 8, 8]).requires_grad_(True), rnd([3, 4, 3, 3]).requires_grad_(True), rnd([2, 4]).requires_grad_(True)
        res = 8 * up
        nz = rnd([res, res]) * 0.3 if noise else None
        y = NW.modulated_conv2d(x=x, weight=w, styles=s, noise=nz, up=up, padding=1, resample_filter=f,
                                demodulate=demod, flip_weight=(up == 1), fused_modconv=fused)
        dy = rnd(y.shape)
        dx, dw, ds = torch.autograd.grad([y], [x, w, s], [dy])
        arrays.update({f'c{i}_x': x, f'c{i}_w': w, f'c{i}_s': s, f'c{i}_y': y, f'c{i}_dy': dy, f'c{i}_dx': dx,
                       f'c{i}_dw': dw, f'c{i}_ds': ds})
        if nz is not None:
            arrays[f'c{i}_noise'] = nz
        cases.append(dict(id=i, op='modulated_conv2d', up=up, demodulate=demod, noise=noise, fused=fused))

    for up in (1, 2):
        for demod in (True, False):
            for noise in (True, False):
                for fused in (True, False):
                    add_mod(up, demod, noise, fused)

    i = len(cases)
    a, b, c = rnd([2, 3, 4, 4]).requires_grad_(True), rnd([2, 3, 1, 1]).requires_grad_(True), rnd([1, 1, 4, 4]).requires_grad_(True)
    y = FMA.fma(a, b, c)
    dy = rnd(y.shape)
    da, db, dc = torch.autograd.grad([y], [a, b, c], [dy])
    arrays.update({f'c{i}_a': a, f'c{i}_b': b, f'c{i}_c': c, f'c{i}_y': y, f'c{i}_dy': dy, f'c{i}_da': da,
                   f'c{i}_db': db, f'c{i}_dc': dc})
    cases.append(dict(id=i, op='fma'))
    save('conv', arrays, dict(cases=cases))


# ----------------------------------------------------------------------------
def gen_montage():
    """alpha composite (image_utils.py:112-163), STN sampling (fukuwarai/networks.py:235-258),
    layout helpers (image_utils.py:216-243, 316-335)."""
    from custom_utils import image_utils as IU
    import torch.nn.functional as F
    from PIL import Image
    g = torch.Generator().manual_seed(99)
    cases, arrays = [], {}

    # (1) the reference's own test images: 9 RGBA layers, centre crop 128x128
    ldir = '/root/reference/montage_gan/custom_utils/test_data/layers'
    names = sorted(os.listdir(ldir))
    lay = np.stack([np.asarray(Image.open(os.path.join(ldir, n)).convert('RGBA'), dtype=np.uint8) for n in names])
    lay = lay[:, 64:192, 64:192, :].transpose(0, 3, 1, 2).copy()        # [L,4,128,128] uint8
    x = torch.from_numpy(lay).float() / 255.0
    y = IU.alpha_composite_pytorch(x.clone())
    arrays['png_layers_u8'], arrays['png_out'] = lay, y
    cases.append(dict(id=0, op='alpha_composite', src='png', layers=names))

    # (2) random [0,1] stacks with exact zeros / ones in alpha, forward + backward
    def comp_case(i, b, l, h, w, lo=0.0, hi=1.0):
        x = torch.rand([b, l, 4, h, w], generator=g) * (hi - lo) + lo
        m = torch.rand([b, l, 1, h, w], generator=g)
        x[:, :, 3:4] = torch.where(m < 0.3, torch.zeros_like(m), torch.where(m > 0.85, torch.ones_like(m), x[:, :, 3:4]))
        xr = x.clone().requires_grad_(True)
        y = IU.alpha_composite_pytorch(xr)
        dy = torch.randn(y.shape, generator=g)
        dx, = torch.autograd.grad([y], [xr], [dy])
        arrays.update({f'c{i}_x': x, f'c{i}_y': y, f'c{i}_dy': dy, f'c{i}_dx': dx})
        cases.append(dict(id=i, op='alpha_composite', src='rand', shape=[b, l, 4, h, w]))

    comp_case(1, 2, 8, 12, 10)
    comp_case(2, 1, 3, 8, 8)
    comp_case(3, 2, 9, 6, 7, lo=-0.25, hi=1.25)      # out-of-range values as a generator would emit
    comp_case(4, 3, 1, 4, 4)                         # single layer = identity

    # (3) STN sampling: translate by (dx, dy), both STNv2b (shift 0) and STNv2c (+1/-1 workaround)
    def stn_case(i, b, l, c, h, w, shift, scale):
        x = torch.randn([b, l, c, h, w], generator=g)
        tr = (torch.rand([b, l, 2], generator=g) * 2 - 1) * scale
        xr, trr = x.clone().requires_grad_(True), tr.clone().requires_grad_(True)
        theta = IU.convert_translate_to_2x3(trr).view(-1, 2, 3)
        x2 = xr.view(-1, c, h, w)
        grid = F.affine_grid(theta, x2.size(), align_corners=False)
        y = F.grid_sample(x2 + shift, grid, align_corners=False) - shift
        dy = torch.randn(y.shape, generator=g)
        dx, dt = torch.autograd.grad([y], [xr, trr], [dy])
        arrays.update({f'c{i}_x': x, f'c{i}_t': tr, f'c{i}_theta': theta, f'c{i}_y': y, f'c{i}_dy': dy, f'c{i}_dx': dx,
                       f'c{i}_dt': dt})
        cases.append(dict(id=i, op='translate_sample', shape=[b, l, c, h, w], shift=shift))

    stn_case(5, 2, 3, 4, 16, 16, 0.0, 0.5)
    stn_case(6, 2, 3, 4, 16, 16, 1.0, 0.5)
    stn_case(7, 1, 2, 4, 12, 20, 1.0, 1.5)           # partly / fully out of frame
    stn_case(8, 1, 4, 2, 9, 7, 0.0, 0.05)

    # (4) layout helpers
    lst = [torch.randn([2, 4, s, s], generator=g) for s in (8, 16, 256)]
    lst_small = [torch.nn.functional.interpolate(t, size=(8, 8)) if t.shape[-1] == 256 else t for t in lst]
    pe = IU.make_batch_for_pos_estimator([lst_small[0], lst_small[1]], pad_value=-1)
    arrays['pe_in0'], arrays['pe_in1'], arrays['pe_out_corner'] = lst_small[0], lst_small[1], pe[:, :, :, 112:144, 112:144]
    arrays['pe_out_shape'] = np.asarray(pe.shape)
    arrays['pe_out_sum'] = pe.double().sum()
    cases.append(dict(id=9, op='make_batch_for_pos_estimator', pad_value=-1))
    save('montage', arrays, dict(cases=cases))


# ----------------------------------------------------------------------------
def _sd(prefix, module, arrays):
    for k, v in module.state_dict().items():
        arrays[f'{prefix}.{k}'] = v.detach().clone()


def gen_nets():
    """Small end-to-end networks with their weights: generator image (train + eval paths),
    discriminator logits, and the gradients every loss phase leaves on the parameters
    (training/networks.py, training/loss.py; custom/networks_aio.py for the aio variants)."""
    from training import networks as NW
    from training import loss as LS
    from custom import networks_aio as NA
    arrays, cases = {}, []
    torch.manual_seed(2024)
    gk = dict(z_dim=32, w_dim=32, img_resolution=32, img_channels=4, mapping_kwargs=dict(num_layers=2),
              synthesis_kwargs=dict(channel_base=256, channel_max=16, num_fp16_res=0, conv_clamp=256))
    dk = dict(img_resolution=32, img_channels=4, channel_base=256, channel_max=16, num_fp16_res=0, conv_clamp=256,
              epilogue_kwargs=dict(mbstd_group_size=2))
    G, D = NW.Generator(**gk), NW.Discriminator(**dk)
    # give the noise strengths / biases non-trivial values so that every term is exercised
    with torch.no_grad():
        for n_, p_ in list(G.named_parameters()) + list(D.named_parameters()):
            if n_.endswith('noise_strength'):
                p_.fill_(0.3)
            elif n_.endswith('bias') and 'affine' not in n_:
                p_.normal_(0, 0.2)
    _sd('G', G, arrays); _sd('D', D, arrays)
    z = torch.randn([4, 32])
    arrays['z'] = z
    G.train(); D.train()
    img_train = G(z, noise_mode='const')
    G.eval()
    img_eval = G(z, noise_mode='const')
    img_trunc = G(z, truncation_psi=0.5, noise_mode='const')
    G.train()
    arrays['img_train'], arrays['img_eval'], arrays['img_trunc'] = img_train, img_eval, img_trunc
    arrays['w_avg_after'] = G.mapping.w_avg.detach().clone()
    arrays['logits'] = D(img_train.detach())
    cases.append(dict(id=0, op='forward', G_kwargs=gk, D_kwargs=dk))

    # loss phases from a fixed seed; all on CPU, default (random) noise
    real = torch.randn([4, 4, 32, 32])
    arrays['real'] = real
    for pi, phase in enumerate(['Gmain', 'Greg', 'Dmain', 'Dreg', 'Gboth', 'Dboth']):
        G.load_state_dict({k[2:]: v for k, v in arrays.items() if k.startswith('G.')})
        D.load_state_dict({k[2:]: v for k, v in arrays.items() if k.startswith('D.')})
        G.requires_grad_(phase[0] == 'G'); D.requires_grad_(phase[0] == 'D')
        for p_ in list(G.parameters()) + list(D.parameters()):
            p_.grad = None
        loss = LS.StyleGAN2Loss(device=torch.device('cpu'), G_mapping=G.mapping, G_synthesis=G.synthesis, D=D,
                                augment_pipe=None, style_mixing_prob=0.9, r1_gamma=10, pl_batch_shrink=2, pl_decay=0.01, pl_weight=2)
        torch.manual_seed(1000 + pi)
        gen_z = torch.randn([4, 32])
        arrays[f'p{pi}_gen_z'] = gen_z
        loss.accumulate_gradients(phase=phase, real_img=real, gen_z=gen_z, sync=True, gain=2.0,
                                  real_c=torch.zeros([4, 0]), gen_c=torch.zeros([4, 0]))   # c_dim = 0: empty labels, as the base loop passes
        mod = G if phase[0] == 'G' else D
        names, sums, abss = [], [], []
        for n_, p_ in mod.named_parameters():
            if p_.grad is not None:
                names.append(n_); sums.append(float(p_.grad.double().sum())); abss.append(float(p_.grad.double().abs().sum()))
        arrays[f'p{pi}_grad_sum'] = np.asarray(sums); arrays[f'p{pi}_grad_abs'] = np.asarray(abss)
        first = [n_ for n_ in names if n_.endswith('b16.conv1.weight') or n_.endswith('b16.conv0.weight')][0]
        arrays[f'p{pi}_grad_sample'] = dict(mod.named_parameters())[first].grad
        arrays[f'p{pi}_pl_mean'] = loss.pl_mean
        cases.append(dict(id=1 + pi, op='loss_phase', phase=phase, seed=1000 + pi, gain=2.0, grad_names=names, sample_name=first))

    # aio variants: synthesis from 8x8 (conv_config_index 3), global mapping network, matching discriminator
    torch.manual_seed(77)
    sk = dict(w_dim=32, img_resolution=64, img_channels=4, init_res=[8, 8], conv_config_index=3, channel_base=256,
              channel_max=16, num_fp16_res=0, conv_clamp=256)
    S = NA.SynthesisNetwork(**sk)
    M = NA.GlobalMappingNetwork(z_dim=32, w_dim=32, num_ws=S.num_ws, num_splits=3, num_layers=2)
    Dk = dict(img_resolution=64, img_channels=4, init_res=[8, 8], conv_config_index=3, channel_base=256, channel_max=16,
              num_fp16_res=0, conv_clamp=256)
    DA = NA.Discriminator(**Dk)
    _sd('S', S, arrays); _sd('M', M, arrays); _sd('DA', DA, arrays)
    z2 = torch.randn([2, 32])
    arrays['aio_z'] = z2
    M.train(); S.train(); DA.train()
    ws = M(z2)
    arrays['aio_ws'] = ws
    imgs = [S(ws[:, li], noise_mode='const') for li in range(3)]
    arrays['aio_imgs'] = torch.stack(imgs)
    arrays['aio_logits'] = DA(imgs[1].detach())
    cases.append(dict(id=7, op='aio_forward', S_kwargs=sk, D_kwargs=Dk, num_ws=S.num_ws))
    save('nets', arrays, dict(cases=cases))


# ----------------------------------------------------------------------------
def gen_augment():
    """AugmentPipe (training/augment.py:156-482): deterministic debug_percentile outputs and seeded random
    outputs, RGBA and RGB; plus the layout glue of the aio loop (image_utils.py:198-291)."""
    from training import augment as AU
    from custom_utils import image_utils as IU
    g = torch.Generator().manual_seed(31)
    arrays, cases = {}, []
    bgc = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)
    allaug = dict(bgc, imgfilter=1, noise=1, cutout=1)
    x4 = torch.rand([3, 4, 32, 32], generator=g) * 2 - 1
    x3 = torch.rand([2, 3, 24, 40], generator=g) * 2 - 1
    arrays['x4'], arrays['x3'] = x4, x3
    i = 0
    for name, spec, x in (('bgc', bgc, 'x4'), ('all', allaug, 'x4'), ('bgc', bgc, 'x3')):
        pipe = AU.AugmentPipe(**spec)
        for pct in (0.1, 0.5, 0.9):
            arrays[f'c{i}_y'] = pipe(arrays[x].clone(), debug_percentile=pct)
            cases.append(dict(id=i, op='augment', spec=name, x=x, debug_percentile=pct, seed=None, p=1.0)); i += 1
        for pval, seed in ((0.3, 5), (1.0, 6)):
            pipe.p.copy_(torch.as_tensor(pval))
            torch.manual_seed(seed)
            xr = arrays[x].clone().requires_grad_(True)
            y = pipe(xr)
            dy = torch.randn(y.shape, generator=g)
            arrays[f'c{i}_y'], arrays[f'c{i}_dy'] = y, dy
            arrays[f'c{i}_dx'], = torch.autograd.grad([y], [xr], [dy])
            cases.append(dict(id=i, op='augment', spec=name, x=x, debug_percentile=None, seed=seed, p=pval)); i += 1

    # layout glue: content boxes / centring of real layers
    lay = torch.zeros([2, 3, 4, 256, 256])
    boxes = [(40, 60, 100, 90), (0, 0, 255, 255), (10, 200, 50, 256), (128, 128, 129, 129), None, (5, 5, 250, 7)]
    for k, bx in enumerate(boxes):
        b_, l_ = divmod(k, 3)
        if bx is None:
            continue
        x1, y1, x2, y2 = bx
        lay[b_, l_, :, y1:y2, x1:x2] = torch.rand([4, y2 - y1, x2 - x1], generator=g) * 0.9 + 0.1
    arrays['lay'] = (lay * 255).round().to(torch.uint8)
    lay = arrays['lay'].float() / 255
    pf = IU.generate_pseudo_fake(lay)
    arrays['pseudo_fake_u8'] = (pf * 255).round().to(torch.uint8)
    sizes = [[256, 256], [128, 64], [32, 256]]
    outs = IU.make_batch_for_local_d(lay, sizes, to_minus11=True)
    for k, o in enumerate(outs):
        arrays[f'local_d_{k}'] = o
    cases.append(dict(id=i, op='layout', sizes=sizes))
    save('augment', arrays, dict(cases=cases))


def stn_dither():
    """[1,1,1,128,128] deterministic ripple (a formula, not stored): without it the 8x8 constant blocks of the STN test image
    make every MaxPool window an exact tie, and which element of a tie receives the gradient is implementation-defined."""
    h = torch.arange(128, dtype=torch.float32).reshape(128, 1)
    w = torch.arange(128, dtype=torch.float32).reshape(1, 128)
    return (0.02 * torch.sin(0.37 * h + 0.71 * w + 0.5) * torch.cos(0.23 * h - 0.11 * w)).reshape(1, 1, 1, 128, 128)


def gen_aio():
    """The global half of the all-in-one step (custom/loss_aio.py:182-341) and the STN end to end
    (fukuwarai/networks.py:148-258), on toy networks whose weights are stored with the vectors:
      * theta_constrain_loss: value and gradient;
      * STNv2b / STNv2c forward: theta, re-placed layers, d/dx and d/dweights for a random upstream gradient;
      * MontageGANLoss global_Gmain / global_Dmain / global_Dreg / global_Dboth / global_Gboth / global_Greg with the analytic
        composite (renderer=None): every statistic the phase reports (count, sum, sum of squares) and the gradient
        sum / abs-sum it leaves on every parameter of every module."""
    from custom import loss_aio as LA
    from custom import networks_aio as NA
    from fukuwarai import networks as FN
    from torch_utils import training_stats as TS
    arrays, cases = {}, []
    g = torch.Generator().manual_seed(4242)

    # (1) theta constraint
    theta = torch.eye(2, 3).repeat(3, 4, 1, 1)
    theta[..., 2] = torch.randn([3, 4, 2], generator=g) * 1.2          # some translations outside [-1, 1]
    th = theta.clone().requires_grad_(True)
    val = LA.theta_constrain_loss(th, torch.device('cpu'))
    dth, = torch.autograd.grad([val], [th])
    arrays['tc_theta'], arrays['tc_value'], arrays['tc_grad'] = theta, val.detach(), dth
    cases.append(dict(id=0, op='theta_constrain_loss'))

    # (2) STN end to end at 128x128 (the smallest size its five conv/pool stages accept), tiny widths
    for ci, (cls, lo) in enumerate((('STNv2b', 0.0), ('STNv2c', -1.0))):
        torch.manual_seed(50 + ci)
        stn = getattr(FN, cls)(img_resolution=128, img_channels=4, img_layers=3, nf1=4, nf2=8)
        with torch.no_grad():
            stn.fc_loc[2].weight.normal_(0, 0.5)
            stn.fc_loc[2].bias.normal_(0, 0.2)
        _sd(f'stn{ci}', stn, arrays)
        x = torch.rand([2, 3, 4, 16, 16], generator=g).repeat_interleave(8, 3).repeat_interleave(8, 4)      # blocky: compresses well
        x = (x * (1 - lo) + lo).contiguous() + stn_dither()           # the dither breaks the exact ties a blocky image gives MaxPool
        xr = x.clone().requires_grad_(True)
        y, th = stn(xr)
        dy = torch.randn([2, 3, 4, 16, 16], generator=g).repeat_interleave(8, 3).repeat_interleave(8, 4).contiguous()
        params = list(stn.parameters())
        grads = torch.autograd.grad([y], [xr] + params, [dy])
        arrays[f'stn{ci}_x16'], arrays[f'stn{ci}_dy16'] = (x - stn_dither())[..., ::8, ::8], dy[..., ::8, ::8]
        arrays[f'stn{ci}_theta'], arrays[f'stn{ci}_y'], arrays[f'stn{ci}_dx'] = th.detach(), y.detach().to(torch.float16), grads[0].to(torch.float16)
        arrays[f'stn{ci}_y_sum'], arrays[f'stn{ci}_dx_sum'] = y.detach().double().sum(), grads[0].double().sum()
        arrays[f'stn{ci}_dparam_sum'] = np.asarray([float(t.double().sum()) for t in grads[1:]])
        arrays[f'stn{ci}_dparam_abs'] = np.asarray([float(t.double().abs().sum()) for t in grads[1:]])
        cases.append(dict(id=1 + ci, op='stn', cls=cls, names=[n for n, _ in stn.named_parameters()]))

    # (3) the global phases of MontageGANLoss
    torch.manual_seed(808)
    L, names = 2, ['hair', 'face_skin']
    sk = dict(w_dim=16, img_resolution=32, img_channels=4, init_res=[8, 8], conv_config_index=3, channel_base=128,
              channel_max=8, num_fp16_res=0, conv_clamp=256)
    dk = dict(img_channels=4, init_res=[8, 8], conv_config_index=3, channel_base=128, channel_max=8, num_fp16_res=0,
              conv_clamp=256, epilogue_kwargs=dict(mbstd_group_size=2))
    S = [NA.SynthesisNetwork(**sk) for _ in range(L)]
    M = NA.GlobalMappingNetwork(z_dim=16, w_dim=16, c_dim=0, num_ws=S[0].num_ws, num_splits=L, num_layers=2)
    DL = [NA.Discriminator(img_resolution=32, **dk) for _ in range(L)]
    GD = NA.Discriminator(img_resolution=256, **dict(dk, channel_base=1024, channel_max=4))
    STN = FN.STNv2c(img_resolution=256, img_channels=4, img_layers=L, nf1=2, nf2=4)
    with torch.no_grad():
        STN.fc_loc[2].weight.normal_(0, 0.8)            # push some translations over the +-1 box: theta loss non-zero
        STN.fc_loc[2].bias.normal_(0, 0.9)
        for m in S:
            for n_, p_ in m.named_parameters():
                if n_.endswith('noise_strength'):
                    p_.fill_(0.2)
    mods = dict(M=M, S0=S[0], S1=S[1], GD=GD, STN=STN)
    for k, m in mods.items():
        _sd(f'aio.{k}', m, arrays)
    for m in list(mods.values()) + DL:
        m.train().requires_grad_(False)
    real16 = torch.rand([2, L, 4, 16, 16], generator=g)
    real16[:, :, 3] = (real16[:, :, 3] > 0.5).float()
    arrays['aio_real16'] = real16
    real = (real16.repeat_interleave(16, 3).repeat_interleave(16, 4) * 2 - 1).contiguous()         # [B,L,4,256,256] in [-1,1]
    # the deterministic variants use alpha in [0.05, 1]: a pixel that is fully transparent in every layer makes the
    # composite's second-order gradient 0/0 = NaN upstream (then EVERY gradient of global_Dreg is NaN and is scrubbed to 0)
    real16_det = real16.clone()
    real16_det[:, :, 3] = 0.05 + 0.95 * torch.rand([2, L, 16, 16], generator=g)
    arrays['aio_real16_det'] = real16_det
    real_det = (real16_det.repeat_interleave(16, 3).repeat_interleave(16, 4) * 2 - 1).contiguous()
    sd0 = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in mods.items()}
    runs = [(phase, False) for phase in ['global_Gmain', 'global_Greg', 'global_Dmain', 'global_Dreg', 'global_Gboth', 'global_Dboth']]
    # deterministic variants (no style mixing, noise strength 0): comparable on a device whose generator differs
    runs += [(phase, True) for phase in ['global_Gmain', 'global_Dmain', 'global_Dreg']]
    for pi, (phase, det) in enumerate(runs):
        for k, m in mods.items():
            m.load_state_dict(sd0[k])
            for p_ in m.parameters():
                p_.grad = None
        if det:
            with torch.no_grad():
                for m in S:
                    for n_, p_ in m.named_parameters():
                        if n_.endswith('noise_strength'):
                            p_.zero_()
        on = [M, *S, STN] if phase[7] == 'G' else [GD]
        for m in on:
            m.requires_grad_(True)
        TS._counters.clear()
        loss = LA.MontageGANLoss(device=torch.device('cpu'), global_d_real_use_renderer=False, renderer_retrain_use_real=False,
                                 mapping_network=M, layer_names=names, local_G_list=S, augment_pipe_list=[None] * L,
                                 local_D_list=DL, pos_estimator=STN, renderer=None, global_augment_pipe=None, global_D=GD,
                                 global_r1_gamma=10, style_mixing_prob=0 if det else 0.9, r1_gamma=10)
        torch.manual_seed(3000 + pi)
        gen_z = torch.randn([2, 16])
        arrays[f'g{pi}_gen_z'] = gen_z
        loss.accumulate_gradients(phase=phase, real_blchw=real_det if det else real, real_list_of_bchw=None, gen_z=gen_z, sync=True, gain=1.5)
        for m in on:
            m.requires_grad_(False)
        gnames, sums, abss = [], [], []
        for k, m in mods.items():
            for n_, p_ in m.named_parameters():
                if p_.grad is not None:
                    gnames.append(f'{k}.{n_}'); sums.append(float(p_.grad.double().sum())); abss.append(float(p_.grad.double().abs().sum()))
        arrays[f'g{pi}_grad_sum'], arrays[f'g{pi}_grad_abs'] = np.asarray(sums), np.asarray(abss)
        snames = sorted(TS._counters)
        arrays[f'g{pi}_stats'] = np.asarray([sum(TS._counters[n_].values()).numpy() for n_ in snames]).reshape(len(snames), 3)
        cases.append(dict(id=3 + pi, op='global_phase', phase=phase, seed=3000 + pi, gain=1.5, deterministic=det, grad_names=gnames, stat_names=snames))
    cases.append(dict(id=99, op='aio_config', S_kwargs=sk, D_kwargs=dk, layer_names=names))
    save('aio', arrays, dict(cases=cases))


def gen_renderer():
    """The two renderer checkpoints the reference ships (pretrained/diff_rendering/*/renderer032000.pth.tar; loaded with
    weights_only=True): key names, shapes and float64 checksums of every tensor, and the output of the reference's own
    RendererTanh / RendererSubPixelConv (diff_rendering/networks.py:6-134) with those weights on a seeded input."""
    from diff_rendering import networks as RN
    arrays, cases = {}, []
    g = torch.Generator().manual_seed(606)
    x16 = torch.rand([1, 9, 4, 16, 16], generator=g) * 2 - 1
    arrays['x16'] = x16
    x = x16.repeat_interleave(4, 3).repeat_interleave(4, 4).contiguous()            # [1,9,4,64,64]
    for i, (sub, cls) in enumerate((('211120-1956-output-tanh', 'RendererTanh'), ('211210-1834-output-subpixel', 'RendererSubPixelConv'))):
        path = f'/root/reference/pretrained/diff_rendering/{sub}/renderer032000.pth.tar'
        saved = torch.load(path, map_location='cpu', weights_only=True)
        sd = saved['renderer']
        net = getattr(RN, cls)(img_resolution=256, img_channels=4, img_layers=9)
        net.load_state_dict(sd, strict=True)
        net.eval()
        with torch.no_grad():
            y = net(x)
        arrays[f'r{i}_y'] = y
        cases.append(dict(id=i, cls=cls, path=f'pretrained/diff_rendering/{sub}/renderer032000.pth.tar', global_step=int(saved['global_step']),
                          keys=list(sd.keys()), shapes=[list(v.shape) for v in sd.values()],
                          sums=[float(v.double().sum()) for v in sd.values()]))
    save('renderer', arrays, dict(cases=cases))


def gen_warp():
    """ADA's resampling step and colour matrix through the reference's own calls:
    F.affine_grid + torch_utils.ops.grid_sample_gradfix.grid_sample (training/augment.py:376-377) and
    `C[:, :3, :3] @ images + C[:, :3, 3:]` (training/augment.py:403-406), with first-order gradients."""
    from torch_utils.ops import grid_sample_gradfix
    g = torch.Generator().manual_seed(77)
    arrays, cases = {}, []

    def rot(deg, sx=1.0, sy=1.0, tx=0.0, ty=0.0):
        a = np.deg2rad(deg)
        return [[sx * np.cos(a), -sy * np.sin(a), tx], [sx * np.sin(a), sy * np.cos(a), ty]]
    specs = [
        ([2, 4, 20, 24], (20, 24), [rot(0), rot(17, 1.2, 0.8, 0.1, -0.2)]),
        ([3, 3, 17, 31], (26, 22), [rot(-40, 0.7, 0.7), rot(90), rot(5, 2.2, 1.9, 0.6, 0.4)]),
        ([2, 1, 9, 9], (12, 7), [rot(180, 1, 1, 1.5, 0), rot(33, 0.3, 3.0)]),
        ([1, 4, 32, 32], (44, 44), [[[1.0, 0.0, 2.0 / 32], [0.0, 1.0, -1.0 / 32]]]),       # integer-pixel shift: exact corners
    ]
    for i, (shape, (oh, ow), th) in enumerate(specs):
        x = torch.randn(shape, generator=g, requires_grad=True)
        theta = torch.tensor(np.asarray(th), dtype=torch.float32)
        grid = torch.nn.functional.affine_grid(theta=theta, size=[shape[0], shape[1], oh, ow], align_corners=False)
        y = grid_sample_gradfix.grid_sample(x, grid)
        gy = torch.randn(y.shape, generator=g)
        dx, = torch.autograd.grad([y], [x], [gy])
        arrays.update({f'w{i}_x': x.detach(), f'w{i}_theta': theta, f'w{i}_y': y.detach(), f'w{i}_gy': gy, f'w{i}_dx': dx})
        cases.append(dict(id=i, kind='warp', shape=shape, out=[oh, ow]))
    for i, ch in enumerate([4, 3]):
        n, hw = 3, 50
        x = torch.randn([n, ch, hw], generator=g, requires_grad=True)
        C = torch.eye(4).repeat(n, 1, 1) + 0.3 * torch.randn([n, 4, 4], generator=g)
        if ch == 4:
            y = torch.cat([C[:, :3, :3] @ x[:, :3, :] + C[:, :3, 3:], x[:, 3:, :]], dim=1)
        else:
            y = C[:, :3, :3] @ x + C[:, :3, 3:]
        gy = torch.randn(y.shape, generator=g)
        dx, = torch.autograd.grad([y], [x], [gy])
        arrays.update({f'm{i}_x': x.detach(), f'm{i}_C': C, f'm{i}_y': y.detach(), f'm{i}_gy': gy, f'm{i}_dx': dx})
        cases.append(dict(id=i, kind='mix', channels=ch))
    save('warp', arrays, dict(cases=cases))


if __name__ == '__main__':
    _import_reference()
    torch.manual_seed(0)
    which = sys.argv[1:] or ['upfirdn2d', 'bias_act', 'conv', 'montage', 'nets', 'augment', 'warp', 'aio', 'renderer']
    for w in which:
        globals()['gen_' + w]()
