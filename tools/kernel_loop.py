#!/usr/bin/env python3
"""Launch ONE native kernel configuration a few times (for rocprofv3 --pmc runs).
usage: kernel_loop.py <upfirdn2d|upfirdn2d128|up2|down2|bias_act|calib4|epilogue|bwd_fused|ada_up|ada_down|demod|modconv|wgrad|torgb> <dtype> [iters] [nchw|nhwc]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
importlib.import_module(PKG)
U = importlib.import_module(f'{PKG}.torch_utils.ops.upfirdn2d')
B = importlib.import_module(f'{PKG}.torch_utils.ops.bias_act')
E = importlib.import_module(f'{PKG}.torch_utils.ops.modconv_epilogue')
which, dt = sys.argv[1], getattr(torch, sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
fmt = torch.channels_last if (len(sys.argv) > 4 and sys.argv[4] == 'nhwc') else torch.contiguous_format
dev = torch.device('cuda', 0)
n, c, r = 32, 64, 256
f = U.setup_filter([1, 3, 3, 1]).to(dev)
if which == 'upfirdn2d':
    x = torch.randn([n, c, r + 1, r + 1], device=dev).to(dt).contiguous(memory_format=fmt)
    fn = lambda: U.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4)
elif which == 'upfirdn2d128':
    c, r = 128, 128
    x = torch.randn([n, c, r + 1, r + 1], device=dev).to(dt).contiguous(memory_format=fmt)
    fn = lambda: U.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4)
elif which == 'wgrad':
    M = importlib.import_module(f'{PKG}.torch_utils.ops.modconv3x3')
    x = torch.randn([n, c, r, r], device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn_like(x)
    w = (torch.randn([64, 64, 3, 3], device=dev) / 24).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    sst = torch.randn([n, c], device=dev) + 1
    fn = lambda: M.wgrad3x3_c64(gy, x, sst, w)[0]
elif which == 'torgb':
    T = importlib.import_module(f'{PKG}.torch_utils.ops.torgb')
    x = torch.randn([n, c, r, r], device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn([4, 64, 1, 1], device=dev) / 8).to(torch.bfloat16)
    sst, b = torch.randn([n, c], device=dev) + 1, torch.randn([4], device=dev)
    fn = lambda: T.torgb(x, w, sst, b, 256)
elif which == 'bias_act':
    x = torch.randn([n, c, r, r], device=dev).to(dt).contiguous(memory_format=fmt)
    b = torch.randn([c], device=dev).to(dt)
    fn = lambda: B.bias_act(x, b, act='lrelu', clamp=256)
elif which == 'calib4':
    # known byte count through 4-byte-per-lane loads: a float32 view that is not 16-byte aligned makes
    # bias_act take its VEC=1 path (one dword per lane per access)
    base = torch.randn([n * c * r * r + 1], device=dev)
    x = base[1:].reshape(n, c, r, r)
    fn = lambda: B.bias_act(x, None, act='lrelu')
elif which == 'up2':
    x = torch.randn([n, c, r // 2, r // 2], device=dev).to(dt).contiguous(memory_format=fmt)
    fn = lambda: U.upfirdn2d(x, f, up=2, padding=[2, 1, 2, 1], gain=4)
elif which == 'down2':
    x = torch.randn([n, c, r, r], device=dev).to(dt).contiguous(memory_format=fmt)
    fn = lambda: U.upfirdn2d(x, f, down=2, padding=[1, 1, 1, 1])
elif which in ('ada_up', 'ada_down'):
    # the separable 12-tap sym6 passes of AugmentPipe's geometric stage on the padded RGBA batch (side-margin size: 256 + 2*96)
    AU = importlib.import_module(f'{PKG}.training.augment')
    f12 = U.setup_filter(AU.wavelets['sym6']).to(dev)
    if which == 'ada_up':
        x = torch.randn([n, 4, 448, 448], device=dev).to(dt)
        fn = lambda: U.upsample2d(x, f12, up=2)
    else:
        x = torch.randn([n, 4, 2 * (r + 6), 2 * (r + 6)], device=dev).to(dt)
        fn = lambda: U.downsample2d(x, f12, down=2, padding=-6, flip_filter=True)
elif which == 'demod':
    D = importlib.import_module(f'{PKG}.torch_utils.ops.demod')
    w = torch.randn([512, 512, 3, 3], device=dev).contiguous(memory_format=fmt)
    sst = torch.randn([n, 512], device=dev) + 1
    x = w
    fn = lambda: D.demod_coefs(w, sst)
elif which == 'modconv':
    M = importlib.import_module(f'{PKG}.torch_utils.ops.modconv3x3')
    x = torch.randn([n, c, r, r], device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn([64, 64, 3, 3], device=dev) / 24).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    sst, dd = torch.randn([n, c], device=dev) + 1, torch.rand([n, c], device=dev) + 0.5
    nz, st, b = torch.randn([n, 1, r, r], device=dev), torch.tensor(0.3, device=dev), torch.randn([c], device=dev)
    fn = lambda: M.fused_modconv3x3(x, w, sst, dd, nz, st, b, act='lrelu', gain=1.4142, clamp=256)
elif which == 'bwd_fused':
    x = torch.randn([n, c, r, r], device=dev).to(dt).contiguous(memory_format=fmt).requires_grad_(True)
    d = (torch.rand([n, c], device=dev) + 0.5).requires_grad_(True)
    nz = torch.randn([r, r], device=dev)
    b = torch.randn([c], device=dev).requires_grad_(True)
    st = torch.ones([], device=dev, requires_grad=True)
    ye = E.modconv_epilogue(x, d, nz, b, act='lrelu', clamp=256, noise_strength=st)
    gy = torch.randn_like(ye)
    torch.set_grad_enabled(True)
    fn = lambda: torch.autograd.grad([ye], [x, d, b, st], [gy], retain_graph=True)[0]
else:
    x = torch.randn([n, c, r, r], device=dev).to(dt).contiguous(memory_format=fmt)
    d = torch.rand([n, c], device=dev) + 0.5
    nz = torch.randn([r, r], device=dev)
    b = torch.randn([c], device=dev)
    fn = lambda: E.modconv_epilogue(x, d, nz, b, act='lrelu', clamp=256)
with (torch.enable_grad() if which == 'bwd_fused' else torch.no_grad()):
    for _ in range(iters):
        fn()
torch.cuda.synchronize()
alg = {'bwd_fused': 4 * x.numel() * x.element_size(), 'wgrad': 2 * x.numel() * 2 + 64 * 9 * 64 * 4, 'torgb': x.numel() * 2 + n * 4 * r * r * 4}.get(which)
print('algorithmic bytes per launch:', alg if alg is not None else (x.numel() + fn().numel()) * x.element_size())
