#!/usr/bin/env python3
"""HIP-event timings of the augmentation-stage kernels at the step's shapes (batch 32, RGBA, 256^2, fixed-margin size):
12-tap x2 resampling one-pass vs two passes, affine warp forward / adjoint, demod styles gradient.
usage: bench_ada_kernels.py [iters]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
importlib.import_module(PKG)
U = importlib.import_module(f'{PKG}.torch_utils.ops.upfirdn2d')
W = importlib.import_module(f'{PKG}.torch_utils.ops.affine_warp')
D = importlib.import_module(f'{PKG}.torch_utils.ops.demod')
AU = importlib.import_module(f'{PKG}.training.augment')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)


def timed(fn, nbytes, label):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    print(f'{label:58s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s ({nbytes / 1e6:.0f} MB in+out)', flush=True)
    return us


f12 = U.setup_filter(AU.wavelets['sym6']).to(dev)
n = 32
with torch.no_grad():
    for m in (32, 96):      # padded size 256 + 2 m
        x = torch.randn([n, 4, 256 + 2 * m, 256 + 2 * m], device=dev)
        for on in (True, False):
            U.sep12_enabled = on
            y = U.upsample2d(x, f12, up=2)
            timed(lambda: U.upsample2d(x, f12, up=2), 4 * (x.numel() + y.numel()), f'upsample2d sym6 {list(x.shape)} one_pass={on}')
        xd = torch.randn([n, 4, 2 * (256 + 12), 2 * (256 + 12)], device=dev)
        big = torch.randn_like(y)
        for on in (True, False):
            U.sep12_enabled = on
            yd = U.downsample2d(xd, f12, down=2, padding=-12, flip_filter=True)
            timed(lambda: U.downsample2d(xd, f12, down=2, padding=-12, flip_filter=True), 4 * (xd.numel() + yd.numel()),
                  f'downsample2d sym6 {list(xd.shape)} one_pass={on}')
            # adjoint of the upsampling: the big image down to the padded one
            pad = [12 - 6 - 1, x.shape[3] * 2 - y.shape[3] + 6 - 2 + 1, 12 - 6 - 1, x.shape[2] * 2 - y.shape[2] + 6 - 2 + 1]
            timed(lambda: U.upfirdn2d(big, f12, down=2, padding=pad, flip_filter=True, gain=4), 4 * (big.numel() + x.numel()),
                  f'adjoint of upsample2d {list(big.shape)} one_pass={on}')
        U.sep12_enabled = True
        # warp: 2x-upsampled padded image -> 2 * (256 + 12) window, random rotations
        ang = torch.rand([n], device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 6.283        # fixed draw: the cost depends on the angles
        theta = torch.stack([torch.cos(ang), -torch.sin(ang), torch.zeros_like(ang), torch.sin(ang), torch.cos(ang), torch.zeros_like(ang)], 1) * (ho_ratio := 536 / (2 * (256 + 2 * m)))
        theta = theta.reshape(n, 2, 3).contiguous()
        src = big.requires_grad_(True)
        ho = wo = 2 * (256 + 12)
        with torch.enable_grad():
            out = W.affine_warp(src, theta, ho, wo)
        timed(lambda: W.affine_warp(src.detach(), theta, ho, wo), 4 * (out.numel() * 2), f'affine_warp fwd {list(src.shape)} -> {ho} (bytes: 2 x out)')
        gy = torch.randn_like(out)
        with torch.enable_grad():
            timed(lambda: torch.autograd.grad([out], [src], [gy], retain_graph=True), 4 * (out.numel() + src.numel()), f'affine_warp adjoint -> {list(src.shape)}')
        src.requires_grad_(False)
    w = torch.randn([512, 512, 3, 3], device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    s = (torch.randn([n, 512], device=dev) + 1).requires_grad_(True)
    with torch.enable_grad():
        d = D.demod_coefs(w, s)
        g = torch.randn_like(d)
        timed(lambda: torch.autograd.grad([d], [s], [g], retain_graph=True), 4 * (512 * 512 + 3 * n * 512), 'demod styles gradient 512x512 (wsq + s, d, g)')
        timed(lambda: torch.autograd.grad([d], [w, s], [g, ], retain_graph=True), 4 * (2 * 512 * 512 * 9 + 512 * 512), 'demod weight + styles gradient 512x512x3x3')
    timed(lambda: D.demod_coefs(w.detach(), s.detach()), 4 * (512 * 512 * 10), 'demod forward 512x512x3x3')
