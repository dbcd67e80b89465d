#!/usr/bin/env python3
"""Per-kernel microbenchmark on the hot-path shapes: achieved algorithmic GB/s.
Usage (GPU box): python tools/microbench.py [--dtype float32] [--batch 16]"""
import argparse
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'


_native = None


def timeit(fn, iters=20, warm=3):
    """Average time of the NATIVE kernels one call launches (HIP event pairs around each launch, on the launch
    stream) -- host gaps and aten helper kernels between them are not counted.  Falls back to wall-clock events
    around the loop when the call launches no native kernel (copy_)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    _native.prof_reset(); _native.prof_enable(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    _native.prof_enable(False)
    rec = _native.prof_collect()
    _native.prof_reset()
    if rec:
        return sum(v['ms'] for v in rec.values()) / iters
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='float32')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--out', default=None)
    ap.add_argument('--layout', default='nchw', choices=['nchw', 'nhwc'])
    args = ap.parse_args()
    global _native
    _native = importlib.import_module(PKG).native
    U = importlib.import_module(f'{PKG}.torch_utils.ops.upfirdn2d')
    B = importlib.import_module(f'{PKG}.torch_utils.ops.bias_act')
    A = importlib.import_module(f'{PKG}.torch_utils.ops.alpha_composite')
    TS = importlib.import_module(f'{PKG}.torch_utils.ops.translate_sample')
    E = importlib.import_module(f'{PKG}.torch_utils.ops.modconv_epilogue')
    dev = torch.device('cuda', 0)
    dt = getattr(torch, args.dtype)
    es = torch.empty([], dtype=dt).element_size()
    n = args.batch
    f = U.setup_filter([1, 3, 3, 1]).to(dev)
    rows = []
    fmt = torch.channels_last if args.layout == 'nhwc' else torch.contiguous_format

    def rec(name, ms, nbytes):
        rows.append(dict(kernel=name, ms=round(ms, 4), GBps=round(nbytes / ms / 1e6, 1)))
        print(f'{name:48s} {ms:8.4f} ms  {nbytes / ms / 1e6:9.1f} GB/s', flush=True)

    # device-to-device copy as the practical HBM ceiling
    big = torch.empty([256 * 1024 * 1024 // es], dtype=dt, device=dev)
    dst = torch.empty_like(big)
    rec('copy_ (256 MiB)', timeit(lambda: dst.copy_(big)), 2 * big.numel() * es)
    del big, dst

    for r, c in [(256, 64), (128, 128), (64, 256), (32, 512)]:
        x = torch.randn([n, c, r, r], device=dev).to(dt).contiguous(memory_format=fmt)
        b = torch.randn([c], device=dev).to(dt)
        rec(f'bias_act lrelu fwd [{n},{c},{r},{r}]', timeit(lambda: B.bias_act(x, b, act='lrelu', clamp=256)), 2 * x.numel() * es)
        y = B.bias_act(x, b, act='lrelu', clamp=256)
        plugin = B._plugin
        rec(f'bias_act lrelu grad1 [{n},{c},{r},{r}]',
            timeit(lambda: plugin.bias_act(x, b, B._null_tensor, y, B._null_tensor, 1, 1, 3, 0.2, 1.414, 256.0)), 3 * x.numel() * es)
        d = torch.rand([n, c], device=dev) + 0.5
        nz = torch.randn([r, r], device=dev)
        with torch.no_grad():
            rec(f'modconv_epilogue fwd [{n},{c},{r},{r}]', timeit(lambda: E.modconv_epilogue(x, d, nz, b, act='lrelu', clamp=256)), 2 * x.numel() * es)
        xr = x.clone().requires_grad_(True)
        dr = d.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        st = torch.ones([], device=dev, requires_grad=True)
        ye = E.modconv_epilogue(xr, dr, nz, br, act='lrelu', clamp=256, noise_strength=st)
        gy = torch.randn_like(ye)      # preserves the layout of ye
        rec(f'modconv_epilogue bwd (dx,gd,gb,gs) [{n},{c},{r},{r}]',
            timeit(lambda: torch.autograd.grad([ye], [xr, dr, br, st], [gy], retain_graph=True)), 4 * x.numel() * es)
        ys = E.channel_scale(xr, dr)
        rec(f'channel_scale fwd [{n},{c},{r},{r}]', timeit(lambda: E.channel_scale(x, d)), 2 * x.numel() * es)
        rec(f'channel_scale bwd (dx,gd) [{n},{c},{r},{r}]',
            timeit(lambda: torch.autograd.grad([ys], [xr, dr], [gy], retain_graph=True)), 3 * x.numel() * es)
        del xr, ye, ys, gy
        xp = torch.randn([n, c, r + 1, r + 1], device=dev).to(dt).contiguous(memory_format=fmt)
        rec(f'upfirdn2d 4x4 1:1 [{n},{c},{r + 1}]->[{r}]', timeit(lambda: U.upfirdn2d(xp, f, padding=[1, 1, 1, 1], gain=4)),
            (xp.numel() + x.numel()) * es)
        rec(f'upfirdn2d 4x4 1:1 [{n},{c},{r}]->[{r + 1}]', timeit(lambda: U.upfirdn2d(x, f, padding=[2, 2, 2, 2])),
            (xp.numel() + x.numel()) * es)
        rec(f'upfirdn2d 4x4 down2 [{n},{c},{r}]->[{r // 2}]', timeit(lambda: U.upfirdn2d(x, f, down=2, padding=[1, 1, 1, 1])),
            (x.numel() + x.numel() // 4) * es)
        xh = torch.randn([n, c, r // 2, r // 2], device=dev).to(dt).contiguous(memory_format=fmt)
        rec(f'upfirdn2d 4x4 up2 [{n},{c},{r // 2}]->[{r}]', timeit(lambda: U.upfirdn2d(xh, f, up=2, padding=[2, 1, 2, 1], gain=4)),
            (x.numel() + xh.numel()) * es)
        del x, xp, xh, y
    img = torch.randn([n, 4, 128, 128], device=dev)
    rec(f'upsample2d img [{n},4,128]->[256]', timeit(lambda: U.upsample2d(img, f)), (img.numel() * 5) * 4)
    lay = torch.rand([n, 8, 4, 256, 256], device=dev)
    rec(f'alpha_composite fwd [{n},8,4,256,256]', timeit(lambda: A.alpha_composite(lay)), (lay.numel() + lay.numel() // 8) * 4)
    t = (torch.rand([n * 8, 2], device=dev) - 0.5)
    flat = lay.reshape(n * 8, 4, 256, 256)
    rec(f'translate_sample fwd [{n * 8},4,256,256]', timeit(lambda: TS.translate_sample(flat, t, 1.0)), 2 * lay.numel() * 4)
    layr = lay.clone().requires_grad_(True)
    yc = A.alpha_composite(layr)
    gy = torch.randn_like(yc)
    rec(f'alpha_composite bwd', timeit(lambda: torch.autograd.grad([yc], [layr], [gy], retain_graph=True)), (2 * lay.numel() + yc.numel()) * 4)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(rows, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
