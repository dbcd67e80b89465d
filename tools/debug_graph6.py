#!/usr/bin/env python3
"""Diagnostic: single library ops recorded into a graph and replayed several times with the graph's private pool poisoned
(NaN) between uses: an op that relies on preparation work which is not recorded shows up as a wrong 2nd replay."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
DEV = torch.device('cuda', 0)


def mod(n):
    importlib.import_module(PKG)
    return importlib.import_module(f'{PKG}.{n}')


def rel(a, b):
    a, b = a.double(), b.double()
    d = float((a - b).norm() / (b.norm() + 1e-30))
    return d if d == d else float('inf')


def check(name, fn, n_out, replays=3, poison_mb=512):
    """fn() -> list of result tensors."""
    outs = None

    def body():
        nonlocal outs
        res = fn()
        if outs is None:
            outs = [torch.empty_like(r) for r in res]
        for o, r in zip(outs, res):
            o.copy_(r)
        del res
        junk = torch.full([poison_mb * 1024 * 1024 // 4], float('nan'), device=DEV)       # poison what was just freed
        del junk
    body(); body()
    torch.cuda.synchronize()
    eager = [o.clone() for o in outs]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream(device=DEV)):
        body()
    res = []
    for _ in range(replays):
        for o in outs:
            o.zero_()
        g.replay()
        torch.cuda.synchronize()
        res.append(max(rel(o, e) for o, e in zip(outs, eager)))
    flag = 'OK ' if max(res) < 1e-2 else 'BAD'
    print(f'{flag} {name:70s} ' + ' '.join(f'{v:.2e}' for v in res), flush=True)


def main():
    mod('tuning').use_tuned_miopen_db()
    CG = mod('torch_utils.ops.conv2d_gradfix')
    CG.enabled = True
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    g = torch.Generator(device=DEV).manual_seed(0)

    def conv_case(name, cin, cout, k, res, dtype, cl, stride=1, transpose=False, pad=None):
        fmt = torch.channels_last if cl else torch.contiguous_format
        x = torch.randn([N, cin, res, res], device=DEV, generator=g).to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
        wshape = [cin, cout, k, k] if transpose else [cout, cin, k, k]
        w = (torch.randn(wshape, device=DEV, generator=g) * 0.05).to(dtype).contiguous(memory_format=fmt).requires_grad_(True)
        p = k // 2 if pad is None else pad

        def fwd():
            if transpose:
                return [CG.conv_transpose2d(x, w, stride=stride, padding=p)]
            return [CG.conv2d(x, w, stride=stride, padding=p)]
        check(f'{name} fwd', fwd, 1)
        y = fwd()[0].detach()
        dy = torch.randn_like(y)

        def bwd():
            yy = fwd()[0]
            return list(torch.autograd.grad([yy], [x, w], [dy]))
        check(f'{name} fwd+dgrad+wgrad', bwd, 2)

        def wgrad_only():
            yy = fwd()[0]
            return list(torch.autograd.grad([yy], [w], [dy]))
        check(f'{name} fwd+wgrad', wgrad_only, 1)

    bf, f32 = torch.bfloat16, torch.float32
    conv_case('conv3x3 64->64 @256 bf16 nhwc', 64, 64, 3, 256, bf, True)
    conv_case('conv3x3 128->128 @128 bf16 nhwc', 128, 128, 3, 128, bf, True)
    conv_case('conv1x1 64->4 @256 bf16 nhwc (toRGB)', 64, 4, 1, 256, bf, True)
    conv_case('conv1x1 4->64 @256 bf16 nhwc (fromRGB)', 4, 64, 1, 256, bf, True)
    conv_case('conv1x1 64->128 @128 bf16 nhwc (skip)', 64, 128, 1, 128, bf, True)
    conv_case('convT3x3 s2 128->64 @128 bf16 nhwc', 128, 64, 3, 128, bf, True, stride=2, transpose=True, pad=0)
    conv_case('conv3x3 s2 64->128 @257 bf16 nhwc', 64, 128, 3, 257, bf, True, stride=2, pad=0)
    conv_case('conv3x3 512->512 @16 fp32 nchw', 512, 512, 3, 16, f32, False)
    conv_case('conv3x3 512->512 @8 fp32 nchw', 512, 512, 3, 8, f32, False)
    conv_case('conv1x1 512->4 @16 fp32 nchw', 512, 4, 1, 16, f32, False)
    conv_case('convT3x3 s2 512->512 @8 fp32 nchw', 512, 512, 3, 8, f32, False, stride=2, transpose=True, pad=0)
    conv_case('conv3x3 s2 512->512 @17 fp32 nchw', 512, 512, 3, 17, f32, False, stride=2, pad=0)
    conv_case('conv7x7 32->64 @256 fp32 nchw (STN)', 32, 64, 7, 256, f32, False, pad=0)

    # fully connected layers
    for m, k, n in ((N, 512, 512), (N, 512, 64), (N, 512, 4), (N, 32768, 512), (N, 512, 1), (2 * N, 512, 4096)):
        x = torch.randn([m, k], device=DEV, generator=g).requires_grad_(True)
        w = (torch.randn([n, k], device=DEV, generator=g) * 0.05).requires_grad_(True)
        b = torch.randn([n], device=DEV, generator=g).requires_grad_(True)
        dy = torch.randn([m, n], device=DEV, generator=g)

        def fc():
            y = torch.addmm(b.unsqueeze(0), x, w.t())
            return [y] + list(torch.autograd.grad([y], [x, w, b], [dy]))
        check(f'addmm [{m},{k}]x[{k},{n}] fwd+bwd', fc, 4)

        def fc2():
            y = x.matmul(w.t())
            return [y] + list(torch.autograd.grad([y], [x, w], [dy]))
        check(f'matmul [{m},{k}]x[{k},{n}] fwd+bwd', fc2, 3)

    # multi-tensor ops + fused Adam
    ps = [torch.randn([s], device=DEV, generator=g).requires_grad_(True) for s in (10, 1000, 100000, 7)]
    opt = torch.optim.Adam(ps, lr=1e-3, betas=(0.0, 0.99), fused=True, capturable=True)
    base = [p.detach().clone() for p in ps]
    gs = [torch.randn_like(p) for p in ps]

    def adam():
        with torch.no_grad():
            torch._foreach_copy_([p for p in ps], base)
        for p, gg in zip(ps, gs):
            p.grad = gg
        opt.step()
        return [p.detach() for p in ps]
    check('fused capturable Adam step', adam, 4)


if __name__ == '__main__':
    main()
