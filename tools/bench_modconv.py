#!/usr/bin/env python3
"""Microbenchmark: mg_modconv3x3_c64 against the three-launch path (modulation pass, library convolution, fused epilogue) at the
benchmark shape [32,64,256,256] bf16 channels-last; forward (training: conv output kept; inference: not) and forward+backward."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
importlib.import_module(PKG)
importlib.import_module(f'{PKG}.tuning').use_tuned_miopen_db()
M = importlib.import_module(f'{PKG}.torch_utils.ops.modconv3x3')
DEV = 'cuda:0'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device=DEV).manual_seed(0)
cl = torch.channels_last
x = torch.randn([N, 64, 256, 256], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
w = (torch.randn([64, 64, 3, 3], device=DEV, generator=g) / 24).to(torch.bfloat16).contiguous(memory_format=cl)
s = torch.randn([N, 64], device=DEV, generator=g) * 0.5 + 1
d = torch.rand([N, 64], device=DEV, generator=g) + 0.5
nz = torch.randn([N, 1, 256, 256], device=DEV, generator=g)
st = torch.tensor(0.3, device=DEV)
b = torch.randn([64], device=DEV, generator=g) * 0.2
dy = torch.randn([N, 64, 256, 256], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
gain = float(np.sqrt(2))


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    bb.record()
    torch.cuda.synchronize()
    return a.elapsed_time(bb) / reps * 1e3


act_bytes = N * 64 * 256 * 256 * 2
flops = 2.0 * N * 256 * 256 * 64 * 64 * 9
rows = []
for name, fused in (('fused kernel', True), ('three launches', False)):
    def fwd_nograd():
        with torch.no_grad():
            return M.fused_modconv3x3(x, w, s, d, nz, st, b, act='lrelu', gain=gain, clamp=256) if fused else \
                M._composition(x, w, s, d, nz, st, b, 'lrelu', 0.2, gain, 256.0)
    leaves = [t.clone().requires_grad_(True) for t in (x, w, s, d, st, b)]

    def fwd_grad():
        xx, ww, ss, dd, stt, b_ = leaves
        return M.fused_modconv3x3(xx, ww, ss, dd, nz, stt, b_, act='lrelu', gain=gain, clamp=256) if fused else \
            M._composition(xx, ww, ss, dd, nz, stt, b_, 'lrelu', 0.2, gain, 256.0)

    def fwd_bwd():
        y = fwd_grad()
        torch.autograd.grad([y], leaves, [dy])
    t0, t1, t2 = timeit(fwd_nograd), timeit(fwd_grad), timeit(fwd_bwd)
    rows.append((name, t0, t1, t2))
    print(f'{name:16s} fwd(no grad) {t0:8.1f} us  ({flops / t0 / 1e6:7.1f} TFLOP/s, {2 * act_bytes / t0 / 1e3:6.0f} GB/s of x+y)   '
          f'fwd(grad) {t1:8.1f} us   fwd+bwd {t2:8.1f} us', flush=True)
print(f'speed-up: fwd(no grad) {rows[1][1] / rows[0][1]:.2f}x, fwd(grad) {rows[1][2] / rows[0][2]:.2f}x, fwd+bwd {rows[1][3] / rows[0][3]:.2f}x')

# phase breakdown of workgroup 0 (debug hook)
import ctypes
native = importlib.import_module(f'{PKG}.torch_utils.native')
L = native.lib()
if hasattr(L, 'mg_modconv_debug_timing'):
    buf = torch.zeros([8], dtype=torch.int64, device=DEV)
    L.mg_modconv_debug_timing.argtypes = [ctypes.c_void_p]
    L.mg_modconv_debug_timing(buf.data_ptr())
    for label, grad in (('no grad', False), ('grad (two outputs)', True)):
        buf.zero_()
        if grad:
            xx = x.clone().requires_grad_(True)
            M.fused_modconv3x3(xx, w, s, d, nz, st, b, act='lrelu', gain=gain, clamp=256)
        else:
            with torch.no_grad():
                M.fused_modconv3x3(x, w, s, d, nz, st, b, act='lrelu', gain=gain, clamp=256)
        torch.cuda.synchronize()
        t = buf.cpu().tolist()
        tiles = (256 // 4) * (256 // 32) // max(1, (512 + N - 1) // N)          # tiles of workgroup (0, 0): one sample's tiles over its workgroups
        per = [v / max(tiles, 1) for v in t[:5]]
        print(f'{label}: cycles per tile (wave 0 of workgroup 0, {tiles} tiles): tile scalars {per[0]:.0f}, wait for the halo + barrier {per[1]:.0f}, '
              f'issue next halo {per[2]:.0f}, MFMA loop {per[3]:.0f}, epilogue + stores {per[4]:.0f}; total {sum(per):.0f}')
    L.mg_modconv_debug_timing(None)
