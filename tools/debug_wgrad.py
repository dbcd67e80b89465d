import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
importlib.import_module(PKG)
M = importlib.import_module(f'{PKG}.torch_utils.ops.modconv3x3')
DEV = 'cuda:0'
cl = torch.channels_last
g = torch.Generator(device=DEV).manual_seed(1)
for (n, h, w) in [(1, 4, 32), (1, 8, 32), (1, 4, 64), (2, 8, 64)]:
    x = torch.randint(-2, 3, [n, 64, h, w], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
    dy = torch.randint(-2, 3, [n, 64, h, w], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
    wt = torch.randint(-1, 2, [64, 64, 3, 3], device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=cl)
    dw, _ = M.wgrad3x3_c64(dy, x, None, None, raw=True)
    ref = torch.ops.aten.convolution_backward(dy.float().contiguous(), x.float().contiguous(), wt.float().contiguous(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    print((n, h, w), 'equal', torch.equal(dw, ref), 'max err', float((dw - ref).abs().max()), 'ref max', float(ref.abs().max()), 'dw max', float(dw.abs().max()))
    cands = {'T(co,ci)': ref.transpose(0, 1), 'flip': ref.flip([2, 3]), 'T+flip': ref.transpose(0, 1).flip([2, 3]), 'flipH': ref.flip([2]), 'flipW': ref.flip([3])}
    for k, v in cands.items():
        print('   vs', k, float((dw - v).abs().max()))
    err = (dw - ref).abs()
    print('   err per tap', [float(err[:, :, i // 3, i % 3].max()) for i in range(9)])
    print('   err per co block', [float(err[32 * a:32 * a + 32].max()) for a in range(2)], 'per ci block', [float(err[:, 32 * a:32 * a + 32].max()) for a in range(2)])
    # single-pixel probes: dy one-hot at (co0, y0, x0), x one-hot at (ci0, y1, x1)
for (co0, py, px, ci0, qy, qx) in [(0, 1, 5, 0, 1, 5), (3, 2, 7, 5, 2, 8), (40, 0, 0, 33, 1, 1), (17, 3, 31, 50, 2, 30)]:
    n, h, w = 1, 4, 32
    x = torch.zeros([n, 64, h, w], device=DEV, dtype=torch.bfloat16).contiguous(memory_format=cl)
    dy = torch.zeros_like(x)
    dy[0, co0, py, px] = 1
    x[0, ci0, qy, qx] = 1
    dw, _ = M.wgrad3x3_c64(dy, x, None, None, raw=True)
    nz = dw.nonzero().tolist()
    print('probe dy', (co0, py, px), 'x', (ci0, qy, qx), 'expected [co, ci, kh, kw] =', [co0, ci0, qy - py + 1, qx - px + 1], 'got', nz[:6], float(dw.sum()))
