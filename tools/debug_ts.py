import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import load_golden
PKG = 'docker-montage-gan_amd'
importlib.import_module(PKG)
TS = importlib.import_module(f'{PKG}.torch_utils.ops.translate_sample')
cases, arr = load_golden('aio')
DEV = 'cuda:0'
for ci in (0, 1):
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).repeat_interleave(8, 3).repeat_interleave(8, 4).contiguous().to(DEV)
    x = up(arr[f'stn{ci}_x16']).reshape(6, 4, 128, 128)
    dy = up(arr[f'stn{ci}_dy16']).reshape(6, 4, 128, 128)
    t = torch.from_numpy(arr[f'stn{ci}_theta'])[..., 2].reshape(6, 2).to(DEV)
    print('t', t.cpu().numpy().round(3).tolist())
    res = {}
    for impl in ('cuda', 'ref'):
        xr, tr = x.clone().requires_grad_(True), t.clone().requires_grad_(True)
        y = TS.translate_sample(xr, tr, shift=float(ci), impl=impl)
        dx, dt = torch.autograd.grad([y], [xr, tr], [dy])
        res[impl] = (y.detach(), dx, dt)
    for k, name in enumerate(('y', 'dx', 'dt')):
        a, b = res['cuda'][k], res['ref'][k]
        d = (a - b).abs()
        print(ci, name, 'max abs diff', float(d.max()), 'ref max', float(b.abs().max()), 'mismatch frac', float((d > 2e-3).float().mean()))
        if name == 'dx' and float(d.max()) > 1e-2:
            idx = torch.nonzero(d > 1e-2)[:5]
            for i in idx.tolist():
                print('   at', i, 'native', float(a[tuple(i)]), 'ref', float(b[tuple(i)]))
            per_m = d.reshape(6, -1).max(dim=1).values
            print('   per-sample max diff', per_m.cpu().numpy().round(4).tolist())
