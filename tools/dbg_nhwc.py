import sys, os, importlib, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from conftest import load_pkg
load_pkg()
PKG='docker-montage-gan_amd'
co = importlib.import_module(f'{PKG}.torch_utils.custom_ops')
import oracle.oracle as orc
plugin = co.get_plugin('upfirdn2d_plugin')
rng = np.random.RandomState(33)
for shape, up, down, pad in [([2,4,37,45],1,1,[1,1,1,1]), ([1,4,8,8],1,1,[1,1,1,1]), ([1,4,8,8],2,1,[2,1,2,1]), ([1,4,8,8],1,2,[1,1,1,1])]:
    x = rng.randint(-8, 9, size=shape).astype(np.float32)
    v = np.asarray([1,3,3,1], dtype=np.float32); f = np.outer(v, v)
    want = orc.upfirdn2d(x, f, (up,up), (down,down), tuple(pad), False, 1.0)
    xg = torch.from_numpy(x).cuda().contiguous(memory_format=torch.channels_last)
    got = plugin.upfirdn2d(xg, torch.from_numpy(f).cuda(), up, up, down, down, pad[0], pad[1], pad[2], pad[3], False, 1.0).cpu().numpy()
    bad = np.argwhere(got != want)
    print(shape, up, down, pad, 'mismatches', len(bad), 'of', want.size)
    if len(bad):
        print(' n', np.unique(bad[:,0]), 'c', np.unique(bad[:,1]), 'rows', np.unique(bad[:,2])[:40], 'cols', np.unique(bad[:,3])[:50])
        b = bad[0]; print(' first', b, got[tuple(b)], want[tuple(b)])
