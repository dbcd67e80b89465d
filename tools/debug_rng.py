import torch
DEV = torch.device('cuda', 0)
N = 4
shapes = []
for rep in range(2):
    shapes += [('random_', []), ('rand', []), ('randn', [N, 512])]
    shapes += [('randn', [N, 1, 8, 8])]
    for r in (16, 32, 64, 128, 256):
        shapes += [('randn', [N, 1, r, r]), ('randn', [N, 1, r, r])]
outs = [torch.empty(s if s else [], device=DEV, dtype=torch.int64 if k == 'random_' else torch.float32) for k, s in shapes]

def body():
    for (k, s), o in zip(shapes, outs):
        if k == 'random_':
            o.copy_(torch.empty([], dtype=torch.int64, device=DEV).random_(1, 12))
        elif k == 'rand':
            o.copy_(torch.rand(s, device=DEV))
        else:
            o.copy_(torch.randn(s, device=DEV))

torch.manual_seed(5)
body(); torch.cuda.synchronize()
S = torch.cuda.get_rng_state(DEV)
body(); torch.cuda.synchronize()
eager = [o.clone() for o in outs]
S_after = torch.cuda.get_rng_state(DEV)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=torch.cuda.Stream()):
    body()
for rep in range(2):
    torch.cuda.set_rng_state(S, DEV)
    g.replay(); torch.cuda.synchronize()
    bad = [(i, shapes[i]) for i, (o, e) in enumerate(zip(outs, eager)) if not torch.equal(o, e)]
    same_state = torch.equal(torch.cuda.get_rng_state(DEV), S_after)
    print('replay', rep, 'mismatching draws:', bad[:6], 'of', len(shapes), '; generator state after == eager:', same_state)
