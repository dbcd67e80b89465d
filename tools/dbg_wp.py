import sys, os, importlib, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from conftest import load_pkg
load_pkg()
PKG='docker-montage-gan_amd'
NW = importlib.import_module(f'{PKG}.training.networks'); WP = importlib.import_module(f'{PKG}.training.weight_prep')
DEV='cuda:0'
torch.manual_seed(2)
kw = dict(channel_base=2048, channel_max=64, conv_clamp=256, num_fp16_res=3, fp16_dtype=torch.bfloat16)
G = NW.Generator(z_dim=64, mapping_kwargs=dict(num_layers=3), synthesis_kwargs=kw, w_dim=64, img_resolution=64, img_channels=4).to(DEV).train()
D = NW.Discriminator(c_dim=0, img_resolution=64, img_channels=4, channel_base=2048, channel_max=64, conv_clamp=256, num_fp16_res=3, block_kwargs=dict(fp16_dtype=torch.bfloat16)).to(DEV).train()
z = torch.randn([4, 64], device=DEV)
out = {}
for on in (False, True):
    WP.enabled = on
    for p in list(G.parameters()) + list(D.parameters()): p.grad = None
    ws = G.mapping(z, None, skip_w_avg_update=True)
    img = G.synthesis(ws, noise_mode='const')
    torch.nn.functional.softplus(-D(img, None)).mean().backward()
    a = {('G.'+n): p.grad is not None for n, p in G.named_parameters()}
    a.update({('D.'+n): p.grad is not None for n, p in D.named_parameters()})
    for p in G.parameters(): p.grad = None
    ws = G.mapping(z, None, skip_w_avg_update=True)
    img = G.synthesis(ws, noise_mode='const')
    gw, = torch.autograd.grad([img.sum()], [ws], create_graph=True)
    gw.square().sum(2).mean(1).sqrt().mean().backward()
    b = {('G2.'+n): (p.grad is not None, float(p.grad.abs().max()) if p.grad is not None else None) for n, p in G.named_parameters()}
    out[on] = (a, b)
for k in out[False][0]:
    if out[False][0][k] != out[True][0][k]: print('first', k, out[False][0][k], out[True][0][k])
for k in out[False][1]:
    if out[False][1][k][0] != out[True][1][k][0]: print('second', k, out[False][1][k], out[True][1][k])
