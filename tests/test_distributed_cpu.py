"""Multi-process data-parallel path on the CPU (gloo, world_size 2): the per-phase flat-buffer gradient
all-reduce that replaces torch DDP, and a whole (tiny) training iteration per rank."""
import importlib
import os
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

PKG = 'docker-montage-gan_amd'


def _init(rank, world, init_file):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    importlib.import_module(PKG)
    torch.distributed.init_process_group(backend='gloo', init_method=f'file://{init_file}', rank=rank, world_size=world)


def _grad_sync_worker(rank, world, init_file, out_dir):
    _init(rank, world, init_file)
    PE = importlib.import_module(f'{PKG}.training.phase_engine')
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    params = list(net.parameters())
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params[1].grad[0] = float('nan') if rank == 0 else 1.0            # NaN on one rank -> scrubbed to 0 everywhere
    params[2].grad[0, 0] = float('inf')                                # +inf -> 1e5
    params[3].grad = None if rank == 1 else params[3].grad            # a parameter one rank did not touch
    sync = PE.GradSync(world_size=world)
    sync(params)
    assert not sync._agreed                                            # a disagreement is never cached: it is re-checked every time
    torch.save([p.grad.clone() for p in params], os.path.join(out_dir, f'g{rank}.pt'))
    # second round: every rank has every gradient -> the signature is verified once, then trusted (no read-back per phase)
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1))
    sync(params)
    assert len(sync._agreed) == 1
    sync(params)
    assert len(sync._agreed) == 1 and all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in params)     # mean(1.5, 1.5): the second call reduced again
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_grad_sync_mean_nan_scrub_and_missing_grads():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_grad_sync_worker, args=(2, os.path.join(d, 'init'), d), nprocs=2, join=True)
        g0, g1 = torch.load(os.path.join(d, 'g0.pt')), torch.load(os.path.join(d, 'g1.pt'))
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)                                        # identical on every rank
    assert torch.allclose(g0[0], torch.full_like(g0[0], 1.5))         # mean of 1 and 2
    assert float(g0[1][0]) == 0.0 and torch.allclose(g0[1][1:], torch.full_like(g0[1][1:], 3.0))
    assert float(g0[2][0, 0]) == 1e5
    assert torch.allclose(g0[3], torch.full_like(g0[3], 2.0))         # ranks disagreed on who has a gradient -> union, (4 + 0) / 2


def _trainer_worker(rank, world, init_file, out_dir):
    _init(rank, world, init_file)
    TL = importlib.import_module(f'{PKG}.custom.training_loop_aio')
    misc = importlib.import_module(f'{PKG}.torch_utils.misc')
    kw = TL.default_kwargs(num_gpus=world, batch_gpu=2, num_fp16_res=0)
    for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
        kw[k].channel_base, kw[k].channel_max = 1024, 8
    kw.mapping_kwargs.update(z_dim=16, w_dim=16, num_layers=1)
    kw.local_G_kwargs.w_dim = 16
    kw.training_set_kwargs.update(num_layers=2, num_samples=16)
    kw.data_loader_kwargs = dict(num_workers=0)
    torch.manual_seed(100 + rank)                                       # different initial weights per rank ...
    t = TL.AioTrainer(rank=rank, device='cpu', cfg=dict(TL.config, bypass_renderer=True), **kw)
    w0 = t.local_D_list[0].b256.conv0.weight.detach().clone()
    ran = t.run_iteration()
    for m in [t.mapping_network, *t.local_G_list, *t.local_D_list, t.pos_estimator, t.global_D]:
        misc.check_ddp_consistency(m, ignore_regex=r'.*\.w_avg')       # ... yet identical after broadcast + synced steps
    torch.save(dict(ran=ran, moved=not torch.equal(w0, t.local_D_list[0].b256.conv0.weight.detach()),
                    w=t.local_D_list[0].b256.conv0.weight.detach().clone(), z_seed=rank), os.path.join(out_dir, f't{rank}.pt'))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_aio_iteration_two_ranks_stay_consistent():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_trainer_worker, args=(2, os.path.join(d, 'init'), d), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, 't0.pt')), torch.load(os.path.join(d, 't1.pt'))
    assert r0['ran'] == r1['ran'] == 12          # 2 x (Gmain, Greg, Dmain, Dreg) + 4 global phases
    assert r0['moved'] and r1['moved']
    assert torch.equal(r0['w'], r1['w'])
