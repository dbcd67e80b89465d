"""All-in-one MontageGAN training loop: L local StyleGAN2s + global placement GAN (+ renderer).

``training_loop(**kwargs)`` keeps the reference's signature and keyword meaning
(custom/training_loop_aio.py:131-187) and its module-level ``config`` / ``renderer_config`` dicts
(:37-81), so ``train_aio.py``'s ``subprocess_fn`` can call it unchanged.  Differences, all deliberate:

  * multi-GPU works: the reference wraps modules in DDP and throws the wrappers away (:409-429), and
    creates the renderer at import time on device 0 for every rank (:41,104-105) -- SURVEY D5.  Here
    every rank builds everything on ITS device, rank 0's weights are broadcast once, and each phase ends
    with one flat-buffer all-reduce over RCCL (``phase_engine.GradSync``) honouring the ``sync`` round.
  * the per-iteration host work is on the device: real batches are centred/cropped with batched gathers
    (``make_batch_for_local_d``), no ``torch.cuda.empty_cache()`` after every phase (:638, App. B-7),
    multi-tensor EMA, fused Adam, one nan_to_num per phase.
  * ``AioTrainer`` exposes setup and ``run_iteration()`` separately; ``training_loop`` is the thin
    driver around it (ticks, snapshots of ``state_dict``s) and bench.py times the same object.
Snapshots are dicts of ``state_dict``s under the reference's names (:809-833) via plain ``torch.save``.
"""
import copy
import json
import os
import time
from itertools import chain

import numpy as np
import torch

from .. import dnnlib
from ..custom_utils.image_utils import make_batch_for_local_d, normalize_minus11
from ..torch_utils import misc
from ..torch_utils import training_stats
from ..torch_utils.ops import conv2d_gradfix
from ..torch_utils.ops import grid_sample_gradfix
from ..training import phase_engine
from .. import tuning

renderer_config = {
    'img_resolution': 256,
    'img_channels': 4,
    'img_layers': 9,
    'device': 'cuda',
    'lr': 1e-3,
    'betas': (0.9, 0.999),
    'amsgrad': True,
    'loss_type': 'mse',             # l1 / mse
    'renderer_pth_path': 'pretrained/diff_rendering/211120-1956-output-tanh/renderer032000.pth.tar',
    'renderer_type': 'tanh',        # tanh / subpixel
}

config = {
    'conv_config_index': 3,
    'train_local': True,
    'train_global': True,
    'train_renderer': True,
    'train_renderer_use_real': True,
    'bypass_renderer': False,
    'global_d_real_use_renderer': True,
    'local_noaug': False,
    'global_noaug': False,
    'aug_p_max': 0.6,
    'global_g_optimize_synthesis': True,
    'use_global_mapping_network': True,
    'global_optimize_interval': 1,
    'debug': False,
}


def default_kwargs(num_gpus=1, batch_gpu=4, gamma=10, num_fp16_res=4, fp16_dtype=torch.float16, augment=None):
    """The kwargs ``train_aio.py --cfg=aio`` hands to ``training_loop`` (train_aio.py:179-235): channel_base 16384,
    channel_max 512, 8 mapping layers, fp16 on the top 4 resolutions with conv_clamp 256, Adam(2.5e-3, (0, 0.99)),
    R1 gamma 10, mbstd group 4, ema 10 kimg."""
    E = dnnlib.EasyDict
    a = E()
    a.mapping_kwargs = E(class_name='custom.networks_aio.MappingNetwork', z_dim=512, w_dim=512, num_layers=8)
    net = dict(channel_base=16384, channel_max=512, num_fp16_res=num_fp16_res, conv_clamp=256)
    a.local_G_kwargs = E(class_name='custom.networks_aio.SynthesisNetwork', w_dim=512, fp16_dtype=fp16_dtype, **net)
    dk = dict(block_kwargs=E(fp16_dtype=fp16_dtype), mapping_kwargs=E(), epilogue_kwargs=E(mbstd_group_size=4), **net)
    a.local_D_kwargs = E(class_name='custom.networks_aio.Discriminator', **copy.deepcopy(dk))
    a.global_D_kwargs = E(class_name='custom.networks_aio.Discriminator', **copy.deepcopy(dk))
    a.pos_estimator_kwargs = E(class_name='fukuwarai.networks.STNv2c')
    for k in ('local_G_opt_kwargs', 'local_D_opt_kwargs', 'global_G_opt_kwargs', 'global_D_opt_kwargs'):
        a[k] = E(class_name='torch.optim.Adam', lr=0.0025, betas=[0, 0.99], eps=1e-8)
    a.loss_kwargs = E(class_name='custom.loss_aio.MontageGANLoss', global_r1_gamma=gamma, r1_gamma=gamma)
    a.training_set_kwargs = E(class_name='custom.dataset_aio.SyntheticLayerDataset', num_layers=8, resolution=256)
    a.data_loader_kwargs = E(pin_memory=True, num_workers=0)
    a.num_gpus, a.batch_gpu, a.batch_size = num_gpus, batch_gpu, batch_gpu * num_gpus
    a.ema_kimg, a.ema_rampup = 10, None
    a.G_reg_interval, a.D_reg_interval = 4, 16
    a.augment_kwargs, a.augment_p, a.ada_target = augment, 0, None
    return a


def _set_requires_grad(modules, flag):
    for m in (modules if isinstance(modules, (list, tuple)) else [modules]):
        if m is not None:
            m.requires_grad_(flag)


class AioTrainer:
    """Everything ``training_loop`` sets up, plus ``run_iteration()`` = one pass over the phases."""

    def __init__(self, rank=0, num_gpus=1, random_seed=0, device=None, mapping_kwargs=None, local_G_kwargs=None,
                 local_D_kwargs=None, pos_estimator_kwargs=None, global_D_kwargs=None, training_set_kwargs=None,
                 data_loader_kwargs=None, batch_size=4, batch_gpu=4, ema_kimg=10, ema_rampup=None, G_reg_interval=4,
                 D_reg_interval=16, local_G_opt_kwargs=None, local_D_opt_kwargs=None, global_G_opt_kwargs=None,
                 global_D_opt_kwargs=None, loss_kwargs=None, augment_kwargs=None, augment_p=0, ada_target=None,
                 ada_interval=4, ada_kimg=500, resume_kimg=0.0, cfg=None, rcfg=None, resident_batches=None, graphs=False,
                 graph_warmup=1, graph_verify=True, grad_wire_dtype=None, **_unused):
        cfg = dict(config if cfg is None else cfg)
        rcfg = dict(renderer_config if rcfg is None else rcfg)
        if not cfg['train_global']:
            cfg['train_renderer'], cfg['bypass_renderer'] = False, True
        self.cfg, self.rcfg = cfg, rcfg
        self.rank, self.num_gpus = rank, num_gpus
        self.device = device = torch.device('cuda', rank) if device is None else torch.device(device)
        self.batch_size, self.batch_gpu = batch_size, batch_gpu
        self.ema_kimg, self.ema_rampup = ema_kimg, ema_rampup
        self.ada_target, self.ada_interval, self.ada_kimg = ada_target, ada_interval, ada_kimg
        E = dnnlib.EasyDict
        mapping_kwargs, local_G_kwargs, local_D_kwargs = E(mapping_kwargs), E(local_G_kwargs), E(local_D_kwargs)
        global_D_kwargs, training_set_kwargs, loss_kwargs = E(global_D_kwargs), E(training_set_kwargs), E(loss_kwargs)

        np.random.seed(random_seed * num_gpus + rank)
        torch.manual_seed(random_seed * num_gpus + rank)
        torch.backends.cudnn.benchmark = True           # MIOpen find mode on ROCm
        if torch.backends.cudnn.benchmark and 'MIOPEN_USER_DB_PATH' not in os.environ:
            tuning.use_tuned_miopen_db(find_mode='FAST')       # replay the shipped search results instead of a 14-minute find
        conv2d_gradfix.enabled = True
        grid_sample_gradfix.enabled = True

        # graphs=True: every phase is recorded into a HIP graph after `graph_warmup` eager executions and replayed
        # from then on (phase_engine.PhaseGraphs); needs optimizers whose step counter lives on the device.
        self.graphs = None
        self.eager_steps = False
        if graphs and device.type == 'cuda':
            self.graphs = phase_engine.PhaseGraphs(device, warmup=graph_warmup, verify=graph_verify)
        self._static = None         # persistent input tensors of the phases (refreshed in place every iteration)

        cci = cfg['conv_config_index']
        local_G_kwargs.conv_config_index = local_D_kwargs.conv_config_index = global_D_kwargs.conv_config_index = cci
        training_set_kwargs.conv_config_index = cci
        loss_kwargs.global_d_real_use_renderer = cfg['global_d_real_use_renderer']
        loss_kwargs.renderer_retrain_use_real = cfg['train_renderer_use_real']

        # data
        self.training_set = ts = dnnlib.util.construct_class_by_name(**training_set_kwargs)
        self.training_set_kwargs = dict(training_set_kwargs)
        self.layer_names = ts.layer_names
        self.layer_sizes = [ts.target_res_layer(n) for n in ts.layer_names]
        if cfg['use_global_mapping_network']:
            mapping_kwargs.class_name = 'custom.networks_aio.GlobalMappingNetwork'
            mapping_kwargs.num_splits = ts.num_layers
        self._resident = resident_batches       # optional list of device tensors [B/num_gpus, L, C, H, W] in [0,1]
        self._iterator = None
        if resident_batches is None:
            sampler = misc.InfiniteSampler(dataset=ts, rank=rank, num_replicas=num_gpus, seed=random_seed)
            self._iterator = iter(torch.utils.data.DataLoader(dataset=ts, sampler=sampler, batch_size=batch_size // num_gpus,
                                                              **(data_loader_kwargs or {})))

        # networks
        def init_module(m):
            return m.train().requires_grad_(False).to(device)

        self.local_G_list, self.local_D_list = [], []
        for name in self.layer_names:
            lk = dict(img_resolution=ts.resolution_layer(name), img_channels=ts.num_channels, init_res=ts.init_res_layer(name))
            self.local_G_list.append(init_module(dnnlib.util.construct_class_by_name(**local_G_kwargs, **lk)))
            self.local_D_list.append(init_module(dnnlib.util.construct_class_by_name(**local_D_kwargs, **lk)))
        max_ws = max(g.num_ws for g in self.local_G_list)
        self.mapping_network = init_module(dnnlib.util.construct_class_by_name(**mapping_kwargs, c_dim=0, num_ws=max_ws))
        self.pos_estimator = self.global_D = self.renderer = None
        if cfg['train_global']:
            self.pos_estimator = init_module(dnnlib.util.construct_class_by_name(
                **pos_estimator_kwargs, img_resolution=ts.resolution, img_channels=ts.num_channels, img_layers=ts.num_layers))
            self.global_D = init_module(dnnlib.util.construct_class_by_name(
                **global_D_kwargs, img_resolution=ts.resolution, img_channels=ts.num_channels, init_res=ts.init_res))
        criterion_renderer = opt_renderer = None
        if not cfg['bypass_renderer']:
            from ..diff_rendering import networks as rnets
            cls = {'tanh': rnets.RendererTanh, 'subpixel': rnets.RendererSubPixelConv}[rcfg['renderer_type']]
            self.renderer = cls(img_resolution=ts.resolution, img_channels=ts.num_channels, img_layers=ts.num_layers).to(device)
            path = rcfg.get('renderer_pth_path')
            if path and os.path.exists(path) and ts.num_layers == rcfg['img_layers']:
                saved = torch.load(path, map_location=device, weights_only=True)
                self.renderer.load_state_dict(saved['renderer'] if isinstance(saved, dict) and 'renderer' in saved else saved)
            opt_renderer = torch.optim.Adam(self.renderer.parameters(), lr=rcfg['lr'], betas=rcfg['betas'], amsgrad=rcfg['amsgrad'],
                                            capturable=self.graphs is not None)
            criterion_renderer = torch.nn.MSELoss() if rcfg['loss_type'] == 'mse' else torch.nn.L1Loss()
        self.global_G_ema = dict(mapping_network=copy.deepcopy(self.mapping_network).eval(),
                                 local_G_list=[copy.deepcopy(g).eval() for g in self.local_G_list],
                                 pos_estimator=copy.deepcopy(self.pos_estimator).eval() if self.pos_estimator is not None else None)

        # augmentation
        self.augment_pipe_list, self.ada_stats_list = [], []
        use_aug = (augment_kwargs is not None) and (augment_p > 0 or ada_target is not None)
        for name in self.layer_names:
            pipe = stats = None
            if use_aug and not cfg['local_noaug']:
                pipe = init_module(dnnlib.util.construct_class_by_name(**augment_kwargs))
                pipe.p.copy_(torch.as_tensor(augment_p))
                if ada_target is not None:
                    stats = training_stats.Collector(regex=f'{name}/Loss/signs/real')
            self.augment_pipe_list.append(pipe)
            self.ada_stats_list.append(stats)
        self.global_augment_pipe = self.global_ada_stats = None
        if cfg['train_global'] and use_aug and not cfg['global_noaug']:
            self.global_augment_pipe = init_module(dnnlib.util.construct_class_by_name(**augment_kwargs))
            self.global_augment_pipe.p.copy_(torch.as_tensor(augment_p))
            if ada_target is not None:
                self.global_ada_stats = training_stats.Collector(regex='global/Loss/signs/real')

        # start every rank from rank 0's weights (what the DDP constructor did upstream)
        if num_gpus > 1:
            phase_engine.broadcast_module_state([self.mapping_network, *self.local_G_list, *self.local_D_list,
                                                 self.pos_estimator, self.global_D, self.renderer,
                                                 self.global_G_ema['mapping_network'], *self.global_G_ema['local_G_list'],
                                                 self.global_G_ema['pos_estimator']])
        self.grad_sync = phase_engine.GradSync(world_size=num_gpus, wire_dtype=grad_wire_dtype)

        # loss
        lk = dict(device=device, layer_names=self.layer_names, mapping_network=self.mapping_network,
                  local_G_list=self.local_G_list, local_D_list=self.local_D_list, augment_pipe_list=self.augment_pipe_list)
        if cfg['train_global']:
            lk.update(pos_estimator=self.pos_estimator, global_D=self.global_D, global_augment_pipe=self.global_augment_pipe)
        if not cfg['bypass_renderer']:
            lk.update(renderer=self.renderer, criterion_renderer=criterion_renderer)
        self.loss = dnnlib.util.construct_class_by_name(**lk, **loss_kwargs)

        # phases (reference :431-509)
        self.phases = []
        if not cfg['bypass_renderer'] and cfg['train_renderer']:
            self.phases.append(phase_engine.Phase(name='Renderer', module=self.renderer, opt=opt_renderer, interval=1))
        specs = []
        if cfg['train_local']:
            for name, g, d in zip(self.layer_names, self.local_G_list, self.local_D_list):
                gm = [self.mapping_network, g]
                specs.append(('local_G{}_' + name, gm, chain(*[m.parameters() for m in gm]), local_G_opt_kwargs, G_reg_interval, 1))
                specs.append(('local_D{}_' + name, d, d.parameters(), local_D_opt_kwargs, D_reg_interval, 1))
        if cfg['train_global']:
            goi = cfg['global_optimize_interval']
            gm = [self.mapping_network, *self.local_G_list, self.pos_estimator]
            gp = gm if cfg['global_g_optimize_synthesis'] else [self.mapping_network, self.pos_estimator]
            specs.append(('global_G{}', gm, chain(*[m.parameters() for m in gp]), global_G_opt_kwargs,
                          None if G_reg_interval is None else G_reg_interval * goi, goi))
            specs.append(('global_D{}', self.global_D, self.global_D.parameters(), global_D_opt_kwargs,
                          None if D_reg_interval is None else D_reg_interval * goi, goi))
        self.phases += phase_engine.build_phases(specs, device, capturable=self.graphs is not None)
        for ph in self.phases:
            ph.start_event = ph.end_event = None
            if rank == 0 and device.type == 'cuda':
                ph.start_event, ph.end_event = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.cur_nimg = int(resume_kimg * 1000)
        self.batch_idx = 0

    # ------------------------------------------------------------------------------------------------
    def _next_reals(self):
        if self._resident is not None:
            data = self._resident[self.batch_idx % len(self._resident)]
        else:
            data = next(self._iterator).to(self.device, non_blocking=True)
        return data.detach()

    def _stage_inputs(self):
        """This iteration's reals and latents, at FIXED addresses (a recorded phase reads them where they were when it
        was recorded): computed as upstream (:599-612), then copied into persistent tensors with one multi-tensor copy."""
        device, batch_gpu = self.device, self.batch_gpu
        data = self._next_reals()                                                           # [B,L,C,H,W] in [0,1]
        blchw = list(normalize_minus11(data).split(batch_gpu))
        lists = [make_batch_for_local_d(chunk, layer_size_list=self.layer_sizes, to_minus11=True) for chunk in data.split(batch_gpu)]
        per_rank = self.batch_size // self.num_gpus
        if self._static is None:
            st = self._static = dnnlib.EasyDict()
            st.blchw = [t.clone() for t in blchw]
            st.lists = [[t.contiguous().clone() for t in ls] for ls in lists]
            st.z = torch.empty([len(self.phases) * per_rank, self.mapping_network.z_dim], device=device)
            st.z_rounds = [z.split(batch_gpu) for z in st.z.split(per_rank)]
            st.flat = st.blchw + [t for ls in st.lists for t in ls]
        else:
            st = self._static
            torch._foreach_copy_(st.flat, blchw + [t for ls in lists for t in ls])
        st.z.normal_()
        return st

    def _run_phase(self, phase, st, gen_z_rounds):
        rounds = len(st.blchw)

        def accumulate():
            for round_idx, (real_layer, gen_z, real_list) in enumerate(zip(st.blchw, gen_z_rounds, st.lists)):
                self.grad_sync.begin_round(round_idx == rounds - 1)
                self.loss.accumulate_gradients(phase=phase.name, real_blchw=real_layer, real_list_of_bchw=real_list, gen_z=gen_z,
                                               sync=(round_idx == rounds - 1), gain=phase.interval)
        # `eager_steps`: a loop built for phase graphs runs the next steps eagerly (bench.py's per-kernel timing) -- same stream, same exchange
        graphs = None if self.eager_steps else self.graphs
        phase_engine.run_phase(phase, accumulate, self.grad_sync, graphs, self.num_gpus, state_fn=lambda: self._phase_state(phase),
                               overlap=self.graphs is None)

    def _phase_state(self, phase):
        mods = [self.mapping_network, *self.local_G_list, *self.local_D_list, self.pos_estimator, self.global_D, self.renderer]
        extra = [l.pl_mean for l in self.loss.local_GAN_loss_list]
        extra += [t for per_dev in training_stats._counters.values() for t in per_dev.values()]
        if not hasattr(self, '_param_names'):
            self._param_names = {id(p): f'{type(m).__name__}{i}.{n}' for i, m in enumerate(mods) if m is not None for n, p in m.named_parameters()}
        return phase_engine.phase_state(phase, mods, extra, flat=self.grad_sync._flat.get(id(phase.opt)), names=self._param_names)

    def run_iteration(self):
        """One loop iteration; with phase graphs everything runs on the recording stream (PhaseGraphs.stream_scope)."""
        if self.graphs is None:
            return self._run_iteration()
        with self.graphs.stream_scope():
            return self._run_iteration()

    def _run_iteration(self):
        """One loop iteration (reference :597-688): data, every phase whose interval divides batch_idx,
        EMA, ADA.  Returns the number of phases executed."""
        device = self.device
        with torch.autograd.profiler.record_function('data_fetch'):
            st = self._stage_inputs()
        ran = 0
        for phase, gen_z_rounds in zip(self.phases, st.z_rounds):
            if self.batch_idx % phase.interval != 0:
                continue
            if phase.start_event is not None:
                phase.start_event.record(torch.cuda.current_stream(device))
            self._run_phase(phase, st, gen_z_rounds)
            if phase.end_event is not None:
                phase.end_event.record(torch.cuda.current_stream(device))
                phase['_timed'] = True
            ran += 1

        with torch.autograd.profiler.record_function('global_G_ema'):
            beta = phase_engine.ema_beta_for(self.batch_size, self.cur_nimg, self.ema_kimg, self.ema_rampup)

            def ema():
                phase_engine.update_ema(self.global_G_ema['mapping_network'], self.mapping_network, beta)
                phase_engine.update_ema(self.global_G_ema['local_G_list'], self.local_G_list, beta)
                if self.pos_estimator is not None:
                    phase_engine.update_ema(self.global_G_ema['pos_estimator'], self.pos_estimator, beta)
            if self.graphs is not None and not self.eager_steps and beta == getattr(self, '_ema_beta_prev', None):
                self.graphs.run(('ema', beta), ema)         # constant beta (no ramp-up, or past it): record once
            else:
                ema()
            self._ema_beta_prev = beta
        self.cur_nimg += self.batch_size
        self.batch_idx += 1

        # ADA heuristic, p clamped to [0, aug_p_max] (reference :672-682)
        for name, stats, pipe in zip(self.layer_names + ['global'], self.ada_stats_list + [self.global_ada_stats],
                                     self.augment_pipe_list + [self.global_augment_pipe]):
            if stats is not None and self.batch_idx % self.ada_interval == 0:
                stats.update()
                adjust = phase_engine.ada_adjust(stats[f'{name}/Loss/signs/real'], self.ada_target, self.batch_size, self.ada_interval, self.ada_kimg)
                pipe.p.copy_((pipe.p + adjust).max(misc.constant(0, device=device)).min(misc.constant(self.cfg['aug_p_max'], device=device)))
        return ran

    def _snapshot_modules(self):
        """(name, module or list of modules) in the reference's order and under its names (:809-826)."""
        mods = [('mapping_network', self.mapping_network), ('local_G_list', self.local_G_list), ('local_D_list', self.local_D_list)]
        if self.renderer is not None:
            mods.append(('renderer', self.renderer))
        mods += [('mapping_network_ema', self.global_G_ema['mapping_network']), ('local_G_ema', self.global_G_ema['local_G_list']),
                 ('augment_pipe_list', self.augment_pipe_list)]
        if self.pos_estimator is not None:
            mods += [('pos_estimator', self.pos_estimator), ('pos_estimator_ema', self.global_G_ema['pos_estimator']),
                     ('global_augment_pipe', self.global_augment_pipe), ('global_D', self.global_D)]
        return mods

    def state_dicts(self):
        """Snapshot: a dict of ``state_dict``s under the reference's names (:809-833), plus ``training_set_kwargs``."""
        def sd(m):
            return [sd(x) for x in m] if isinstance(m, list) else (None if m is None else copy.deepcopy(m.state_dict()))
        out = dict(training_set_kwargs=dict(self.training_set_kwargs))
        for name, module in self._snapshot_modules():
            out[name] = sd(module)
        return out

    def load_state_dicts(self, saved):
        """Resume (:345-366): every module the reference restores, missing entries skipped; a plain ``MappingNetwork``
        state is tiled into the ``GlobalMappingNetwork`` where the sizes differ (:319-331).  Returns the names restored."""
        restored = []
        resumable = ('mapping_network', 'local_G_list', 'local_D_list', 'renderer', 'mapping_network_ema', 'local_G_ema',
                     'pos_estimator', 'pos_estimator_ema', 'global_D')
        for name, module in self._snapshot_modules():
            data = saved.get(name)
            if name not in resumable or data is None or module is None:
                continue
            if isinstance(module, list):
                assert len(module) == len(data), name
                for m, d in zip(module, data):
                    m.load_state_dict(d)
            elif hasattr(module, 'num_splits'):
                cur, new = module.state_dict(), {}
                for k, v in zip(cur.keys(), data.values()):
                    if v.size() == cur[k].size():
                        new[k] = v
                    else:           # one latent stack -> one per image layer
                        rep = [1] * v.dim()
                        rep[0] = module.num_splits
                        new[k] = v.repeat(*rep)
                module.load_state_dict(new)
            else:
                module.load_state_dict(data)
            restored.append(name)
        return restored


def training_loop(rank=0, run_dir='.', random_seed=0, num_gpus=1, kimg_per_tick=4, image_snapshot_ticks=50,
                  network_snapshot_ticks=50, mapping_kwargs=None, local_G_kwargs=None, local_D_kwargs=None,
                  pos_estimator_kwargs=None, global_D_kwargs=None, training_set_kwargs=None, data_loader_kwargs=None,
                  total_kimg=25000, batch_size=4, batch_gpu=4, ema_kimg=10, ema_rampup=None, G_reg_interval=4,
                  D_reg_interval=16, local_G_opt_kwargs=None, local_D_opt_kwargs=None, global_G_opt_kwargs=None,
                  global_D_opt_kwargs=None, loss_kwargs=None, metrics=None, augment_kwargs=None, augment_p=0,
                  ada_target=None, ada_interval=4, ada_kimg=500, resume_pkl=None, resume_kimg=0.0, abort_fn=None,
                  progress_fn=None, device=None):
    assert all([mapping_kwargs, local_G_kwargs, local_D_kwargs, pos_estimator_kwargs, global_D_kwargs, training_set_kwargs,
                local_G_opt_kwargs, local_D_opt_kwargs, global_G_opt_kwargs, global_D_opt_kwargs, loss_kwargs])
    start_time = time.time()
    if rank == 0:
        os.makedirs(run_dir, exist_ok=True)
        with open(os.path.join(run_dir, 'renderer_options.json'), 'wt') as f:
            json.dump(renderer_config, f, indent=2)
        with open(os.path.join(run_dir, 'montage_gan_options.json'), 'wt') as f:
            json.dump(config, f, indent=2)
    trainer = AioTrainer(rank=rank, num_gpus=num_gpus, random_seed=random_seed, device=device, mapping_kwargs=mapping_kwargs,
                         local_G_kwargs=local_G_kwargs, local_D_kwargs=local_D_kwargs, pos_estimator_kwargs=pos_estimator_kwargs,
                         global_D_kwargs=global_D_kwargs, training_set_kwargs=training_set_kwargs,
                         data_loader_kwargs=data_loader_kwargs, batch_size=batch_size, batch_gpu=batch_gpu, ema_kimg=ema_kimg,
                         ema_rampup=ema_rampup, G_reg_interval=G_reg_interval, D_reg_interval=D_reg_interval,
                         local_G_opt_kwargs=local_G_opt_kwargs, local_D_opt_kwargs=local_D_opt_kwargs,
                         global_G_opt_kwargs=global_G_opt_kwargs, global_D_opt_kwargs=global_D_opt_kwargs, loss_kwargs=loss_kwargs,
                         augment_kwargs=augment_kwargs, augment_p=augment_p, ada_target=ada_target, ada_interval=ada_interval,
                         ada_kimg=ada_kimg, resume_kimg=resume_kimg)
    if resume_pkl is not None:
        # every rank reads the same file (upstream: rank 0 loads, the DDP constructor broadcasts), so all ranks agree
        saved = torch.load(resume_pkl, map_location=trainer.device, weights_only=True)
        trainer.load_state_dicts(saved)
    stats_collector = training_stats.Collector(regex='.*')
    stats_jsonl = open(os.path.join(run_dir, 'stats.jsonl'), 'wt') if rank == 0 else None
    cur_tick, tick_start_nimg, tick_start_time = 0, trainer.cur_nimg, time.time()
    if progress_fn is not None:
        progress_fn(0, total_kimg)
    while True:
        trainer.run_iteration()
        done = trainer.cur_nimg >= total_kimg * 1000
        if not done and cur_tick != 0 and trainer.cur_nimg < tick_start_nimg + kimg_per_tick * 1000:
            continue
        tick_end_time = time.time()
        sec_per_kimg = (tick_end_time - tick_start_time) / max(trainer.cur_nimg - tick_start_nimg, 1) * 1e3
        if rank == 0:
            print(f'tick {cur_tick:<5d} kimg {trainer.cur_nimg / 1e3:<8.1f} time {tick_end_time - start_time:<9.1f} '
                  f'sec/kimg {sec_per_kimg:<7.2f}', flush=True)
        if not done and abort_fn is not None and abort_fn():
            done = True
        if rank == 0 and network_snapshot_ticks is not None and (done or cur_tick % network_snapshot_ticks == 0):
            torch.save(trainer.state_dicts(), os.path.join(run_dir, f'network-snapshot-{trainer.cur_nimg // 1000:06d}.pth'))
        for ph in trainer.phases:
            # every rank registers every name (rank != 0 reports nothing under it): Collector._sync all-reduces one
            # [n_names, 3] tensor, so the name tables must have the same length everywhere (reference :411-416)
            value = []
            if ph.start_event is not None and ph.get('_timed') and ph.end_event.query():
                value = ph.start_event.elapsed_time(ph.end_event)
            training_stats.report0('Timing/' + ph.name, value)
        stats_collector.update()
        if stats_jsonl is not None:
            stats_jsonl.write(json.dumps(dict(stats_collector.as_dict(), timestamp=time.time())) + '\n')
            stats_jsonl.flush()
        if progress_fn is not None:
            progress_fn(trainer.cur_nimg // 1000, total_kimg)
        cur_tick += 1
        tick_start_nimg, tick_start_time = trainer.cur_nimg, time.time()
        if done:
            break
    return trainer
