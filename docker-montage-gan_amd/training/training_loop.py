"""StyleGAN2-ADA training loop for one G/D pair.

``training_loop(**kwargs)`` keeps the reference's signature and keyword meaning
(training/training_loop.py:93-128) so ``train.py``'s ``subprocess_fn`` can call it unchanged.
The phase machinery is ``phase_engine`` (shared with the aio loop); DDP is replaced by one
flat-buffer all-reduce per phase over RCCL.  ``Trainer`` separates setup from ``run_iteration()`` so
that bench.py times exactly what ``training_loop`` runs.  Out of scope here, as in SURVEY 2.1:
metrics, image grids, the pickle snapshot format (snapshots are ``state_dict``s).
"""
import copy
import json
import os
import time

import numpy as np
import torch

from .. import dnnlib
from ..torch_utils import misc
from ..torch_utils import training_stats
from ..torch_utils.ops import conv2d_gradfix
from ..torch_utils.ops import grid_sample_gradfix
from . import phase_engine
from .. import tuning


def default_kwargs(num_gpus=1, batch_gpu=32, resolution=256, img_channels=4, gamma=None, num_fp16_res=4,
                   fp16_dtype=torch.float16, augment=None):
    """The kwargs ``train.py --cfg=auto`` produces at 256x256 (train.py:153-215): fmaps 0.5 -> channel_base 16384,
    2 mapping layers, lr 2.5e-3, gamma = 0.0002 * res^2 / batch, fp16 on the top 4 resolutions, conv_clamp 256."""
    E = dnnlib.EasyDict
    mb = batch_gpu * num_gpus
    a = E()
    net = dict(channel_base=16384, channel_max=512, num_fp16_res=num_fp16_res, conv_clamp=256)
    a.G_kwargs = E(class_name='training.networks.Generator', z_dim=512, w_dim=512, mapping_kwargs=E(num_layers=2),
                   synthesis_kwargs=E(fp16_dtype=fp16_dtype, **net))
    a.D_kwargs = E(class_name='training.networks.Discriminator', block_kwargs=E(fp16_dtype=fp16_dtype), mapping_kwargs=E(),
                   epilogue_kwargs=E(mbstd_group_size=min(batch_gpu, 4)), **net)
    a.G_opt_kwargs = E(class_name='torch.optim.Adam', lr=0.0025, betas=[0, 0.99], eps=1e-8)
    a.D_opt_kwargs = E(class_name='torch.optim.Adam', lr=0.0025, betas=[0, 0.99], eps=1e-8)
    a.loss_kwargs = E(class_name='training.loss.StyleGAN2Loss', r1_gamma=(0.0002 * resolution ** 2 / mb) if gamma is None else gamma)
    a.training_set_kwargs = E(class_name='training.dataset.SyntheticImageDataset', resolution=resolution, num_channels=img_channels)
    a.data_loader_kwargs = E(pin_memory=True, num_workers=0)
    a.num_gpus, a.batch_gpu, a.batch_size = num_gpus, batch_gpu, mb
    a.ema_kimg, a.ema_rampup = mb * 10 / 32, 0.05
    a.G_reg_interval, a.D_reg_interval = 4, 16
    a.augment_kwargs, a.augment_p, a.ada_target = augment, 0, None
    return a


class Trainer:
    def __init__(self, rank=0, num_gpus=1, random_seed=0, device=None, training_set_kwargs=None, data_loader_kwargs=None,
                 G_kwargs=None, D_kwargs=None, G_opt_kwargs=None, D_opt_kwargs=None, augment_kwargs=None, loss_kwargs=None,
                 batch_size=4, batch_gpu=4, ema_kimg=10, ema_rampup=None, G_reg_interval=4, D_reg_interval=16, augment_p=0,
                 ada_target=None, ada_interval=4, ada_kimg=500, cudnn_benchmark=True, allow_tf32=False, resident_batches=None,
                 graphs=False, graph_warmup=1, graph_verify=True, grad_wire_dtype=None, **_unused):
        self.rank, self.num_gpus = rank, num_gpus
        self.device = device = torch.device('cuda', rank) if device is None else torch.device(device)
        self.batch_size, self.batch_gpu = batch_size, batch_gpu
        self.ema_kimg, self.ema_rampup = ema_kimg, ema_rampup
        self.ada_target, self.ada_interval, self.ada_kimg = ada_target, ada_interval, ada_kimg
        np.random.seed(random_seed * num_gpus + rank)
        torch.manual_seed(random_seed * num_gpus + rank)
        torch.backends.cudnn.benchmark = cudnn_benchmark
        if torch.backends.cudnn.benchmark and 'MIOPEN_USER_DB_PATH' not in os.environ:
            tuning.use_tuned_miopen_db(find_mode='FAST')       # replay the shipped search results instead of a 14-minute find
        torch.backends.cuda.matmul.allow_tf32 = allow_tf32
        torch.backends.cudnn.allow_tf32 = allow_tf32
        conv2d_gradfix.enabled = True
        grid_sample_gradfix.enabled = True

        self.training_set = ts = dnnlib.util.construct_class_by_name(**training_set_kwargs)
        self._resident, self._iterator = resident_batches, None
        if resident_batches is None:
            sampler = misc.InfiniteSampler(dataset=ts, rank=rank, num_replicas=num_gpus, seed=random_seed)
            self._iterator = iter(torch.utils.data.DataLoader(dataset=ts, sampler=sampler, batch_size=batch_size // num_gpus,
                                                              **(data_loader_kwargs or {})))
        common = dict(c_dim=ts.label_dim, img_resolution=ts.resolution, img_channels=ts.num_channels)
        self.G = dnnlib.util.construct_class_by_name(**G_kwargs, **common).train().requires_grad_(False).to(device)
        self.D = dnnlib.util.construct_class_by_name(**D_kwargs, **common).train().requires_grad_(False).to(device)
        self.G_ema = copy.deepcopy(self.G).eval()
        self.augment_pipe = self.ada_stats = None
        if augment_kwargs is not None and (augment_p > 0 or ada_target is not None):
            self.augment_pipe = dnnlib.util.construct_class_by_name(**augment_kwargs).train().requires_grad_(False).to(device)
            self.augment_pipe.p.copy_(torch.as_tensor(augment_p))
            if ada_target is not None:
                self.ada_stats = training_stats.Collector(regex='Loss/signs/real')
        if num_gpus > 1:
            phase_engine.broadcast_module_state([self.G, self.D, self.G_ema])
        self.grad_sync = phase_engine.GradSync(world_size=num_gpus, wire_dtype=grad_wire_dtype)
        # graphs=True: phases are recorded into HIP graphs after `graph_warmup` eager executions (phase_engine.PhaseGraphs)
        self.graphs = phase_engine.PhaseGraphs(device, warmup=graph_warmup, verify=graph_verify) if graphs and device.type == 'cuda' else None
        self.eager_steps = False        # a loop built for phase graphs runs the next steps eagerly (bench.py's per-kernel timing): same stream, same exchange
        self._static = None
        self.loss = dnnlib.util.construct_class_by_name(device=device, G_mapping=self.G.mapping, G_synthesis=self.G.synthesis,
                                                        D=self.D, augment_pipe=self.augment_pipe, **loss_kwargs)
        self.phases = phase_engine.build_phases([('G{}', self.G, self.G.parameters(), G_opt_kwargs, G_reg_interval, 1),
                                                 ('D{}', self.D, self.D.parameters(), D_opt_kwargs, D_reg_interval, 1)], device,
                                                capturable=self.graphs is not None)
        for ph in self.phases:
            ph.start_event = ph.end_event = None
            if rank == 0 and device.type == 'cuda':
                ph.start_event, ph.end_event = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.cur_nimg, self.batch_idx = 0, 0

    def _next_reals(self):
        if self._resident is not None:
            return self._resident[self.batch_idx % len(self._resident)]
        img, c = next(self._iterator)
        return img.to(self.device, non_blocking=True), c.to(self.device, non_blocking=True)

    def _stage_inputs(self):
        """This iteration's reals, labels and latents at FIXED addresses (a recorded phase reads them where they were when
        it was recorded): drawn as upstream (:279-286), then copied into persistent tensors."""
        device, batch_gpu, ts = self.device, self.batch_gpu, self.training_set
        per_rank = self.batch_size // self.num_gpus
        img, c = self._next_reals()
        real_img = img.to(torch.float32) / 127.5 - 1
        n = len(self.phases) * per_rank
        if ts.label_dim > 0:
            labels = np.stack([ts.get_label(np.random.randint(len(ts))) for _ in range(n)])
            gen_c = torch.from_numpy(labels).pin_memory().to(device, non_blocking=True)
        else:       # unconditional: empty labels, no host work (upstream draws len(phases)*batch numpy labels per iteration)
            gen_c = None
        if self._static is None:
            st = self._static = dnnlib.EasyDict()
            st.real_img, st.real_c = real_img.clone(), c.clone()
            st.z = torch.empty([n, self.G.z_dim], device=device)
            st.gen_c = gen_c.clone() if gen_c is not None else torch.zeros([n, 0], device=device)
            st.img_rounds, st.c_rounds = st.real_img.split(batch_gpu), st.real_c.split(batch_gpu)
            st.z_rounds = [z.split(batch_gpu) for z in st.z.split(per_rank)]
            st.gen_c_rounds = [cc.split(batch_gpu) for cc in st.gen_c.split(per_rank)]
        else:
            st = self._static
            st.real_img.copy_(real_img)
            if c.numel():
                st.real_c.copy_(c)
            if gen_c is not None:
                st.gen_c.copy_(gen_c)
        st.z.normal_()
        return st

    def run_iteration(self):
        """One loop iteration; with phase graphs everything runs on the recording stream (PhaseGraphs.stream_scope)."""
        if self.graphs is None:
            return self._run_iteration()
        with self.graphs.stream_scope():
            return self._run_iteration()

    def _run_iteration(self):
        """One loop iteration (reference :277-337)."""
        device = self.device
        with torch.autograd.profiler.record_function('data_fetch'):
            st = self._stage_inputs()
        ran, rounds = 0, len(st.img_rounds)
        for phase, phase_gen_z, phase_gen_c in zip(self.phases, st.z_rounds, st.gen_c_rounds):
            if self.batch_idx % phase.interval != 0:
                continue
            if phase.start_event is not None:
                phase.start_event.record(torch.cuda.current_stream(device))

            def accumulate():
                for round_idx, (real_img, real_c, gen_z, gen_c) in enumerate(zip(st.img_rounds, st.c_rounds, phase_gen_z, phase_gen_c)):
                    self.grad_sync.begin_round(round_idx == rounds - 1)
                    self.loss.accumulate_gradients(phase=phase.name, real_img=real_img, real_c=real_c, gen_z=gen_z, gen_c=gen_c,
                                                   sync=(round_idx == rounds - 1), gain=phase.interval)
            phase_engine.run_phase(phase, accumulate, self.grad_sync, None if self.eager_steps else self.graphs, self.num_gpus,
                                   state_fn=lambda phase=phase: phase_engine.phase_state(
                                       phase, [self.G, self.D], [self.loss.pl_mean] + [t for d in training_stats._counters.values() for t in d.values()],
                                       flat=self.grad_sync._flat.get(id(phase.opt))),
                                   overlap=self.graphs is None)
            if phase.end_event is not None:
                phase.end_event.record(torch.cuda.current_stream(device))
                phase['_timed'] = True
            ran += 1
        with torch.autograd.profiler.record_function('Gema'):
            beta = phase_engine.ema_beta_for(self.batch_size, self.cur_nimg, self.ema_kimg, self.ema_rampup)
            if self.graphs is not None and not self.eager_steps and beta == getattr(self, '_ema_beta_prev', None):
                self.graphs.run(('ema', beta), lambda: phase_engine.update_ema(self.G_ema, self.G, beta))
            else:
                phase_engine.update_ema(self.G_ema, self.G, beta)
            self._ema_beta_prev = beta
        self.cur_nimg += self.batch_size
        self.batch_idx += 1
        if self.ada_stats is not None and self.batch_idx % self.ada_interval == 0:
            self.ada_stats.update()
            adjust = phase_engine.ada_adjust(self.ada_stats['Loss/signs/real'], self.ada_target, self.batch_size, self.ada_interval, self.ada_kimg)
            self.augment_pipe.p.copy_((self.augment_pipe.p + adjust).max(misc.constant(0, device=device)))
        return ran


def training_loop(run_dir='.', training_set_kwargs={}, data_loader_kwargs={}, G_kwargs={}, D_kwargs={}, G_opt_kwargs={},
                  D_opt_kwargs={}, augment_kwargs=None, loss_kwargs={}, metrics=[], random_seed=0, num_gpus=1, rank=0,
                  batch_size=4, batch_gpu=4, ema_kimg=10, ema_rampup=None, G_reg_interval=4, D_reg_interval=16, augment_p=0,
                  ada_target=None, ada_interval=4, ada_kimg=500, total_kimg=25000, kimg_per_tick=4, image_snapshot_ticks=50,
                  network_snapshot_ticks=50, resume_pkl=None, cudnn_benchmark=True, allow_tf32=False, abort_fn=None,
                  progress_fn=None, device=None):
    start_time = time.time()
    trainer = Trainer(rank=rank, num_gpus=num_gpus, random_seed=random_seed, device=device, training_set_kwargs=training_set_kwargs,
                      data_loader_kwargs=data_loader_kwargs, G_kwargs=G_kwargs, D_kwargs=D_kwargs, G_opt_kwargs=G_opt_kwargs,
                      D_opt_kwargs=D_opt_kwargs, augment_kwargs=augment_kwargs, loss_kwargs=loss_kwargs, batch_size=batch_size,
                      batch_gpu=batch_gpu, ema_kimg=ema_kimg, ema_rampup=ema_rampup, G_reg_interval=G_reg_interval,
                      D_reg_interval=D_reg_interval, augment_p=augment_p, ada_target=ada_target, ada_interval=ada_interval,
                      ada_kimg=ada_kimg, cudnn_benchmark=cudnn_benchmark, allow_tf32=allow_tf32)
    if resume_pkl is not None:
        saved = torch.load(resume_pkl, map_location=trainer.device, weights_only=True)
        for name, m in (('G', trainer.G), ('D', trainer.D), ('G_ema', trainer.G_ema)):
            if name in saved:
                m.load_state_dict(saved[name])
    if rank == 0:
        os.makedirs(run_dir, exist_ok=True)
    stats_collector = training_stats.Collector(regex='.*')
    stats_jsonl = open(os.path.join(run_dir, 'stats.jsonl'), 'wt') if rank == 0 else None
    cur_tick, tick_start_nimg, tick_start_time = 0, 0, time.time()
    if progress_fn is not None:
        progress_fn(0, total_kimg)
    while True:
        trainer.run_iteration()
        done = trainer.cur_nimg >= total_kimg * 1000
        if not done and cur_tick != 0 and trainer.cur_nimg < tick_start_nimg + kimg_per_tick * 1000:
            continue
        now = time.time()
        if rank == 0:
            print(f'tick {cur_tick:<5d} kimg {trainer.cur_nimg / 1e3:<8.1f} time {now - start_time:<9.1f} sec/kimg '
                  f'{(now - tick_start_time) / max(trainer.cur_nimg - tick_start_nimg, 1) * 1e3:<7.2f} '
                  f'augment {float(trainer.augment_pipe.p.cpu()) if trainer.augment_pipe is not None else 0:.3f}', flush=True)
        if not done and abort_fn is not None and abort_fn():
            done = True
        if rank == 0 and network_snapshot_ticks is not None and (done or cur_tick % network_snapshot_ticks == 0):
            torch.save(dict(G=trainer.G.state_dict(), D=trainer.D.state_dict(), G_ema=trainer.G_ema.state_dict()),
                       os.path.join(run_dir, f'network-snapshot-{trainer.cur_nimg // 1000:06d}.pth'))
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
