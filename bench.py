#!/usr/bin/env python3
"""Benchmark of the MontageGAN training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload aio|single] [--batch-gpu B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the training loop over one batch of synthetic, HBM-resident input: every phase
whose interval divides the iteration index (Gmain/Dmain each step, Greg every 4th, Dreg every 16th -- per local
GAN and for the global GAN), optimizer steps, EMA.  Default workload ``aio``: the full two-stage train_aio step,
8 RGBA layers of 256x256, local StyleGAN2s + global placement GAN with the analytic alpha composite
(``bypass_renderer``; the shipped renderer checkpoint is 9-layer only).  ``value`` = composed images
per second over all ranks (weak scaling: the per-GPU batch is fixed).  Rank 0 prints ONE JSON line, with
``roofline`` for the dominant hand-written kernel (HIP events on the launch stream, measured in this process)
and ``cpu_baseline`` (the same loss/network code with the aten ``impl='ref'`` ops on the host cores, bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = 'docker-montage-gan_amd'
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.3 TB/s achievable)


def mod(name):
    importlib.import_module(PKG)
    return importlib.import_module(f'{PKG}.{name}')


def build_trainer(args, device, rank, world):
    dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[args.dtype]
    nfp16 = 0 if args.dtype == 'fp32' else 4
    if args.workload == 'aio':
        TL = mod('custom.training_loop_aio')
        DS = mod('custom.dataset_aio')
        kw = TL.default_kwargs(num_gpus=world, batch_gpu=args.batch_gpu, num_fp16_res=nfp16, fp16_dtype=dtype)
        if not args.no_ada:
            AU = mod('training.augment')
            # do not wait for the whole queued step at every D call (training/augment.py docstring)
            AU.side_stream_margin, AU.fixed_margin = args.ada_margin == 'side', args.ada_margin == 'fixed'
            kw.augment_kwargs = dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc'])
            kw.augment_p, kw.ada_target = args.ada_p, None          # fixed p: timing determinism
        kw.training_set_kwargs.num_layers = args.layers
        kw.training_set_kwargs.resolution = args.res
        if args.tiny:
            for k in ('local_G_kwargs', 'local_D_kwargs', 'global_D_kwargs'):
                kw[k].channel_base, kw[k].channel_max = 512, 32
        if args.layout == 'nhwc':
            importlib.import_module(f'{PKG}.torch_utils.ops.conv2d_resample').nhwc_1x1_workaround = (args.nhwc_1x1 == 'matmul')
            kw.local_G_kwargs.fp16_channels_last = True
            kw.local_D_kwargs.block_kwargs.fp16_channels_last = True
            kw.global_D_kwargs.block_kwargs.fp16_channels_last = True
        cfg = dict(TL.config, bypass_renderer=True)
        reals = [DS.synthetic_layers(args.batch_gpu, args.layers, args.res, seed=1000 * rank + i, device=device) for i in range(2)]
        trainer = TL.AioTrainer(rank=rank, device=device, cfg=cfg, resident_batches=reals, random_seed=0, graphs=args.graphs == 'on', **kw)
        name = (f'train_aio full step: {args.layers} local StyleGAN2 (G+D) + global GAN (STN + alpha composite + D), '
                f'{args.res}x{args.res} RGBA, bypass_renderer, ' + ('no ADA' if args.no_ada else f'ADA bgc at fixed p={args.ada_p}'))
    else:
        TL = mod('training.training_loop')
        kw = TL.default_kwargs(num_gpus=world, batch_gpu=args.batch_gpu, resolution=args.res, num_fp16_res=nfp16, fp16_dtype=dtype)
        if not args.no_ada:
            AU = mod('training.augment')
            AU.side_stream_margin, AU.fixed_margin = args.ada_margin == 'side', args.ada_margin == 'fixed'
            kw.augment_kwargs = dict(class_name='training.augment.AugmentPipe', **AU.augpipe_specs['bgc'])
            kw.augment_p, kw.ada_target = args.ada_p, None
        if args.layout == 'nhwc':
            importlib.import_module(f'{PKG}.torch_utils.ops.conv2d_resample').nhwc_1x1_workaround = (args.nhwc_1x1 == 'matmul')
            kw.G_kwargs.synthesis_kwargs.fp16_channels_last = True
            kw.D_kwargs.block_kwargs.fp16_channels_last = True
        if args.tiny:
            kw.G_kwargs.synthesis_kwargs.channel_base = kw.D_kwargs.channel_base = 512
            kw.G_kwargs.synthesis_kwargs.channel_max = kw.D_kwargs.channel_max = 32
        g = torch.Generator(device='cpu').manual_seed(rank)
        reals = [(torch.randint(0, 256, [args.batch_gpu, 4, args.res, args.res], generator=g, dtype=torch.uint8).to(device),
                  torch.zeros([args.batch_gpu, 0], device=device)) for _ in range(2)]
        trainer = TL.Trainer(rank=rank, device=device, resident_batches=reals, random_seed=0, graphs=args.graphs == 'on', **kw)
        name = f'train.py step: one StyleGAN2 G+D, {args.res}x{args.res} RGBA, ' + ('no ADA' if args.no_ada else f'ADA bgc at fixed p={args.ada_p}')
    return trainer, name


def cpu_baseline(args):
    """The same network/loss code on the host with the aten reference ops (impl='ref' because the tensors are
    on the CPU): ONE local GAN's Gmain + Dmain phases at a small batch.  The full step also runs Greg/Dreg and,
    for aio, the other L-1 local GANs and the global GAN; the sample is scaled by L only, so the reported CPU
    rate is an UPPER bound."""
    NW, LS = mod('training.networks'), mod('training.loss')
    # the GPU box gives one GPU a share of 16 host cores: never oversubscribe it
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    threads = int(os.environ.get('BENCH_CPU_THREADS', min(avail, 16)))
    torch.set_num_threads(threads)
    b = args.cpu_batch
    cci = 3 if args.workload == 'aio' else 2
    init = [8, 8] if cci == 3 else [4, 4]
    base = (512, 32) if args.tiny else (16384, 512)
    S = NW.SynthesisNetwork(w_dim=512, img_resolution=args.res, img_channels=4, init_res=init, conv_config_index=cci,
                            channel_base=base[0], channel_max=base[1], conv_clamp=256).train()
    M = NW.MappingNetwork(z_dim=512, w_dim=512, num_ws=S.num_ws, num_layers=8 if cci == 3 else 2).train()
    D = NW.Discriminator(img_resolution=args.res, img_channels=4, init_res=init, conv_config_index=cci, channel_base=base[0],
                         channel_max=base[1], conv_clamp=256).train()
    loss = LS.StyleGAN2Loss(device=torch.device('cpu'), G_mapping=M, G_synthesis=S, D=D)
    real = torch.randn([b, 4, args.res, args.res])
    c0 = torch.zeros([b, 0])

    def one():
        for phase, mods in (('Gmain', [M, S]), ('Dmain', [D])):
            for m in mods:
                m.requires_grad_(True)
            loss.accumulate_gradients(phase=phase, real_img=real, gen_z=torch.randn([b, 512]), sync=True, gain=1, real_c=c0, gen_c=c0)
            for m in mods:
                m.requires_grad_(False)
                for p in m.parameters():
                    p.grad = None
    print(f'[bench] cpu_baseline: {threads} threads, batch {b} ...', file=sys.stderr, flush=True)
    t_w = time.time()
    one()                                   # warm-up
    print(f'[bench] cpu_baseline warm-up {time.time() - t_w:.1f} s', file=sys.stderr, flush=True)
    t0 = time.time()
    reps = 0
    while reps < 1 or (time.time() - t0 < args.cpu_seconds and reps < 8):
        one()
        reps += 1
    dt = (time.time() - t0) / reps
    layers = args.layers if args.workload == 'aio' else 1
    return dict(value=round(b / (dt * layers), 4), unit='img/s', cores=threads, kind='port',
                sample=(f'1 local GAN, phases Gmain+Dmain only, batch {b}, {args.res}x{args.res} fp32, {reps} repetition(s) of '
                        f'{dt:.1f} s, scaled by {layers} layer(s); Greg/Dreg/global phases not run => upper bound of the CPU rate'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=16)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='aio', choices=['aio', 'single'])
    ap.add_argument('--batch-gpu', type=int, default=None)
    ap.add_argument('--layers', type=int, default=8)
    ap.add_argument('--res', type=int, default=256)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16', 'fp32'])
    ap.add_argument('--tiny', action='store_true', help='small channel counts (smoke runs only; the line is marked)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=2)
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--prof-steps', type=int, default=4, help='extra (untimed) steps with per-kernel HIP events for the roofline object')
    ap.add_argument('--no-stats', action='store_true', help='turn training_stats.report into a no-op')
    ap.add_argument('--no-ada', action='store_true', help='run without the ADA augmentation pipe (default: bgc at fixed p = 0.3, SURVEY 8d)')
    ap.add_argument('--ada-p', type=float, default=0.3)
    ap.add_argument('--ada-margin', default='side', choices=['side', 'fixed', 'sync'],
                    help='how AugmentPipe learns its padding margin: side-stream read-back (default), clamp bound, or blocking read-back as upstream')
    ap.add_argument('--miopen', default='tuned', choices=['tuned', 'immediate', 'find'],
                    help='convolution solver selection: tuned = immediate mode over the find-db shipped in <package>/tuning/miopen '
                         '(db hit -> measured-fastest solver, miss -> heuristics); immediate = heuristics only; find = exhaustive search '
                         '(~14 min on a fresh box; writes the user db to --miopen-db-out)')
    ap.add_argument('--miopen-find', action='store_true', help='same as --miopen find')
    ap.add_argument('--miopen-db-out', default=None, help='with --miopen find: directory for MIOpen\'s user find-db / perf-db')
    ap.add_argument('--graphs', default='on', choices=['on', 'off'],
                    help='record every phase into a HIP graph after 2 eager executions and replay it (phase_engine.PhaseGraphs); '
                         'implies --ada-margin fixed (a recorded phase cannot read the padding margin back to the host)')
    ap.add_argument('--quiet', action='store_true')
    ap.add_argument('--layout', choices=['nhwc', 'nchw'], default='nhwc',
                    help='memory layout of the reduced-precision blocks (the reference\'s fp16_channels_last option): nhwc keeps MIOpen\'s '
                         'implicit-GEMM convolutions free of layout transposes; the native kernels have both')
    ap.add_argument('--nhwc-1x1', choices=['matmul', 'conv'], default='conv', help='with nhwc: few-channel 1x1 convolutions straight to the convolution library (conv) or as the reference routes them around a cuDNN pitfall (matmul)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the control flow)')
    ap.add_argument('--share-gpu', action='store_true', help='rehearsal only: every rank uses cuda:0')
    ap.add_argument('--host-profile', default=None, help='path prefix: cProfile the host over the timed steps (diagnostic; perturbs the timing)')
    ap.add_argument('--marker-file', default=None, help='write clock readings at the start/end of the timed region (for cutting a rocprofv3 trace)')
    args = ap.parse_args()
    if args.batch_gpu is None:
        args.batch_gpu = 32
    if args.graphs == 'on' and args.ada_margin != 'fixed':
        args.ada_margin = 'fixed'
    if args.miopen_find:
        args.miopen = 'find'
    # solver databases must be in place before the first convolution of the process
    miopen_db = None
    if args.miopen == 'tuned':
        miopen_db = mod('tuning').use_tuned_miopen_db()
        if miopen_db is None:
            args.miopen = 'immediate'
    elif args.miopen == 'find' and args.miopen_db_out:
        os.makedirs(args.miopen_db_out, exist_ok=True)
        os.environ['MIOPEN_USER_DB_PATH'] = os.path.abspath(args.miopen_db_out)
    if 'MIOPEN_USER_DB_PATH' not in os.environ:
        # immediate / find without the shipped results: an empty user db (the trainers would otherwise pick the tuned one up)
        import tempfile
        os.environ['MIOPEN_USER_DB_PATH'] = tempfile.mkdtemp(prefix='montage_miopen_empty_')

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1'
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            torch.distributed.init_process_group(backend='nccl', device_id=device)    # nccl == RCCL on ROCm
        else:
            torch.distributed.init_process_group(backend=args.backend)

    native = mod('torch_utils.native')
    native.lib()                                # fail loudly if the HIP library is missing
    if args.no_stats:
        mod('torch_utils.training_stats').enabled = False
    mod('torch_utils.custom_ops').verbosity = 'none'

    trainer, workload = build_trainer(args, device, rank, world)
    # MIOpen: the reference sets cudnn.benchmark=True (training_loop.py:124), i.e. find mode; an exhaustive search costs
    # ~14 minutes on a fresh box, so the default replays the search results shipped under <package>/tuning/miopen.
    torch.backends.cudnn.benchmark = args.miopen == 'find'

    def progress(msg):
        if rank == 0 and not args.quiet:
            print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)
    progress(f'built {workload}; batch/gpu {args.batch_gpu}; phases {len(trainer.phases)}')

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize(device)

    for i in range(args.warmup):
        t_w = time.perf_counter()
        trainer.run_iteration()
        torch.cuda.synchronize(device)
        progress(f'warmup {i}: {time.perf_counter() - t_w:.2f} s')
    sync_all()
    counts0 = native.launch_counts()

    def clocks():
        return {n: time.clock_gettime_ns(getattr(time, n)) for n in ('CLOCK_MONOTONIC', 'CLOCK_BOOTTIME', 'CLOCK_REALTIME', 'CLOCK_MONOTONIC_RAW')}
    host_prof = None
    if args.host_profile and rank == 0:          # cProfile over the timed steps only (slows the host: not a benchmark number)
        import cProfile
        host_prof = cProfile.Profile()
        host_prof.enable()
    mark0 = clocks()
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.run_iteration()
        if (i + 1) % 4 == 0:
            progress(f'step {i + 1}/{args.steps} enqueued at +{time.perf_counter() - t0:.2f} s')
    host_enqueue = time.perf_counter() - t0
    sync_all()
    elapsed = time.perf_counter() - t0
    if host_prof is not None:
        host_prof.disable()
        import io
        import pstats
        for key in ('cumulative', 'tottime'):
            buf = io.StringIO()
            pstats.Stats(host_prof, stream=buf).sort_stats(key).print_stats(90)
            open(f'{args.host_profile}_{key}.txt', 'w').write(buf.getvalue())
    progress(f'timed region: {elapsed:.2f} s')
    if args.marker_file and rank == 0:
        json.dump(dict(start=mark0, end=clocks(), steps=args.steps), open(args.marker_file, 'w'))
    counts1 = native.launch_counts()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel timing of the hand-written kernels (extra steps, outside the timed region)
    roofline = None
    if args.prof_steps > 0:
        # every rank runs these steps (they contain the gradient all-reduces); only rank 0 records events
        if rank == 0:
            native.prof_reset()
            native.prof_enable(True)
        for _ in range(args.prof_steps):
            trainer.run_iteration()
        torch.cuda.synchronize(device)
        rec = None
        if rank == 0:
            rec = native.prof_collect()
            native.prof_enable(False)
            native.prof_reset()
        if rec:
            total_ms = sum(v['ms'] for v in rec.values())
            name, top = max(rec.items(), key=lambda kv: kv[1]['ms'])
            achieved = top['bytes'] / (top['ms'] * 1e-3) / 1e9
            # HBM traffic: PMC counters cannot be read from inside this process; profiles/r01_pmc_traffic.json holds the
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE measurement (separate passes, calibrated) of this kernel on its
            # dominant shape as a ratio to the algorithmic bytes; applied here to this run's bytes per launch.
            traffic = None
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))['kernels'].get(name)
                if pmc:
                    traffic = round(pmc['traffic_over_algorithmic'] * top['bytes'] / top['launches'])
            except Exception:
                pass
            roofline = dict(bound='hbm', kernel=name, achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit='GB/s',
                            frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic,
                            launches_per_step=round(top['launches'] / args.prof_steps, 1),
                            avg_launch_us=round(top['ms'] * 1e3 / top['launches'], 2),
                            algorithmic_bytes_per_launch=round(top['bytes'] / top['launches']),
                            all_kernels={k: dict(ms_per_step=round(v['ms'] / args.prof_steps, 3), launches_per_step=round(v['launches'] / args.prof_steps, 1),
                                                 GBps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1)) for k, v in rec.items()},
                            native_ms_per_step=round(total_ms / args.prof_steps, 3))

    if rank == 0:
        imgs = args.batch_gpu * world * args.steps
        out = {
            'metric': 'img/sec (full G+D step), 8-layer 256^2 RGBA' if args.workload == 'aio' else 'img/sec (full G+D step), single StyleGAN2 256^2 RGBA',
            'value': round(imgs / elapsed, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 2), 'host_enqueue_ms_per_step': round(host_enqueue / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': workload + (' [TINY channel counts: smoke only]' if args.tiny else ''), 'global_batch': args.batch_gpu * world,
                       'batch_per_gpu': args.batch_gpu, 'layers': args.layers if args.workload == 'aio' else 1, 'resolution': args.res,
                       'layout': f'{args.layout} in the {args.dtype} blocks (fp16_channels_last)' if args.dtype != 'fp32' else 'nchw',
                       'conv_solver_selection': {'tuned': 'MIOpen immediate mode over the find-db shipped in tuning/miopen', 'immediate': 'MIOpen immediate-mode heuristics',
                                                 'find': 'MIOpen exhaustive find in this run'}[args.miopen],
                       'parallelism': f'dp{world} (one flat-buffer RCCL all-reduce per phase)',
                       'native_launches_per_step': {k: round((counts1[k] - counts0[k]) / args.steps, 1) for k in counts1 if counts1[k] - counts0[k]}},
            'roofline': roofline,
            'cpu_baseline': None,
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
