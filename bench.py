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
MFMA_PEAK_TFLOPS_16 = 2500.0        # MI355X_MICROARCH.md: bf16 / fp16 MFMA, dense
MFMA_PEAK_TFLOPS_F32 = 157.3        # MI355X_MICROARCH.md: fp32 matrix (= vector) rate


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
    """The same network/loss code on the host with the aten reference ops (impl='ref' because the tensors are on the
    CPU), SURVEY 8d: ONE local GAN's four phases at a small batch, weighted by their frequencies (Gmain + Dmain every
    step, Greg every 4th, Dreg every 16th), scaled by the L layers of the aio step (the global GAN's phases are not run:
    the reported CPU rate is an UPPER bound); plus BASELINE configs[0] (64x64, 1 layer, batch 4) as whole iterations of
    ``training.training_loop.Trainer``."""
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

    def one(phase):
        mods = [M, S] if phase[0] == 'G' else [D]
        for m in mods:
            m.requires_grad_(True)
        t0 = time.time()
        loss.accumulate_gradients(phase=phase, real_img=real, gen_z=torch.randn([b, 512]), sync=True, gain=1, real_c=c0, gen_c=c0)
        dt = time.time() - t0
        for m in mods:
            m.requires_grad_(False)
            for p in m.parameters():
                p.grad = None
        return dt
    print(f'[bench] cpu_baseline: {threads} threads, batch {b} ...', file=sys.stderr, flush=True)
    t_w = time.time()
    one('Gmain')                             # warm-up
    print(f'[bench] cpu_baseline warm-up {time.time() - t_w:.1f} s', file=sys.stderr, flush=True)
    t0 = time.time()
    times = {ph: [] for ph in ('Gmain', 'Dmain', 'Greg', 'Dreg')}
    while time.time() - t0 < args.cpu_seconds and len(times['Dreg']) < 4:
        for ph in times:
            times[ph].append(one(ph))
    avg = {ph: sum(v) / len(v) for ph, v in times.items()}
    dt = avg['Gmain'] + avg['Dmain'] + avg['Greg'] / 4 + avg['Dreg'] / 16          # one step of one local GAN
    layers = args.layers if args.workload == 'aio' else 1
    out = dict(value=round(b / (dt * layers), 4), unit='img/s', cores=threads, kind='port',
               sample=(f'1 local GAN, all four phases at their frequencies (Gmain {avg["Gmain"]:.2f} s + Dmain {avg["Dmain"]:.2f} s + Greg {avg["Greg"]:.2f} s / 4 '
                       f'+ Dreg {avg["Dreg"]:.2f} s / 16), batch {b}, {args.res}x{args.res} fp32, {len(times["Dreg"])} repetition(s), scaled by {layers} layer(s); '
                       f'the global GAN phases are not run => upper bound of the CPU rate'))
    # BASELINE configs[0]: 64x64, 1 RGBA layer, batch 4, whole iterations of the base loop on the CPU
    try:
        TL = mod('training.training_loop')
        kw = TL.default_kwargs(num_gpus=1, batch_gpu=4, resolution=64, num_fp16_res=0)
        kw.training_set_kwargs.num_samples = 64
        kw.data_loader_kwargs = dict(num_workers=0)
        if args.tiny:
            kw.G_kwargs.synthesis_kwargs.update(channel_base=512, channel_max=32)
            kw.D_kwargs.update(channel_base=512, channel_max=32)
        t1 = TL.Trainer(device='cpu', random_seed=0, **kw)
        t1.run_iteration()                  # iteration 0: all four phases (warm-up)
        t0 = time.time()
        n = 0
        while n < 4 and (n == 0 or time.time() - t0 < args.cpu_seconds / 2):
            t1.run_iteration()              # iterations 1..4: Gmain + Dmain (+ Greg at the 4th)
            n += 1
        out['config1'] = dict(value=round(4 * n / (time.time() - t0), 3), unit='img/s',
                              sample=f'BASELINE configs[0]: training_loop.Trainer, 64x64 RGBA, batch 4, fp32, {n} iteration(s) after the first, CPU aten impl=ref ops')
    except Exception as e:          # the headline does not depend on it
        out['config1'] = dict(error=repr(e))
    return out


def resolve_batch(args, world):
    """--global-batch B (strong scaling: B split evenly over the ranks; --gpus 8 --global-batch 32 is BASELINE configs[4]) or
    --batch-gpu b (weak scaling, default 32 per GPU); then the settings that follow from the per-GPU batch.  -> 'strong' | 'weak'."""
    scaling = 'weak'
    if args.global_batch is not None:
        assert args.batch_gpu is None, '--global-batch and --batch-gpu are exclusive'
        assert args.global_batch % world == 0, f'--global-batch {args.global_batch} does not divide over {world} ranks'
        args.batch_gpu = args.global_batch // world
        scaling = 'strong'
    if args.batch_gpu is None:
        args.batch_gpu = 32
    if args.graphs == 'auto':
        args.graphs = 'on' if args.batch_gpu <= 16 else 'off'
    if args.graphs == 'on' and args.ada_margin != 'fixed':
        args.ada_margin = 'fixed'
    return scaling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=32, help='timed iterations (32 = two lazy-regularisation periods, SURVEY 8d)')
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--workload', default='aio', choices=['aio', 'single'])
    ap.add_argument('--batch-gpu', type=int, default=None, help='per-GPU batch (default 32: weak scaling)')
    ap.add_argument('--global-batch', type=int, default=None,
                    help='total batch over all ranks, split evenly (strong scaling): --gpus 8 --global-batch 32 is BASELINE configs[4]')
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
    ap.add_argument('--graphs', default='auto', choices=['auto', 'on', 'off'],
                    help='record every phase into a HIP graph after its first eager execution, verify it and replay it '
                         '(phase_engine.PhaseGraphs); implies --ada-margin fixed (a recorded phase cannot read the padding margin back '
                         'to the host, and the fixed bound costs ~35 ms of GPU time at batch 32).  auto = on where the step is '
                         'launch-bound (per-GPU batch <= 16), off where the GPU is the bottleneck anyway')
    ap.add_argument('--conv-table', default=None, help='write the per-problem convolution table (ms, calls, TFLOP/s) of the profiled steps to this CSV')
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
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    scaling = resolve_batch(args, world_env)
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

    # per-kernel timing of the hand-written kernels and of the convolution calls (extra EAGER steps outside the timed
    # region: a replayed graph bypasses the launch functions that record the HIP events)
    roofline = roofline_mfma = None
    prof_counts = None
    if args.prof_steps > 0:
        # every rank runs these steps (they contain the gradient all-reduces); only rank 0 records events
        CG = mod('torch_utils.ops.conv2d_gradfix')
        # same stream and same gradient exchange as the timed steps: a loop built for phase graphs stays inside its stream scope
        # and on the plain exchange, it only launches the kernels one by one instead of replaying them (round-2 advisor finding)
        trainer.eager_steps = True
        c0 = native.launch_counts()
        if rank == 0:
            native.prof_reset()
            native.prof_enable(True)
            CG.prof = []
        for _ in range(args.prof_steps):
            trainer.run_iteration()
        torch.cuda.synchronize(device)
        trainer.eager_steps = False
        c1 = native.launch_counts()
        prof_counts = {k: round((c1[k] - c0[k]) / args.prof_steps, 1) for k in c1 if c1[k] - c0[k]}
        rec = conv_rec = None
        if rank == 0:
            rec = native.prof_collect()
            native.prof_enable(False)
            native.prof_reset()
            conv_rec, CG.prof = CG.prof, None
        if rec:
            total_ms = sum(v['ms'] for v in rec.values())
            name, top = max(rec.items(), key=lambda kv: kv[1]['ms'])
            achieved = top['bytes'] / (top['ms'] * 1e-3) / 1e9
            # HBM traffic: PMC counters cannot be read from inside this process; profiles/r01_pmc_traffic.json holds the
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE measurement (separate passes, calibrated) of this kernel on its
            # dominant shape as a ratio to the algorithmic bytes; applied here to this run's bytes per launch.
            traffic = traffic_source = None
            for src in ('r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
                try:
                    pmc = json.load(open(os.path.join(ROOT, 'profiles', src)))['kernels'].get(name)
                    if pmc:
                        traffic = round(pmc['traffic_over_algorithmic'] * top['bytes'] / top['launches'])
                        traffic_source = f'profiles/{src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE ratio to algorithmic bytes on the dominant shape, applied to this run\'s bytes per launch'
                        break
                except Exception:
                    pass
            roofline = dict(bound='hbm', kernel=name, achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit='GB/s',
                            frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=traffic_source,
                            launches_per_step=round(top['launches'] / args.prof_steps, 1),
                            avg_launch_us=round(top['ms'] * 1e3 / top['launches'], 2),
                            algorithmic_bytes_per_launch=round(top['bytes'] / top['launches']),
                            all_kernels={k: dict(ms_per_step=round(v['ms'] / args.prof_steps, 3), launches_per_step=round(v['launches'] / args.prof_steps, 1),
                                                 GBps=round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1)) for k, v in rec.items()},
                            native_ms_per_step=round(total_ms / args.prof_steps, 3))
        if conv_rec:
            # the MFMA half: every convolution call of the step (forward, data gradient, weight gradient through
            # torch_utils.ops.conv2d_gradfix; MIOpen / CK kernels plus their helper kernels inside the timed interval)
            peaks = {torch.bfloat16: ('bf16', MFMA_PEAK_TFLOPS_16), torch.float16: ('fp16', MFMA_PEAK_TFLOPS_16), torch.float32: ('fp32', MFMA_PEAK_TFLOPS_F32)}
            by = {}
            shapes = {}
            for flops, dt, ev_a, ev_b, tag in conv_rec:
                nm, pk = peaks.get(dt, (str(dt), MFMA_PEAK_TFLOPS_F32))
                ms = ev_a.elapsed_time(ev_b)
                e = by.setdefault(nm, dict(flops=0.0, ms=0.0, calls=0, peak=pk))
                e['flops'] += flops; e['ms'] += ms; e['calls'] += 1
                sh = shapes.setdefault(f'{nm} {tag}', dict(flops=0.0, ms=0.0, calls=0))
                sh['flops'] += flops; sh['ms'] += ms; sh['calls'] += 1
            if args.conv_table:
                rows = sorted(shapes.items(), key=lambda kv: -kv[1]['ms'])
                with open(args.conv_table, 'w') as fh:
                    fh.write('# convolution calls of the step by problem (HIP events around each aten call, eager steps; helper kernels included)\n')
                    fh.write('ms_per_step,calls_per_step,avg_us,TFLOPps,problem\n')
                    for k, v in rows:
                        fh.write(f"{v['ms'] / args.prof_steps:.3f},{v['calls'] / args.prof_steps:.1f},{v['ms'] / v['calls'] * 1e3:.1f},{v['flops'] / (v['ms'] * 1e-3) / 1e12:.1f},{k}\n")
            tot_f, tot_ms = sum(e['flops'] for e in by.values()), sum(e['ms'] for e in by.values())
            ideal_ms = sum(e['flops'] / (e['peak'] * 1e12) * 1e3 for e in by.values())
            main_nm = max(by, key=lambda k: by[k]['flops'])
            m = by[main_nm]
            roofline_mfma = dict(bound='mfma', ops='all convolution calls of the step (fwd + dgrad + wgrad; HIP events around each aten call, helper kernels included)',
                                 dtype=main_nm, achieved=round(m['flops'] / (m['ms'] * 1e-3) / 1e12, 1), peak=m['peak'], unit='TFLOP/s',
                                 frac=round(m['flops'] / (m['ms'] * 1e-3) / 1e12 / m['peak'], 4),
                                 by_dtype={k: dict(TFLOP_per_step=round(e['flops'] / args.prof_steps / 1e12, 2), ms_per_step=round(e['ms'] / args.prof_steps, 2),
                                                   calls_per_step=round(e['calls'] / args.prof_steps, 1), TFLOPps=round(e['flops'] / (e['ms'] * 1e-3) / 1e12, 1),
                                                   peak=e['peak'], frac=round(e['flops'] / (e['ms'] * 1e-3) / 1e12 / e['peak'], 4)) for k, e in by.items()},
                                 conv_TFLOP_per_step=round(tot_f / args.prof_steps / 1e12, 2), conv_ms_per_step=round(tot_ms / args.prof_steps, 2),
                                 conv_ms_per_step_at_peak=round(ideal_ms / args.prof_steps, 2),
                                 whole_step_TFLOPps=round(tot_f / args.prof_steps / (elapsed / args.steps) / 1e12, 1))

    if rank == 0:
        imgs = args.batch_gpu * world * args.steps
        out = {
            'metric': 'img/sec (full G+D step), 8-layer 256^2 RGBA' if args.workload == 'aio' else 'img/sec (full G+D step), single StyleGAN2 256^2 RGBA',
            'value': round(imgs / elapsed, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 2), 'host_enqueue_ms_per_step': round(host_enqueue / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': workload + (' [TINY channel counts: smoke only]' if args.tiny else ''), 'global_batch': args.batch_gpu * world,
                       'batch_per_gpu': args.batch_gpu, 'layers': args.layers if args.workload == 'aio' else 1, 'resolution': args.res,
                       'layout': f'{args.layout} in the {args.dtype} blocks (fp16_channels_last)' if args.dtype != 'fp32' else 'nchw',
                       'conv_solver_selection': {'tuned': 'MIOpen immediate mode over the find-db shipped in tuning/miopen', 'immediate': 'MIOpen immediate-mode heuristics',
                                                 'find': 'MIOpen exhaustive find in this run'}[args.miopen],
                       'parallelism': f'dp{world} (per phase: gradient mean over RCCL through one persistent flat buffer per optimizer -- ' + ('bucketed, started from inside the backward pass' if args.graphs != 'on' else 'one all-reduce between the two recorded halves of the phase') + ')',
                       'phase_graphs': (dict(recorded_and_verified=trainer.graphs.captured, replays=trainer.graphs.replays,
                                            left_eager={str(k): v[:200] for k, v in trainer.graphs.rejected.items()},
                                            verification='every recorded phase: 3 replays from saved state within max(2 %, 4 x the spread of a second EAGER execution) of the eager result on gradients / moments / buffers',
                                            worst_eager_spread=round(max([v['eager_spread'] for v in trainer.graphs.report.values()] or [0.0]), 4),
                                            worst_replay_deviation=round(max([d for k, v in trainer.graphs.report.items() if k in trainer.graphs._graphs for d in v['replays']] or [0.0]), 4))
                                       if trainer.graphs is not None else 'off'),
                       'hbm_allocated_GB': round(torch.cuda.max_memory_allocated(device) / 1e9, 1),
                       'ada_margin': args.ada_margin if not args.no_ada else None,
                       'native_launches_per_step': prof_counts if prof_counts is not None else {k: round((counts1[k] - counts0[k]) / args.steps, 1) for k in counts1 if counts1[k] - counts0[k]}},
            'roofline': roofline,
            'roofline_mfma': roofline_mfma,
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
