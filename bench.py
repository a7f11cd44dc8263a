#!/usr/bin/env python3
"""Headline benchmark: images/sec of the 23-block RRDBNet x4 (128x128 -> 512x512), fp32 inference,
batch 16 per GPU (BASELINE.json configs[1]); one "step" = one forward of the batch, inputs resident
in HBM.  Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work.  Typed without a launcher, ``--gpus N`` (N > 1) starts the second form itself as a fresh child process BEFORE this
process has touched the GPU (the reference's one-command entry: scripts/dist_train.sh:15-16 -> utils/dist_util.py:21-25) and
exits with the child's return code; under a launcher (WORLD_SIZE in the environment) the ranks run in place.

Multi-GPU: images are independent, so ranks shard the tiles with no data-path collective (weak
scaling: 16 tiles per GPU); the only collectives are the timing barrier and the max over ranks.

The same line carries a ``secondary`` object (never the judged ``value``; timed OUTSIDE the headline's timed region, after it):
``c3_train_step`` = BASELINE configs[2]/[3], the full-size bf16 ESRGAN step (23-block bf16 generator + bf16 UNetDiscriminatorSN, batch
32 of 128x128 LR per GPU; with N ranks the data-parallel path with one RCCL all-reduce per gradient arena) and ``c5_tiled_4k`` =
BASELINE configs[4], one 4K frame through the tile-sharded path — each with its own roofline of its dominant kernel.
``--no-secondary`` skips them; ``--mode train|tiled`` and ``--dtype bf16`` print those workloads as lines of their own (tuning).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
BATCH = 16
TILE = 128
PEAK_F32_TFLOPS = 157.3   # MI355X fp32 MFMA (v_mfma_f32_32x32x2_f32), /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16); secondary --dtype bf16 run only
PEAK_HBM_GBS = 8000.0
FLOPS_PER_IMAGE = 5.8743e11


def usable_cores():
    """Cores this process may actually use: min(affinity, cgroup CPU quota).  The GPU box shows 256
    CPUs but grants a 16-core quota; 256 oneDNN threads on 16 cores would measure the scheduler."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def cpu_baseline():
    """Oracle (CPU restatement of the reference, PyTorch CPU fp32) on this host's cores, bounded sample: after one warm-up, three
    forwards each at batch 1 and batch 2 of the workload's 128x128 tiles (>= 3 repetitions, best and median)."""
    from image_restoration_amd.utils import synth
    from oracle import rrdbnet_ref as R
    sd = {k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(0, **CFG).items()}
    cores = usable_cores()
    torch.set_num_threads(cores)
    x = torch.from_numpy(synth.uniform_input(1234, (2, 3, TILE, TILE)))
    rates = {1: [], 2: []}
    with torch.no_grad():
        R.rrdbnet_forward(x[:1, :, :32, :32], sd, 4, CFG['num_block'])  # warm-up (thread pool, allocator)
        t_all = time.perf_counter()
        for b in (1, 2):
            for _ in range(3):
                if time.perf_counter() - t_all > 30 and rates[b]:
                    break
                t0 = time.perf_counter()
                R.rrdbnet_forward(x[:b], sd, 4, CFG['num_block'])
                rates[b].append(b / (time.perf_counter() - t0))
    best_b = max(rates, key=lambda b: max(rates[b]))
    chosen = sorted(rates[best_b])
    return {'value': round(chosen[-1], 4), 'median': round(chosen[len(chosen) // 2], 4), 'unit': 'images/sec', 'cores': cores,
            'kind': 'port',
            'sample': f'{len(rates[1])} + {len(rates[2])} forwards of 1 and 2 of the 16 128x128 tiles after one warm-up; best and median at '
                      f'batch {best_b}; oracle/rrdbnet_ref.py (PyTorch CPU fp32, {torch.get_num_threads()} threads)',
            'all_rates': {str(b): [round(r, 4) for r in v] for b, v in rates.items()}}


_BRACKET_MS = None
_BRACKET_SPREAD = None


def event_bracket_ms():
    """What a pair of timing events costs with NOTHING between them, recorded on the kernels' stream behind running work the way the
    profiled launches are (median of 64 pairs).  ``sr_profile_*`` brackets every launch with such a pair; the events are barrier
    packets the command processor handles before / after the dispatch, so a bracket reads the kernel's duration PLUS this constant
    (about 10 us here; rocprofv3 times the dispatch alone).  profile_launches subtracts it, so that its averages agree with the
    rocprofv3 --kernel-trace --stats summaries committed under profiles/."""
    global _BRACKET_MS, _BRACKET_SPREAD
    if _BRACKET_MS is None:
        busy = torch.zeros(1 << 24, device='cuda')
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        for _ in range(2):
            for a, b in pairs:
                busy.add_(1.0)       # a ~20 us kernel: the queue is never empty when the pair is reached, as in a profiled forward
                a.record()
                b.record()
            torch.cuda.synchronize()
        gaps = sorted(a.elapsed_time(b) for a, b in pairs)
        _BRACKET_MS = gaps[len(gaps) // 2]
        _BRACKET_SPREAD = {'min_ms': round(gaps[0], 5), 'p10_ms': round(gaps[6], 5), 'median_ms': round(_BRACKET_MS, 5),
                           'p90_ms': round(gaps[57], 5), 'max_ms': round(gaps[-1], 5), 'pairs': len(gaps)}
        # the correction is only trusted while the calibration is tight: a wide spread means the queue state moved under it, and
        # the table then keeps the raw bracketed averages (the conservative side: a longer duration, a lower fraction)
        if gaps[57] - gaps[6] > 0.5 * _BRACKET_MS:
            _BRACKET_SPREAD['rejected'] = 'p90 - p10 above half the median: no correction applied'
            _BRACKET_MS = 0.0
    return _BRACKET_MS


def profile_launches(fn, peak_tflops=PEAK_F32_TFLOPS):
    """Runs ``fn()`` once with HIP events around every conv / weight-gradient launch (sr_profile_*), on the stream the kernels run
    on, minus the cost of an empty event bracket (event_bracket_ms).  Returns per-kernel aggregates sorted by time; the first one is
    the roofline line of that workload."""
    from image_restoration_amd import _lib
    lib = _lib.load()
    bracket = event_bracket_ms()
    cap = 16384
    recs = (_lib.LaunchRecord * cap)()
    n = C.c_int(0)
    _lib.check(lib.sr_profile_start(cap), 'sr_profile_start')
    try:
        fn()
    finally:
        _lib.check(lib.sr_profile_stop(recs, cap, C.byref(n)), 'sr_profile_stop')
    agg = {}
    for r in recs[:n.value]:
        a = agg.setdefault(r.kernel_id, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        a['launches'] += 1
        a['ms'] += max(r.ms - bracket, 0.25 * r.ms)
        a['raw_ms'] = a.get('raw_ms', 0.0) + r.ms
        a['flops'] += r.flops
        a['bytes'] += r.bytes
    out = []
    for kid, a in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
        tf = a['flops'] / (a['ms'] * 1e-3) / 1e12
        gbs = a['bytes'] / (a['ms'] * 1e-3) / 1e9
        out.append({'kernel': lib.sr_kernel_name(kid).decode(), 'launches': a['launches'],
                    'avg_ms': round(a['ms'] / a['launches'], 5), 'total_ms': round(a['ms'], 4),
                    'avg_ms_bracketed': round(a['raw_ms'] / a['launches'], 5), 'event_bracket_ms': round(bracket, 5),
                    'tflops': round(tf, 2), 'flop_frac': round(tf / peak_tflops, 4),
                    'alg_gbs': round(gbs, 1), 'hbm_frac': round(gbs / PEAK_HBM_GBS, 4),
                    'alg_flops_per_launch': a['flops'] / a['launches'], 'alg_bytes_per_launch': a['bytes'] / a['launches']})
    return out


def kernel_rooflines(net, x, peak_tflops=PEAK_F32_TFLOPS):
    def one_forward():
        with torch.no_grad():
            net(x)
    return profile_launches(one_forward, peak_tflops)


def stored_counters(kernel, suffix=''):
    """HBM traffic and MFMA-busy of ``kernel`` from the committed rocprofv3 --pmc passes (NOT measured in this run: they need the
    profiler).  ``suffix`` '_c3' selects the tables of the C3 training step (same kernels, other launch sizes).  Returns (traffic
    bytes per launch | None, mfma busy | None, provenance string)."""
    traffic = mfma = None
    tpath = os.path.join(ROOT, 'profiles', f'traffic{suffix}.json')
    upath = os.path.join(ROOT, 'profiles', f'mfma_util{suffix}.json')
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get(kernel)
    if os.path.exists(upath):
        mfma = (json.load(open(upath)).get(kernel) or {}).get('mfma_util')
    src = f'stored: profiles/traffic{suffix}.json + profiles/mfma_util{suffix}.json (separate rocprofv3 --pmc passes of this bench, tools/profile_round.sh; ' \
          'FETCH_SIZE x2 + WRITE_SIZE per the gfx950 correction); null = no stored counter for this kernel'
    return traffic, mfma, src


def roofline_of(ks, peak_tflops, suffix=''):
    """The roofline object of a workload from its kernel table: dominant kernel, binding roof = the one it sits closer to."""
    k0 = ks[0]
    traffic, mfma_pmc, src = stored_counters(k0['kernel'], suffix)
    mfma_frac, hbm_frac = k0['tflops'] / peak_tflops, k0['hbm_frac']
    common = {'traffic': traffic, 'traffic_source': src, 'kernel': k0['kernel'], 'avg_launch_ms': k0['avg_ms'],
              'avg_launch_ms_note': 'HIP events around each launch on its stream minus the cost of an empty event bracket (%.4f ms, measured in '
                                    'this run); bracketed raw average %.5f ms' % (k0['event_bracket_ms'], k0['avg_ms_bracketed']),
              'launches': k0['launches'], 'share_of_profiled_time': round(k0['total_ms'] / sum(k['total_ms'] for k in ks), 4),
              'mfma_util_pmc_stored': mfma_pmc, 'event_bracket_calibration': _BRACKET_SPREAD,
              'frac_from_raw_brackets': round(max(k0['tflops'] / peak_tflops, k0['hbm_frac']) * k0['avg_ms'] / k0['avg_ms_bracketed'], 4)}
    if peak_tflops == PEAK_BF16_TFLOPS:
        common['peak_note'] = ('nominal dense bf16 peak; a bare v_mfma_f32_32x32x16_bf16 loop on random data sustains 0.65-0.75 of it on this '
                               'pool (tools/mfma_peak.hip, DESIGN.md section 7)')
    if mfma_frac >= hbm_frac:
        return dict({'bound': 'mfma', 'achieved': k0['tflops'], 'peak': peak_tflops, 'unit': 'TFLOP/s', 'frac': round(mfma_frac, 4),
                     'hbm_frac_algorithmic': k0['hbm_frac']}, **common)
    return dict({'bound': 'hbm', 'achieved': k0['alg_gbs'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(hbm_frac, 4),
                 'mfma_frac': round(mfma_frac, 4)}, **common)


def _sync(dist):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(dt, dist, dev, backend):
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def measure_train(world, rank, dev, dist, backend, *, yml, dtype, disc, disc_dtype, batch, lq, steps, warmup, profile):
    """Whole ESRGANModel.optimize_parameters steps (generator forward/backward, discriminator passes, losses, both fused Adam steps,
    EMA) on synthetic batches resident in HBM; with N ranks the model's own data-parallel path runs (replica alignment at
    construction, one RCCL all-reduce of each gradient arena per step) and per-GPU work is fixed (weak scaling)."""
    from image_restoration_amd.models import build_model
    from image_restoration_amd.utils import synth
    from image_restoration_amd.utils.options import parse, set_random_seed
    opt = parse(os.path.join(ROOT, 'training_config', yml), ROOT, is_train=True)
    opt['dist'], opt['rank'], opt['world_size'], opt['num_gpu'] = world > 1, rank, world, 1
    opt['network_g']['compute_dtype'] = dtype
    d_dtype = disc_dtype or dtype
    if disc == 'unet':
        opt['network_d'] = dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=64, skip_connection=True, compute_dtype=d_dtype)
    elif lq != 32:
        raise SystemExit('VGGStyleDiscriminator128 needs 128x128 inputs: --lq 32, or --disc unet')
    else:
        opt['network_d']['compute_dtype'] = d_dtype
    if os.environ.get('SR_BENCH_OVERLAP_G') == '0':   # tuning: G's optimiser step (and its weight gradients) before the critic phase
        opt['train']['overlap_g_wgrad'] = False
    if os.environ.get('SR_BENCH_PREFETCH_D') == '0':   # tuning: net_d(gt) of the generator phase on the caller's stream
        opt['train']['prefetch_d_real'] = False
    if os.environ.get('SR_BENCH_REUSE_D') == '0':   # tuning: every discriminator call of a step runs its own forward
        opt['train']['reuse_d_forwards'] = False
    set_random_seed(opt['manual_seed'] + rank)   # like parse_options: ranks start different, the model aligns its replicas
    model = build_model(opt)
    x_lq = torch.from_numpy(synth.uniform_input(100 + rank, (batch, 3, lq, lq))).to(dev)
    x_gt = torch.from_numpy(synth.uniform_input(200 + rank, (batch, 3, 4 * lq, 4 * lq))).to(dev)

    def step(i):
        model.update_learning_rate(i, warmup_iter=-1)
        model.feed_data({'lq': x_lq, 'gt': x_gt})
        model.optimize_parameters(i)

    for i in range(1, warmup + 1):
        step(i)
    _sync(dist)
    t0 = time.perf_counter()
    for i in range(warmup + 1, warmup + 1 + steps):
        step(i)
    _sync(dist)
    dt = _max_over_ranks(time.perf_counter() - t0, dist, dev, backend)
    log = model.get_current_log()
    assert all(v == v and abs(v) < 1e30 for v in log.values()), log
    res = {
        'metric': 'images/sec (ESRGAN training step: 23-block RRDBNet + %s, L1 + relativistic GAN loss, Adam, EMA)' % opt['network_d']['type'],
        'value': round(world * batch * steps / dt, 3), 'unit': 'images/sec', 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32' if dtype == 'fp32' else ('bf16 generator and discriminator, f32 master weights/optimiser'
                                                if d_dtype == 'bf16' else 'bf16 generator, f32 discriminator/optimiser'),
        'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[2-3]: ESRGANModel.optimize_parameters, batch %d of %dx%d LR patches per GPU'
                               % (batch, lq, lq), 'global_batch': world * batch,
                   'parallelism': 'dp%d, replicas aligned from rank 0 at construction, one all-reduce of each gradient arena per step' % world},
        'losses': {k: float(v) for k, v in log.items()},
        'd_forwards_per_step': {'calls': 5 if type(model).__name__ == 'ESRGANModel' else 2, 'run': getattr(model, 'd_forwards_run', None)}}
    if profile and world == 1:
        peak = PEAK_F32_TFLOPS if dtype == 'fp32' else PEAK_BF16_TFLOPS
        ks = profile_launches(lambda: step(warmup + steps + 1), peak)
        torch.cuda.synchronize()
        res['roofline'] = roofline_of(ks, peak, '_c3' if (disc == 'unet' and batch == 32 and lq == 128) else '_none')
        res['kernels'] = ks[:6]
    del model
    torch.cuda.empty_cache()
    return res


def measure_tiled(net, world, rank, dev, dist, backend, *, dtype, steps, warmup, tile_batch, profile):
    """BASELINE configs[4]: one 4K frame (3840x2160 LR) through the tiler — 512x512 cells + 16 px pad, the cells sharded over the
    ranks, HR crops gathered onto rank 0 by one RCCL gather (uint8 output convention).  One frame is split over all ranks, so this
    is strong scaling; a step = one whole frame assembled on rank 0."""
    from image_restoration_amd.tiling import plan_tiles, tiled_forward
    from image_restoration_amd.utils import synth
    net.set_compute_dtype(dtype)
    H, W = 2160, 3840
    img = torch.from_numpy(synth.uniform_input(77, (1, 3, H, W))).to(dev)  # the same frame on every rank

    def step(frame=img):
        return tiled_forward(net, frame, tile=512, pad=16, scale=4, max_batch=tile_batch, rank=rank, world_size=world,
                             out_dtype=torch.uint8)

    out = None
    for _ in range(warmup):
        out = step()
    _sync(dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    _sync(dist)
    dt = _max_over_ranks(time.perf_counter() - t0, dist, dev, backend)
    if rank == 0:
        assert out.shape == (1, 3, 4 * H, 4 * W) and out.dtype == torch.uint8
    from image_restoration_amd import watchdog
    res = {
        'handoff_fallbacks': watchdog.fallback_count,  # tiled_forward repeats timed-out cells on the chain launch: 0 = the fused kernel ran
        'metric': '4K frames/sec (3840x2160 -> 15360x8640 x4 SR, 23-block RRDBNet, 512x512 tiles + 16 px pad)',
        'value': round(steps / dt, 4), 'unit': 'frames/sec', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32' if dtype == 'fp32' else 'bf16', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[4]: tiled 4K frame, %d cells sharded over the ranks, uint8 crops gathered '
                               'on rank 0' % len(plan_tiles(H, W, 512, 16)), 'tile': 512, 'tile_pad': 16,
                   'parallelism': f'tile-sharded x{world}, one gather per frame'},
        'lr_megapixels_per_sec': round(steps * H * W / dt / 1e6, 3)}
    if profile and world == 1:
        peak = PEAK_F32_TFLOPS if dtype == 'fp32' else PEAK_BF16_TFLOPS
        ks = profile_launches(step, peak)      # the WHOLE frame once more, every launch bracketed (not a corner of it)
        torch.cuda.synchronize()
        res['roofline'] = roofline_of(ks, peak, '_tiled')  # no stored counters at the tiler's launch sizes: traffic null
        res['roofline']['note'] = 'kernel table of one whole frame (%d cells, at most %d per forward), launches bracketed one by one' \
            % (len(plan_tiles(H, W, 512, 16)), tile_batch)
        res['kernels'] = ks[:5]
        res['profiled_launch_ms_total'] = round(sum(k['total_ms'] for k in ks), 2)
    del out
    torch.cuda.empty_cache()
    return res


def secondary_workloads(net, args, world, rank, dev, dist, backend):
    """BASELINE configs 3/4 and 5 next to the headline, each guarded: a failure is reported in place, never raised."""
    out = {}

    def guarded(name, fn):
        args.secondary_running = name
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001
            out[name] = {'error': f'{type(exc).__name__}: {exc}'[:400]}
            torch.cuda.empty_cache()

    def infer_bf16():
        from image_restoration_amd.utils import synth
        net.set_compute_dtype('bf16')
        x = torch.from_numpy(synth.uniform_input(1234 + rank, (BATCH, 3, TILE, TILE))).to(dev)
        steps = 20
        with torch.no_grad():
            for _ in range(3):
                y = net(x)
            _sync(dist)
            t0 = time.perf_counter()
            for _ in range(steps):
                y = net(x)
            _sync(dist)
            dt = _max_over_ranks(time.perf_counter() - t0, dist, dev, backend)
        assert bool(torch.isfinite(y).all())
        from image_restoration_amd import watchdog
        watchdog.verify('bench.py c2_infer_bf16', synchronize=False)
        res = {'metric': 'images/sec (128x128->512x512 x4 SR, 23-block RRDBNet)', 'value': round(world * BATCH * steps / dt, 3),
               'unit': 'images/sec', 'steps': steps, 'ms_per_step': round(dt / steps * 1e3, 3), 'dtype': 'bf16', 'data': 'synthetic',
               'config': {'workload': 'BASELINE configs[1] in bf16 (fp32 weights rounded once, fp32 accumulation): batch 16 of 128x128 tiles per GPU'}}
        if world == 1:
            ks = kernel_rooflines(net, x, PEAK_BF16_TFLOPS)
            res['roofline'] = roofline_of(ks, PEAK_BF16_TFLOPS)
            res['kernels'] = ks[:4]
        return res

    guarded('c2_infer_bf16', infer_bf16)
    guarded('c5_tiled_4k_bf16', lambda: measure_tiled(net, world, rank, dev, dist, backend, dtype='bf16', steps=2, warmup=1,
                                                      tile_batch=args.tile_batch, profile=True))
    guarded('c5_tiled_4k_fp32', lambda: measure_tiled(net, world, rank, dev, dist, backend, dtype='fp32', steps=2, warmup=1,
                                                      tile_batch=args.tile_batch, profile=True))
    net.set_compute_dtype(args.dtype)
    guarded('c3_train_step', lambda: measure_train(world, rank, dev, dist, backend, yml='train_rrdbnet_esrgan_x4_mi355x_bf16_unet.yml',
                                                   dtype='bf16', disc='unet', disc_dtype='bf16', batch=32, lq=128, steps=3, warmup=1,
                                                   profile=True))
    # the reference's own recipe (train_ESRGAN_x4.yml:24,51-54: 128x128 ground-truth patches = 32x32 LR, VGGStyleDiscriminator128),
    # all bf16 and all fp32: what `python bench.py --mode train --lq 32 --batch 32 [--dtype bf16 --disc-dtype bf16]` prints
    guarded('recipe_train_step_bf16', lambda: measure_train(world, rank, dev, dist, backend, yml='train_rrdbnet_esrgan_x4_mi355x.yml',
                                                            dtype='bf16', disc='vgg', disc_dtype='bf16', batch=32, lq=32, steps=20,
                                                            warmup=5, profile=False))
    guarded('recipe_train_step_fp32', lambda: measure_train(world, rank, dev, dist, backend, yml='train_rrdbnet_esrgan_x4_mi355x.yml',
                                                            dtype='fp32', disc='vgg', disc_dtype=None, batch=32, lq=32, steps=10,
                                                            warmup=3, profile=False))
    return out


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: one torch.distributed.run child with N ranks (one process per GPU), started
    from a process that has made no HIP call (``import torch`` alone does not initialise the device), its output passed through,
    its return code ours.  A child process, never an exec of this one."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault('OMP_NUM_THREADS', str(max(1, usable_cores() // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('bench.py: starting %d ranks: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def ranks_view(world, rank, dev, dist, backend, dt_local, images_local):
    """What the process group itself reports, gathered onto rank 0: world size, backend, every rank's device and its own rate, so a
    scaling record shows that N ranks on N devices took part."""
    if dist is None:
        return None
    mine = {'rank': rank, 'device': torch.cuda.current_device(), 'device_name': torch.cuda.get_device_name(),
            'pci_bus': getattr(torch.cuda.get_device_properties(dev), 'pci_bus_id', None),
            'images_per_sec': round(images_local / dt_local, 3)}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    rates = [r['images_per_sec'] for r in everyone]
    return {'world_size': dist.get_world_size(), 'backend': dist.get_backend(), 'visible_devices': torch.cuda.device_count(),
            'distinct_devices': len({(r['device'], r['pci_bus']) for r in everyone}), 'per_rank': everyone,
            'images_per_sec_min': min(rates), 'images_per_sec_max': max(rates)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', choices=('fp32', 'bf16'), default='fp32',
                    help="fp32 = the BASELINE metric (default, the judged line); bf16 = secondary reduced-precision run")
    ap.add_argument('--groups', type=int, default=0, help='override sr_set_forward_groups (tuning; 0 = library default)')
    ap.add_argument('--tile-batch', type=int, default=8, help='--mode tiled: most cells per forward (tiled_forward\'s own default; equal-shape cells are dealt out in equal shares)')
    ap.add_argument('--mode', choices=('infer', 'train', 'tiled'), default='infer',
                    help='infer = the BASELINE metric (default, the judged line); train = secondary line: full ESRGAN '
                         'optimize_parameters steps (BASELINE configs 3-4), data-parallel over the ranks; tiled = secondary line: '
                         'one 4K frame through the tile-sharded path (BASELINE configs 5)')
    ap.add_argument('--batch', type=int, default=32, help='--mode train: batch per GPU')
    ap.add_argument('--lq', type=int, default=32, help='--mode train: LR patch size (gt = 4x)')
    ap.add_argument('--disc', choices=('vgg', 'unet'), default='vgg', help='--mode train: discriminator')
    ap.add_argument('--disc-dtype', choices=('fp32', 'bf16'), default=None, help='--mode train: discriminator arithmetic (default: --dtype)')
    ap.add_argument('--chain', type=int, default=0, help='tuning: dense blocks as one persistent chain launch each (sr_set_conv_chain*): 1 = 32-row tiles, 2 = 16-row tiles (bf16)')
    ap.add_argument('--rows8', type=int, default=-1, help='tuning: fp32 32-cout convs on 16- (0), 8- (1) or 4-row (2) tiles (development switch)')
    ap.add_argument('--tall64', action='store_true', help='tuning: fp32 64-cout convs on 16-row tiles (development switch)')
    ap.add_argument('--stream', type=int, default=-1, help='tuning: 0 = large few-channel bf16 convs on the per-tile kernel instead of the streaming kernel (development switch)')
    ap.add_argument('--overlap', type=int, default=-1, help='tuning: weight gradients of the backward drivers on a side stream: -1 automatic (small launches), 0 never, 1 always (development switch)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the C3 training step and the C5 tiled frame next to the headline')
    ap.add_argument('--secondary-timeout', type=float, default=300.0)
    ap.add_argument('--profile', action='store_true', help='--mode train|tiled: add the roofline of the dominant kernel')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args.gpus)   # nothing above or in self_launch touches the GPU

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    ndev = torch.cuda.device_count()
    dev = torch.device('cuda', local_rank % ndev)
    torch.cuda.set_device(dev)
    dist = None
    backend = os.environ.get('SR_BENCH_BACKEND', 'nccl')  # 'nccl' = RCCL; 'gloo' only to rehearse N>1 on a 1-GPU box
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'

    if args.groups:
        from image_restoration_amd import _lib
        _lib.check(_lib.load().sr_set_forward_groups(args.groups), 'sr_set_forward_groups')
    if args.rows8 >= 0:
        from image_restoration_amd import _lib
        _lib.load().sr_dev_set_f32_rows8(args.rows8)
    if args.tall64:
        from image_restoration_amd import _lib
        _lib.load().sr_dev_set_f32_tall64(1)
    if args.stream >= 0:
        from image_restoration_amd import _lib
        _lib.load().sr_dev_set_conv_stream(args.stream)
    if args.overlap != -1:
        from image_restoration_amd import _lib
        _lib.load().sr_dev_set_backward_overlap(args.overlap)
    if os.environ.get('SR_DEV_FUSED_ROWS8'):   # tuning: 3 = every fused dense block on 8-row tiles (34 steps per tile)
        from image_restoration_amd import _lib
        _lib.load().sr_dev_set_fused_rows8.argtypes = [C.c_int]
        _lib.load().sr_dev_set_fused_rows8(int(os.environ['SR_DEV_FUSED_ROWS8']))
    if os.environ.get('SR_DEV_MIDS_SCRATCH'):   # A/B: -1 = the inference forward stores x1..x4 whole (as the training forward must)
        from image_restoration_amd import _lib
        _lib.load().sr_dev_set_chain_mids_scratch.argtypes = [C.c_int]
        _lib.load().sr_dev_set_chain_mids_scratch(int(os.environ['SR_DEV_MIDS_SCRATCH']))
    if os.environ.get('SR_DEV_FUSED_WAVE4'):    # A/B: 1 = the lean 16-row dense blocks on four waves of four rows
        from image_restoration_amd import _lib
        _lib.load().sr_dev_set_fused_wave4.argtypes = [C.c_int]
        _lib.load().sr_dev_set_fused_wave4(int(os.environ['SR_DEV_FUSED_WAVE4']))
    if args.chain:
        from image_restoration_amd import _lib
        _lib.check(_lib.load().sr_set_conv_chain(args.chain), 'sr_set_conv_chain')
        _lib.check(_lib.load().sr_set_conv_chain_f32(1), 'sr_set_conv_chain_f32')
    def finish(line):
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if args.mode == 'train':
        return finish(measure_train(world, rank, dev, dist, backend, yml='train_rrdbnet_esrgan_x4_mi355x.yml', dtype=args.dtype,
                                    disc=args.disc, disc_dtype=args.disc_dtype, batch=args.batch, lq=args.lq, steps=args.steps,
                                    warmup=args.warmup, profile=args.profile))

    import image_restoration_amd as ira
    from image_restoration_amd.utils import synth
    net = ira.build_network(dict(type='RRDBNet', **CFG)).to(dev).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(0, **CFG).items()}, strict=True)
    if args.mode == 'tiled':
        return finish(measure_tiled(net, world, rank, dev, dist, backend, dtype=args.dtype, steps=args.steps, warmup=args.warmup,
                                    tile_batch=args.tile_batch, profile=args.profile))
    net.set_compute_dtype(args.dtype)
    x = torch.from_numpy(synth.uniform_input(1234 + rank, (BATCH, 3, TILE, TILE))).to(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            y = net(x)
        _sync(dist)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = net(x)
        _sync(dist)
        dt = time.perf_counter() - t0
    dt_local = dt
    dt = _max_over_ranks(dt, dist, dev, backend)
    assert bool(torch.isfinite(y).all())
    from image_restoration_amd import watchdog
    watchdog.verify('bench.py headline', synchronize=False)   # a timed-out dense-block launch would have produced a fast, wrong number

    value = world * BATCH * args.steps / dt
    line = {
        'metric': 'images/sec (128x128->512x512 x4 SR, 23-block RRDBNet)', 'value': round(value, 3),
        'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32' if args.dtype == 'fp32' else 'bf16', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: RRDBNet num_block=23 nf=64 x4 %s inference, batch 16 of '
                               '128x128 tiles per GPU' % args.dtype, 'global_batch': world * BATCH, 'tile': TILE,
                   'parallelism': f'tile-sharded x{world}, no data-path collective'},
        'net_tflops': round(value * FLOPS_PER_IMAGE / 1e12 / world, 2),
    }
    if world > 1:
        line['ranks'] = ranks_view(world, rank, dev, dist, backend, dt_local, BATCH * args.steps)
    if world == 1:
        peak = PEAK_F32_TFLOPS if args.dtype == 'fp32' else PEAK_BF16_TFLOPS
        ks = kernel_rooflines(net, x, peak)
        line['roofline'] = roofline_of(ks, peak)
        line['kernels'] = ks
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
    if not args.no_secondary:
        # Outside the headline's timed region.  A watchdog prints the line without them if a secondary workload hangs (every
        # rank exits), so the judged value can never be lost to a secondary number.
        import threading

        def give_up():  # the judged line is still printed, but a hang is a failure: non-zero exit, and say what was running
            line['secondary'] = {'error': f'secondary workloads did not finish within {args.secondary_timeout} s; running: '
                                          f'{getattr(args, "secondary_running", None)}'}
            if rank == 0:
                print(json.dumps(line), flush=True)
                print(f'bench.py: secondary workload {getattr(args, "secondary_running", None)} hung', file=sys.stderr, flush=True)
            os._exit(3)
        dog = threading.Timer(args.secondary_timeout, give_up)
        dog.daemon = True
        dog.start()
        line['secondary'] = secondary_workloads(net, args, world, rank, dev, dist, backend)
        dog.cancel()
    finish(line)


if __name__ == '__main__':
    main()
