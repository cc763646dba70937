"""bench.py -- denoised frames/sec of the DDPM sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: WORLD_SIZE is set) or, when WORLD_SIZE is unset, this process spawns them itself through
torch.distributed.run and relays rank 0's JSON line -- the parent never touches the GPU.

A "step" is ONE reverse-diffusion step (Unet3D forward + p_sample) over the per-GPU batch, replayed from a hipGraph.
Workload = BASELINE.json configs[1] ("config_v2_2 Unet3D dim=64, 16-frame 64x64, 1000-step p_sample_loop bf16"),
synthetic x_T ~ Philox N(0,1), random-init weights.  metric value = N * B * F / (T * seconds_per_step).
Sampling is batch-parallel (reference gaussian_diffusion.py:290-301): ranks are independent, no data-path collective.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_STEPS = 1000
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}      # dense peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILES = ('r03_traffic.json',)        # newest first; keyed by 'kernel | shape' (tools/traffic.sh)


def time_kernels_in_step(run_eager, dev, steps=5):
    """Roofline leg: HIP-event duration of EVERY kernel launch INSIDE the real denoising step.  The library calls a hook before /
    after each launch (vdx_set_launch_hook, include/vdx.h) with the kernel's name, its template arguments + operand shape and the
    algorithmic FLOPs / bytes of that launch; the hook records a torch event on the launch stream, so each launch of `steps` eager
    (not graph-replayed) steps of the timed region's own loop is bracketed by its own pair.  Records are keyed by (kernel, shape):
    one symbol serves several levels of the network and its per-launch time differs 4x between them.
    Why not a stand-alone replay: back-to-back repetitions of one MFMA + HBM-heavy shape on random data run 25-80 % above the
    same kernel's duration inside the step (rocprofv3 traces of both) -- sustained identical launches pull the clocks down in a
    way the step's kernel mix does not.
    Returns (per_key: {(kernel, shape): {ms, flops, bytes, launches}} per step, sequence of one step, event overhead in us)."""
    import ctypes as C
    import torch
    from video_diffusion_nnx_amd import _lib as L

    class Info(C.Structure):
        _fields_ = [('kernel', C.c_char_p), ('shape', C.c_char_p), ('flops', C.c_double), ('bytes', C.c_double)]
    HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(Info), C.c_void_p)
    set_hook = L._sig('vdx_set_launch_hook', None, [HOOK, C.c_void_p])
    st = torch.cuda.current_stream(dev)
    records, open_ev = [], []

    def hook(user, phase, info, stream):
        e = torch.cuda.Event(enable_timing=True)
        e.record(st)
        if phase == 0:
            open_ev.append(e)
        else:
            i = info.contents
            records.append((i.kernel.decode(), i.shape.decode(), i.flops, i.bytes, open_ev.pop(), e))
    cb = HOOK(hook)
    run_eager(1)                                       # eager path warm-up (first-call attributes), not recorded
    torch.cuda.synchronize(dev)
    # what an event pair with NOTHING between its records reads on this stream (the two marker packets themselves): subtracted
    pairs = []
    for _ in range(32):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); b.record(st)
        pairs.append((a, b))
    torch.cuda.synchronize(dev)
    overhead_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]
    set_hook(cb, None)
    try:
        run_eager(steps)
        torch.cuda.synchronize(dev)
    finally:
        set_hook(C.cast(None, HOOK), None)
    per_key = {}
    for k, sh, fl, by, e0, e1 in records:
        d = per_key.setdefault((k, sh), dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, samples=[]))
        d['samples'].append(max(e0.elapsed_time(e1) - overhead_ms, 0.0)); d['flops'] += fl / steps; d['bytes'] += by / steps; d['launches'] += 1
    for d in per_key.values():
        # per-launch time of a row = the MEDIAN of its launches over the recorded steps (a row has >= `steps` samples): one stalled launch
        # (a 5 ms hiccup was seen once on a shared box) must not make a 85 us kernel the step's "dominant" row
        smp = sorted(d.pop('samples'))
        d['launches'] = d['launches'] // steps
        d['ms'] = smp[len(smp) // 2] * d['launches']
    n = len(records) // steps
    sequence = [{'kernel': k, 'shape': sh, 'flops': fl, 'bytes': by} for k, sh, fl, by, _, _ in records[:n]]
    return per_key, sequence, overhead_ms * 1e3


def cpu_baseline(dim, frames, size, budget_s=20.0, train=True):
    """CPU restatement (PyTorch-CPU, NOT JAX) of the same UNet forward at B=1, timed on this node's host cores; `train`: one
    p_losses forward + backward of the restatement (torch autograd) beside it."""
    import torch
    from oracle import unet3d_ref as R
    cfg = R.UnetConfig(dim=dim, channels=1)
    p = R.random_params(cfg, seed=0)
    x = torch.randn(1, 1, frames, size, size)
    t = torch.tensor([500])
    cores = min(len(os.sched_getaffinity(0)), os.cpu_count() or 1, 16)     # the box's CPU share, not the host's core count
    torch.set_num_threads(cores)
    with torch.no_grad():
        R.unet_forward(p, cfg, x, t)                                                     # warm-up
        n, t0 = 0, time.time()
        while n < 3 or (time.time() - t0 < budget_s and n < 10):
            R.unet_forward(p, cfg, x, t); n += 1
        dt = (time.time() - t0) / n
    out = dict(value=frames / (T_STEPS * dt), unit='frames/s', cores=cores, kind='port',
               sample=f'{n} Unet3D forwards (B=1, fp32, oracle/unet3d_ref.py on torch-CPU, {dt*1e3:.0f} ms each) extrapolated x{T_STEPS} steps')
    if train:
        leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        noise = torch.randn(1, frames, size, size, 1)
        t0 = time.time()
        loss = ((R.unet_forward(leaves, cfg, x, t) - noise) ** 2).mean()
        torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
        dtt = time.time() - t0
        out['train'] = dict(value=1.0 / dtt, unit='samples/s', cores=cores, kind='port',
                            sample=f'1 l2 p_losses forward + backward (B=1, fp32, torch autograd through oracle/unet3d_ref.py, {dtt*1e3:.0f} ms), no optimizer step')
    return out


def log(msg):
    print(f'[bench +{time.time() - _T0:6.1f}s] {msg}', file=sys.stderr, flush=True)


_T0 = time.time()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('VDX_BENCH_BATCH', 64)), help='videos per GPU (the sampling batch is free; end of round 2 on one box: 64 -> 45.5, 96 -> 46.3, 128 -> 46.6 frames/s)')
    ap.add_argument('--mode', default=os.environ.get('VDX_BENCH_MODE', 'bf16'), choices=['bf16', 'f32'])
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--act-storage', default=os.environ.get('VDX_BENCH_ACT', 'auto'), choices=['auto', 'f32', 'bf16'],
                    help='storage of the inter-kernel activations (auto: bf16 in bf16 mode, as GaussianDiffusion.sample does)')
    ap.add_argument('--attn-fp8', action='store_true', help='bf16 mode: fp8 (e4m3) QK^T / PV in the <= 16-token attention blocks (BASELINE configs[4]); off by default')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-y-shape', action='store_true', help='skip the secondary timing of the YAML-literal config_v2_2 (dim 32, 10 frames)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the secondary timings of BASELINE.json configs[3] (dim 128, 32f x 128 x 128, DDIM, fp16 operands) and configs[4] (text-conditioned, classifier-free guidance, fp8 attention)')
    ap.add_argument('--launch-seq', default='', help='write the (kernel, shape, flops, bytes) sequence of one denoising step to this JSON file (tools/shape_table.py joins it with a rocprofv3 kernel trace)')
    ap.add_argument('--no-train', action='store_true', help='skip the training leg (p_losses fwd+bwd, bucketed all-reduce, Adam/EMA)')
    ap.add_argument('--comm', default=os.environ.get('VDX_COMM', 'torch'), choices=['torch', 'abi'], help="gradient all-reduce of the training leg: torch.distributed (RCCL under 'nccl') or the communicator behind the C ABI (vdx_allreduce_bucket)")
    ap.add_argument('--train-batch', type=int, default=4, help='training samples per GPU (BASELINE.json configs[2]: 4)')
    ap.add_argument('--train-steps', type=int, default=10)
    return ap.parse_args(argv)


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start N ranks through torch.distributed.run (one per GPU, RCCL over xGMI) as a child
    process and pass its output through.  Nothing here imports torch or initialises HIP: the parent only waits."""
    import socket
    import subprocess
    with socket.socket() as sk:                                  # a free rendezvous port on the loopback interface
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f'spawning {args.gpus} ranks: {" ".join(cmd)}')
    return subprocess.call(cmd, env=env)


def train_leg(args, dev, world, rank):
    """Secondary metric: training samples/s of BASELINE.json configs[2] -- p_losses forward + staged backward with the
    bucketed gradient all-reduce (RCCL) issued per finished stage + Adam/EMA, batch 4 per GPU at the north-star shape.
    allreduce_exposed_ms = step time with the all-reduce minus step time without it (max over ranks both)."""
    import tempfile
    import torch
    import torch.distributed as dist
    from video_diffusion_nnx_amd import trainer as T
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.unet3d import Unet3D
    B, Fr, S = args.train_batch, args.frames, args.size
    unet = Unet3D(dim=args.dim, rngs=0, channels=1, mode=args.mode, device=dev)
    gd = GaussianDiffusion(unet, image_size=S, num_frames=Fr, channels=1, timesteps=T_STEPS, loss_type='l2')
    tmp = tempfile.mkdtemp(prefix=f'vdx_bench_r{rank}_')
    T.Trainer.comm_backend = args.comm
    tr = T.Trainer(gd, tmp, dataset_path='synthetic:8', train_batch_size=B * world, train_num_steps=10 ** 9, results_folder=tmp)
    g = torch.Generator().manual_seed(1000 + rank)                # S2 of SURVEY 8d: x0 ~ U[0,1), seed 1000 + rank
    x = torch.rand(B, 1, Fr, S, S, generator=g).to(dev)

    def timed(n, first):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n):
            tr.train_step(x, first + i)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        return dt / n * 1e3
    timed(3, 0)                                                   # warm-up (first-call kernel attributes, allocator, RCCL channels)
    ms = timed(args.train_steps, 3)
    exposed = 0.0
    if world > 1:
        tr.comm_enabled = False                                   # same step, gradient all-reduce switched off (timing only)
        try:
            timed(2, 100)
            ms_nocomm = timed(args.train_steps, 102)
        finally:
            tr.comm_enabled = True
        exposed = max(0.0, ms - ms_nocomm)
    n_buckets = len(tr.buckets)
    return {'samples_per_s': world * B / (ms * 1e-3), 'ms_per_step': ms, 'batch_per_gpu': B, 'global_batch': B * world,
            'allreduce_exposed_ms': exposed, 'comm': args.comm, 'grad_bytes': int(unet.flat_params.numel()) * 4, 'buckets': n_buckets,
            'what': 'q_sample + Unet3D fwd (bf16 activation storage kept for the backward in bf16 mode) + l2 loss + staged backward + bucketed sum all-reduce + Adam + EMA'}


def sampling_leg(args, dev, world, rank, dim, Fr, S, B, steps, warmup, roofline):
    """K graph-replayed reverse-diffusion steps of a dim / Fr x S x S network at batch B on this rank, bracketed by barrier +
    synchronize on both sides, max over ranks.  Returns (ms_per_step, act, roofline records or None)."""
    import torch
    import torch.distributed as dist
    from video_diffusion_nnx_amd import _lib as L
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion, vdx_p_sample_loop
    from video_diffusion_nnx_amd.unet3d import Unet3D
    unet = Unet3D(dim=dim, rngs=0, channels=1, mode=args.mode, device=dev, attn_fp8=args.attn_fp8)
    gd = GaussianDiffusion(unet, image_size=S, num_frames=Fr, channels=1, timesteps=T_STEPS, loss_type='l2')
    h = unet.handle(Fr, S)
    act = args.act_storage if args.act_storage != 'auto' else ('bf16' if args.mode == 'bf16' else 'f32')
    unet.act_bf16 = (act == 'bf16')
    unet.apply_activation_storage(h)
    ws = unet.workspace(B, Fr, S)
    stream = torch.cuda.Stream(device=dev)
    rec = None
    with torch.cuda.stream(stream):
        img = gd.randn((B, 1, Fr, S, S), 1000 + rank, 0)
        eps = torch.empty(B, Fr, S, S, 1, device=dev)
        t_dev = torch.full((B,), T_STEPS - 1, dtype=torch.int32, device=dev)
        step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        packed = unet.packed()

        def run(n, graph=1):
            L.check(vdx_p_sample_loop(h.ptr, L.ptr(unet.flat_params), L.ptr(packed), L.ptr(img), L.ptr(eps), L.ptr(t_dev), L.ptr(step_dev),
                                      L.ptr(gd._ptab), T_STEPS, n, 0, 1000 + rank, 1, L.ptr(ws), ws.numel(), B, graph, L.stream_ptr()))
        log(f'rank {rank}: dim {dim} {Fr}f x {S}x{S} ready (B={B}, mode={args.mode}); warm-up ...')
        run(max(warmup, 2))                            # untimed: eager step + graph capture + replays
        torch.cuda.synchronize(dev)
        log('warm-up done; timing ...')
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(steps)                                     # timed: exactly K graph replays
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        if rank == 0 and roofline:
            log('roofline leg: events around every kernel launch of 5 eager steps of the same loop ...')
            rec = time_kernels_in_step(lambda n: run(n, 0), dev)
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    if not os.environ.get('VDX_BENCH_DIAG'):          # (tools/ab_step.sh with knock-out variants of a kernel: results are garbage by construction)
        assert torch.isfinite(img).all(), 'sampling produced non-finite values'
    del img, eps, ws
    unet._ws.clear()
    torch.cuda.empty_cache()
    return elapsed / steps * 1e3, act, rec


def other_configs_leg(dev, world):
    """Secondary timings of the two BASELINE.json configurations that have no reference code (DDIM / fp16 / fp8 are extensions of this
    build, parity unpinned): per-step time of the captured loops through the public Python surface, as the difference of a long and
    a short chain (capture and first-touch costs cancel).  This rank only; no roofline claim -- these shapes run on the generic
    kernels in fp16 mode."""
    import torch
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.unet3d import Unet3D

    def timed(fn):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(dev)
        assert torch.isfinite(out).all()
        return time.perf_counter() - t0

    res = {}
    # configs[3]: dim 128, C = 3, 32 frames of 128 x 128, DDIM (100 steps in the config; 2 and 6 timed), fp16 operands (fp32 tensors)
    for mode in ('f16', 'bf16'):
        B, Fr, S, C = 1, 32, 128, 3
        unet = Unet3D(dim=128, rngs=0, channels=C, mode=mode, device=dev)
        gd = GaussianDiffusion(unet, image_size=S, num_frames=Fr, channels=C, timesteps=T_STEPS)
        gd.ddim_sample_loop((B, C, Fr, S, S), 1, steps=2)                      # warm-up: packing, workspace, capture
        t2 = timed(lambda: gd.ddim_sample_loop((B, C, Fr, S, S), 1, steps=2))
        t6 = timed(lambda: gd.ddim_sample_loop((B, C, Fr, S, S), 1, steps=6))
        ms = (t6 - t2) / 4 * 1e3
        res[f'configs3_{mode}'] = {'workload': f'dim=128 C=3 {Fr}f x {S}x{S}, DDIM eta=0 (100-step chain = 100 x this step), {mode} MFMA operands, '
                                               + ('bf16 activation storage' if mode == 'bf16' else 'fp32 tensors'),
                                   'batch_per_gpu': B, 'ms_per_step': ms, 'frames_per_s_ddim100': world * B * Fr / (100 * ms * 1e-3),
                                   'tflops': B * 6707.9e9 / (ms * 1e-3) / 1e12}
        log(f"configs[3] {mode}: {ms:.2f} ms/step at B = {B}")
        del unet, gd
        torch.cuda.empty_cache()
    # configs[4]: text-conditioned (cond_dim 768) 16f x 64 x 64, classifier-free guidance (one 2B-batched forward per step), fp8 QK^T / PV
    for fp8 in (True, False):
        B, Fr, S = 32, 16, 64
        unet = Unet3D(dim=64, rngs=0, channels=1, cond_dim=768, mode='bf16', device=dev, attn_fp8=fp8)
        cond = torch.randn(B, 768, device=dev)
        ts = {}
        for T in (2, 6):
            gd = GaussianDiffusion(unet, image_size=S, num_frames=Fr, channels=1, timesteps=T)
            gd.p_sample_loop((B, 1, Fr, S, S), 1, cond=cond, cond_scale=2.0)
            ts[T] = timed(lambda: gd.p_sample_loop((B, 1, Fr, S, S), 1, cond=cond, cond_scale=2.0))
        ms = (ts[6] - ts[2]) / 4 * 1e3
        res['configs4_fp8attn' if fp8 else 'configs4_bf16attn'] = {
            'workload': f'dim=64 C=1 cond_dim=768 {Fr}f x {S}x{S}, DDPM step with classifier-free guidance (cond_scale 2: 2B forwards per step), '
                        + ('fp8 e4m3 QK^T / PV' if fp8 else 'bf16 QK^T / PV'), 'batch_per_gpu': B, 'ms_per_step': ms,
            'frames_per_s_ddpm1000': world * B * Fr / (T_STEPS * ms * 1e-3)}
        log(f"configs[4] fp8 attention {fp8}: {ms:.2f} ms/step at B = {B} (2B forwards)")
        del unet, gd, cond
        torch.cuda.empty_cache()
    return res


def kernel_row(key, d, mode, ms_per_step):
    """One (kernel, shape) entry of the roofline table: both fractions, and which roofline binds by arithmetic intensity."""
    mfma_peak = MFMA_PEAK_TFLOPS[mode]
    tflops = d['flops'] / (d['ms'] * 1e-3) / 1e12 if d['ms'] > 0 else 0.0
    gbs = d['bytes'] / (d['ms'] * 1e-3) / 1e9 if d['ms'] > 0 else 0.0
    ai = d['flops'] / d['bytes'] if d['bytes'] > 0 else 0.0
    hbm_bound = ai < (mfma_peak * 1e12) / (HBM_PEAK_GBS * 1e9)
    return {'kernel': key[0], 'shape': key[1], 'launches_per_step': d['launches'], 'avg_launch_us': d['ms'] / max(d['launches'], 1) * 1e3,
            'ms_per_step': d['ms'], 'share_of_step': d['ms'] / ms_per_step, 'gflop_per_launch': d['flops'] / max(d['launches'], 1) / 1e9,
            'algorithmic_mb_per_launch': d['bytes'] / max(d['launches'], 1) / 1e6, 'arithmetic_intensity_flop_per_byte': ai,
            'bound': 'hbm' if hbm_bound else 'mfma', 'mfma_tflops': tflops, 'mfma_frac': tflops / mfma_peak, 'hbm_gbs': gbs, 'hbm_frac': gbs / HBM_PEAK_GBS}


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (the HIP path has no CPU fallback)'
    if args.gpus != world:
        log(f'warning: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=dev)
        backend = f'{dist.get_backend()} (RCCL), world_size {dist.get_world_size()}'
        log(f'rank {rank}/{world} on cuda:{local}: process group up: {backend}')

    B, Fr, S = args.batch, args.frames, args.size
    ms_per_step, act, rec = sampling_leg(args, dev, world, rank, args.dim, Fr, S, B, args.steps, args.warmup, not args.no_roofline)
    value = world * B * Fr / (T_STEPS * ms_per_step * 1e-3)

    line = {
        'metric': 'denoised frames/sec, 16fx64x64 1000-step DDPM', 'value': value, 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.mode, 'data': 'synthetic',
        'config': {'workload': f'config_v2_2 (north-star shape): Unet3D dim={args.dim} C=1, {Fr}f x {S}x{S}, DDPM T={T_STEPS} p_sample_loop '
                               f'(UNet forward + p_sample per step, hipGraph replay); value = n_gpus*B*F/(T*s_per_step)',
                   'batch_per_gpu': B, 'timesteps': T_STEPS, 'parallelism': f'dp{world} (independent samples, no collective)',
                   'mfma_operands': args.mode + (' (attention QK^T / PV: fp8 e4m3)' if args.attn_fp8 else ''), 'storage': f'weights fp32 master, activations {act}, fp32 accumulate',
                   'process_group': backend or 'none (single process)'},
    }
    log(f'timed region done: {ms_per_step:.3f} ms/step')
    if not args.no_y_shape:
        # secondary: the YAML-literal config_v2_2 (configs/config_v2_2.yaml: dim 32, 10 frames, 64 x 64) -- what `sample.py --config
        # configs/config_v2_2.yaml` runs; same loop, same batch, fewer steps
        yms, _, _ = sampling_leg(args, dev, world, rank, 32, 10, 64, B, max(5, args.steps // 2), 3, False)
        line['y_shape'] = {'workload': f'configs/config_v2_2.yaml as written: Unet3D dim=32 C=1, 10f x 64x64, T={T_STEPS}', 'batch_per_gpu': B,
                           'ms_per_step': yms, 'frames_per_s': world * B * 10 / (T_STEPS * yms * 1e-3),
                           'tflops': world * B * 54.09e9 / (yms * 1e-3) / 1e12, 'n_shape_tflops': world * B * 250.77e9 / (ms_per_step * 1e-3) / 1e12}
        log(f"Y shape: {yms:.3f} ms/step, {line['y_shape']['frames_per_s']:.1f} frames/s")
    if rank == 0 and not args.no_other_configs:
        try:                                              # secondary object: a failure here must not cost the line its primary fields
            line['other_configs'] = other_configs_leg(dev, world)
        except Exception as e:                            # noqa: BLE001
            line['other_configs'] = {'error': f'{type(e).__name__}: {e}'}
            log(f'other_configs leg failed: {e}')
    if not args.no_train:
        log('training leg (p_losses fwd+bwd + bucketed all-reduce + Adam/EMA) ...')
        line['train'] = train_leg(args, dev, world, rank)         # every rank takes part (collectives inside)
        log(f"training leg done: {line['train']['ms_per_step']:.2f} ms/step, {line['train']['samples_per_s']:.1f} samples/s")
    if rank == 0 and rec is not None:
        per, sequence, ev_overhead_us = rec
        rows = sorted((kernel_row(k, d, args.mode, ms_per_step) for k, d in per.items()), key=lambda r: -r['ms_per_step'])
        top = rows[0]                                     # the (kernel, shape) with the largest time share of the step, over ALL kernels
        traffic, traffic_source = None, None              # HBM bytes per launch from committed rocprofv3 PMC passes of THIS workload
        for tname in TRAFFIC_FILES:
            tpath = os.path.join(ROOT, 'profiles', tname)
            if not os.path.exists(tpath):
                continue
            tj = json.load(open(tpath))
            tk = f"{top['kernel']} | {top['shape']}"
            if (tj.get('batch') == B and tj.get('mode') == args.mode and tj.get('dim') == args.dim and tj.get('act', 'f32') == act
                    and tk in tj.get('kernels', {})):
                traffic = tj['kernels'][tk]['hbm_bytes_per_launch']
                traffic_source = (f'profiles/{tname}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (FETCH doubled per '
                                  'MI355X_MICROARCH.md), committed -- NOT collected in this run')
                break
        hb = top['bound'] == 'hbm'
        line['roofline'] = {'bound': top['bound'], 'achieved': top['hbm_gbs'] if hb else top['mfma_tflops'],
                            'peak': HBM_PEAK_GBS if hb else MFMA_PEAK_TFLOPS[args.mode], 'unit': 'GB/s' if hb else 'TFLOP/s',
                            'frac': top['hbm_frac'] if hb else top['mfma_frac'], 'traffic': traffic, 'traffic_source': traffic_source,
                            'kernel': f"{top['kernel']} | {top['shape']}",
                            'how': 'HIP events around EVERY kernel launch of 5 eager steps of the timed loop (library hook vdx_set_launch_hook), keyed by '
                                   f'(kernel, template arguments + shape), per-launch time = the median of a row\'s launches; empty event pair = {ev_overhead_us:.1f} us, subtracted; the row with the largest share of the step',
                            **{k: top[k] for k in ('launches_per_step', 'avg_launch_us', 'gflop_per_launch', 'algorithmic_mb_per_launch',
                                                   'arithmetic_intensity_flop_per_byte', 'mfma_tflops', 'mfma_frac', 'hbm_gbs', 'hbm_frac', 'share_of_step')},
                            'bracketed_ms_per_step': sum(r['ms_per_step'] for r in rows),
                            'all_kernels': [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]}
        if args.launch_seq:
            json.dump({'batch': B, 'mode': args.mode, 'dim': args.dim, 'act': act, 'sequence': sequence}, open(args.launch_seq, 'w'))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log('cpu_baseline leg (oracle on host cores) ...')
        line['cpu_baseline'] = cpu_baseline(args.dim, Fr, S)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
