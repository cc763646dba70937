"""Data-parallel train step on the GPU (reference trainer.py:161-177,307-320,361):

  1. bucket readiness: the staged backward is run stage by stage and every gradient bucket [lo, hi) is snapshotted at the moment
     GradBucketReducer would hand it to RCCL (its `stage_done`); each snapshot must be BIT-equal to the final gradient --
     a parameter whose gradient a later stage still touches (the r01 bug: every ResnetBlock's time-MLP gradients were written
     by the stem stage) fails this test;
  2. sharding: grads(shard 0) + grads(shard 1), with the 1/world fold of the optimizer read, equal the gradient of the global
     batch (mean over the global batch, gaussian_diffusion.py:464-466 / trainer.py:361);
  3. two ranks (two processes on this one GPU, gloo carrying the device tensors) take two train steps on the shards of a global
     batch and end with the parameters of ONE rank stepping on the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(ukw, frames, size, mode, seed=3):
    from oracle import unet3d_ref as R
    from video_diffusion_nnx_amd.unet3d import Unet3D
    cfg = R.UnetConfig(**ukw)
    p = R.random_params(cfg, seed=seed, dtype=torch.float32)
    m = Unet3D(rngs=0, mode=mode, **ukw)
    m.load_state_dict(p)
    return m


@pytest.mark.parametrize('ukw,frames,size,minb', [
    (dict(dim=64, channels=1), 16, 64, 4 << 20),           # north-star shape: the production bucket size (>= 16 MB)
    (dict(dim=16, channels=3, cond_dim=32), 4, 16, 1 << 14),
])
def test_buckets_are_final_when_reduced(ukw, frames, size, minb):
    from video_diffusion_nnx_amd.train_step import stage_of_param
    from video_diffusion_nnx_amd.trainer import make_buckets
    m = _setup(ukw, frames, size, 'bf16')
    nlev = len(m.dim_mults)
    ns = m.num_stages
    assert ns == stage_of_param('__count__', nlev)
    total = m.flat_params.numel()
    buckets = make_buckets(m.param_table, total, lambda nm: stage_of_param(nm, nlev), ns, min_bucket_floats=minb)
    assert len(buckets) >= 3 and sum(hi - lo for lo, hi, _ in buckets) == total
    g = torch.Generator().manual_seed(0)
    B = 2
    x = torch.randn(B, ukw['channels'], frames, size, size, generator=g)
    t = torch.tensor([11, 900])
    cond = torch.randn(B, 32, generator=g) if ukw.get('cond_dim') else None
    y = m(x, t, cond=cond)
    d_out = torch.randn(y.shape, generator=g).to(m.device)
    grads = torch.full_like(m.flat_params, float('nan'))      # the head stage must zero the whole buffer itself
    snaps, nxt = [], 0
    for stage in range(ns - 1, -1, -1):
        m.backward(d_out, grads, stage, stage)
        while nxt < len(buckets) and buckets[nxt][2] >= stage:   # GradBucketReducer.stage_done(stage)
            lo, hi, _ = buckets[nxt]
            snaps.append((lo, hi, stage, grads[lo:hi].clone()))
            nxt += 1
    assert nxt == len(buckets)
    torch.cuda.synchronize()
    assert torch.isfinite(grads).all()
    changed = []
    for lo, hi, stage, snap in snaps:
        if not torch.equal(snap, grads[lo:hi]):
            idx = (snap != grads[lo:hi]).nonzero()[:, 0] + lo
            names = sorted({n for n, s, o in m.param_table for i in idx[:64].tolist() if o <= i < o + int(np.prod(s))})
            changed.append((lo, hi, stage, names[:4]))
    assert not changed, f'buckets modified after they were handed to the reducer: {changed}'
    # and no stage writes a parameter that stage_of_param files under a LATER-running (lower) stage... covered by the above;
    # the overlap is real: all but the last buckets are ready before the stem stage
    assert sum(1 for _, _, st, _ in snaps if st > 0) >= len(snaps) - 2


@pytest.mark.parametrize('ukw,frames,size', [
    (dict(dim=64, channels=1), 16, 64),
    (dict(dim=16, channels=1, dim_mults=(1, 2)), 4, 8),
])
def test_shard_gradients_sum_to_global_batch_gradient(ukw, frames, size):
    """f32 mode: sum over shards of d(mean over shard)/dp, times 1/world, == d(mean over the global batch)/dp."""
    from video_diffusion_nnx_amd import _lib as L
    from video_diffusion_nnx_amd.train_step import vdx_loss_grad
    m = _setup(ukw, frames, size, 'f32')
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 1, frames, size, size, generator=g)
    t = torch.tensor([40, 777])
    noise = torch.randn(2, 1, frames, size, size, generator=g).to(m.device)

    def grads_of(xs, ts, ns):
        B = xs.shape[0]
        eps = m(xs, ts)
        d_eps = torch.empty_like(eps)
        fhw = frames * size * size
        L.check(vdx_loss_grad(L.ptr(eps), L.ptr(ns.contiguous()), L.ptr(d_eps), B, 1, fhw, 1, L.stream_ptr()))
        gr = torch.zeros_like(m.flat_params)
        m.backward(d_eps, gr)
        torch.cuda.synchronize()
        return gr.clone()
    full = grads_of(x, t, noise)
    parts = grads_of(x[:1], t[:1], noise[:1]) + grads_of(x[1:], t[1:], noise[1:])
    folded = parts * 0.5                                       # grad_scale = 1/world in vdx_adam_ema_step
    rel = ((folded - full).double().norm() / full.double().norm()).item()
    print(f'shard sum vs global batch {ukw}: rel-L2 {rel:.3e}')
    assert rel < 2e-5, rel


def _dp_rank(rank, world, port, ukw, frames, size, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        q.put((rank, _dp_steps(ukw, frames, size, world, rank)))
    finally:
        dist.destroy_process_group()


def _dp_steps(ukw, frames, size, world, rank, steps=2, global_batch=2):
    import tempfile
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.trainer import Trainer
    m = _setup(ukw, frames, size, 'f32')
    gd = GaussianDiffusion(m, image_size=size, num_frames=frames, channels=1, timesteps=100, loss_type='l2')
    tmp = tempfile.mkdtemp()
    Trainer.min_bucket_floats = 1 << 12                       # several buckets on this small network
    tr = Trainer(gd, tmp, dataset_path='synthetic:8', train_batch_size=global_batch, train_num_steps=steps, train_lr=1e-3,
                 results_folder=tmp, step_start_ema=0, update_ema_every=1, ema_decay=0.9)
    assert tr.world == world and tr.per_device_bs == global_batch // world
    assert len(tr.buckets) >= 2
    g = torch.Generator().manual_seed(5)
    per = global_batch // world
    losses = []
    for s in range(steps):
        x = torch.rand(global_batch, 1, frames, size, size, generator=g)
        t = torch.randint(0, 100, (global_batch,), generator=g)
        noise = torch.randn(global_batch, 1, frames, size, size, generator=g)
        sl = slice(rank * per, (rank + 1) * per)
        losses.append(float(tr.train_step(x[sl], s, t=t[sl], noise=noise[sl]).item()))
    torch.cuda.synchronize()
    return m.flat_params.cpu().numpy(), tr.ema.cpu().numpy(), losses       # numpy: pickled by value through the mp queue


def test_two_rank_train_steps_match_one_rank_on_global_batch():
    import torch.multiprocessing as mp
    ukw, frames, size = dict(dim=16, channels=1, dim_mults=(1, 2)), 4, 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    ps = [ctx.Process(target=_dp_rank, args=(r, 2, port, ukw, frames, size, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    p1, ema1, l1 = _dp_steps(ukw, frames, size, 1, 0)
    (pa, ea, la), (pb, eb, lb) = outs[0], outs[1]
    assert np.array_equal(pa, pb) and np.array_equal(ea, eb), 'replicated parameters diverged between ranks'
    # loss of the global batch = mean of the equal-size shard means
    assert np.allclose([(a + b) / 2 for a, b in zip(la, lb)], l1, rtol=1e-5)
    # two Adam steps at lr 1e-3: compare the UPDATE (first-step Adam is lr * sign(g), so near-zero gradients may flip sign)
    orig = _orig(ukw)
    upd1, upd2 = (p1 - orig).astype(np.float64), (pa - orig).astype(np.float64)
    rel = np.linalg.norm(upd2 - upd1) / np.linalg.norm(upd1)
    print(f'2-rank vs 1-rank parameter update: rel-L2 {rel:.3e}')
    assert rel < 2e-2, rel
    assert np.linalg.norm((ea - ema1).astype(np.float64)) / np.linalg.norm((ema1 - orig).astype(np.float64)) < 2e-2


def _orig(ukw):
    m = _setup(ukw, 4, 8, 'f32')
    return m.flat_params.cpu().numpy()


# ---- sampling shards by sample (reference gaussian_diffusion.py:278-298: batch split over the devices, no collective in a step) ----

def _sample_rank(rank, world, port, out_dir, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
        m = _setup(dict(dim=16, channels=1, dim_mults=(1, 2)), 4, 8, 'bf16')
        gd = GaussianDiffusion(m, image_size=8, num_frames=4, channels=1, timesteps=6)
        v = gd.sample(21, batch_size=4)                                    # GLOBAL batch 4 -> this rank's 2
        import sample                                                      # the CLI under the same group: rank r writes sample_{2r}, sample_{2r+1}
        cfg = {'unet': dict(dim=16, dim_mults=[1, 2], channels=1, rngs_seed=0, use_bert_text_cond=False),
               'diffusion': dict(image_size=8, num_frames=4, channels=1, timesteps=6, loss_type='l2'), 'trainer': {}}
        import yaml
        cfg_path = os.path.join(out_dir, f'cfg{rank}.yaml')
        open(cfg_path, 'w').write(yaml.safe_dump(cfg))
        os.environ['VDX_KEEP_PROCESS_GROUP'] = '1'
        sample.main(['--config', cfg_path, '--random-init', '--batch-size', '4', '--seed', '21', '--output-path', os.path.join(out_dir, 'gifs')])
        q.put((rank, v.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sampling_shards_the_batch(tmp_path):
    """GaussianDiffusion.sample / sample.py inside a 2-rank group (two processes on one GPU, gloo): each rank draws HALF of the
    global batch from its own Philox stream shard_key(key, rank) -- distinct videos, their union = what one process draws with the
    two shard keys -- and writes its own GIF indices."""
    import torch.multiprocessing as mp
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion, shard_key
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    ps = [ctx.Process(target=_sample_rank, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    assert outs[0].shape == outs[1].shape == (2, 1, 4, 8, 8)
    assert not np.array_equal(outs[0], outs[1]), 'both ranks drew the same videos'
    m = _setup(dict(dim=16, channels=1, dim_mults=(1, 2)), 4, 8, 'bf16')
    gd = GaussianDiffusion(m, image_size=8, num_frames=4, channels=1, timesteps=6)
    for r in range(2):
        one = gd.sample(shard_key(21, r, 2), batch_size=2).cpu().numpy()
        assert np.array_equal(one, outs[r]), f'rank {r} shard != the 1-rank draw with its shard key'
    gifs = sorted(p.name for p in (tmp_path / 'gifs').glob('sample_*.gif'))
    assert gifs == ['sample_0.gif', 'sample_1.gif', 'sample_2.gif', 'sample_3.gif'], gifs


# ---- the communicator behind the C ABI (vdx_comm_init / vdx_allreduce_bucket; RCCL inside libvdx.so) ----

def test_abi_communicator_one_rank_train_step():
    """One GPU = a 1-rank RCCL communicator: a train step whose buckets go through vdx_allreduce_bucket on the reducer's side stream
    must leave exactly the parameters of the torch.distributed-free step (sum over one rank = identity), the handle reports its world,
    a second vdx_comm_init on the same handle fails loudly.  N > 1 ranks over xGMI: unmeasured on hardware (no multi-GPU box here)."""
    import ctypes as C
    import tempfile
    from video_diffusion_nnx_amd import _lib as L
    from video_diffusion_nnx_amd.gaussian_diffusion import GaussianDiffusion
    from video_diffusion_nnx_amd.trainer import Trainer
    from video_diffusion_nnx_amd.unet3d import vdx_allreduce_bucket, vdx_comm_init, vdx_comm_unique_id, vdx_comm_world
    ukw, frames, size = dict(dim=16, channels=1, dim_mults=(1, 2)), 4, 8
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, frames, size, size, generator=g)
    t = torch.randint(0, 100, (2,), generator=g)
    noise = torch.randn(2, 1, frames, size, size, generator=g)
    res = {}
    keep = Trainer.comm_backend, Trainer.min_bucket_floats
    try:
        Trainer.min_bucket_floats = 1 << 12
        for backend in ('torch', 'abi'):
            Trainer.comm_backend = backend
            m = _setup(ukw, frames, size, 'f32')
            gd = GaussianDiffusion(m, image_size=size, num_frames=frames, channels=1, timesteps=100, loss_type='l2')
            tmp = tempfile.mkdtemp()
            tr = Trainer(gd, tmp, dataset_path='synthetic:8', train_batch_size=2, train_num_steps=2, train_lr=1e-3, results_folder=tmp)
            assert len(tr.buckets) >= 2
            for s in range(2):
                loss = tr.train_step(x, s, t=t, noise=noise)
            torch.cuda.synchronize()
            res[backend] = (m.flat_params.clone(), float(loss.item()))
            if backend == 'abi':
                h = m.handle(frames, size)
                assert vdx_comm_world(h.ptr) == 1
                buf = C.create_string_buffer(128)
                L.check(vdx_comm_unique_id(buf))
                assert vdx_comm_init(h.ptr, 0, 1, buf.raw) != 0 and b'already' in L.vdx_last_error()
                v = torch.arange(1000, dtype=torch.float32, device=m.device)
                L.check(vdx_allreduce_bucket(h.ptr, L.ptr(v), v.numel(), L.stream_ptr()))
                torch.cuda.synchronize()
                assert torch.equal(v.cpu(), torch.arange(1000, dtype=torch.float32))
            else:
                h = m.handle(frames, size)
                assert vdx_allreduce_bucket(h.ptr, L.ptr(m.flat_params), 16, L.stream_ptr()) != 0       # no communicator: loud, not silent
    finally:
        Trainer.comm_backend, Trainer.min_bucket_floats = keep
    # (weight gradients accumulate with fp32 atomics: two runs of the SAME step agree to rounding, not bitwise)
    d = (res['torch'][0] - res['abi'][0]).double().norm() / res['torch'][0].double().norm()
    assert d < 2e-4 and abs(res['torch'][1] - res['abi'][1]) < 1e-4, (d, res['torch'][1], res['abi'][1])       # (Adam's first steps are ~lr * sign(g): near-zero gradients flip)
