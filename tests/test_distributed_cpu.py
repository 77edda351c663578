"""Data-parallel path on CPU: world_size 2, gloo, 127.0.0.1.  Two ranks each run the (oracle) generator on their
shard of one synthetic batch; GradReducer (bucketed, hook-driven all-reduce over the flat gradient buffer) must
produce exactly the average of the per-shard gradients computed by a single process looping over the shards --
the semantics documented in scene_generation_amd/parallel.py / SURVEY 8e."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _small_model():
    from oracle import sg_oracle as O
    from scene_generation_amd.synthetic import make_vocab, fill_deterministic
    torch.manual_seed(0)
    m = O.Model(make_vocab(12, 4, 35), image_size=(16, 16), gconv_hidden_dim=32, gconv_num_layers=2, mask_size=8,
                appearance_normalization='batch', activation='leakyrelu-0.2', n_downsample_global=1, use_attributes=True,
                pool_size=0, ngf=4, n_blocks_global=1)
    fill_deterministic(m)
    m.noise_override = torch.linspace(-1, 1, 64).view(1, -1)
    return m


def _loss(model, b):
    out = model(b.imgs, b.objs, b.triples, b.obj_to_img, boxes_gt=b.boxes, masks_gt=b.masks, attributes=b.attributes)
    return out[0].pow(2).mean() + (out[1] - b.boxes).pow(2).mean() + out[2].mean()


def _batch():
    from scene_generation_amd.synthetic import make_batch
    return make_batch(N=4, min_objs=2, max_objs=3, size=16, mask_size=8, num_objs=12, num_preds=4, seed=3)


def _worker(rank, world, port, bucket_bytes, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from scene_generation_amd.optim import FlatParams
    from scene_generation_amd.parallel import GradReducer, init_distributed, broadcast_params
    from scene_generation_amd.synthetic import shard_batch
    r, w = init_distributed('gloo')
    assert (r, w) == (rank, world)
    model = _small_model()
    if rank == 1:                                     # perturb rank 1: broadcast must restore rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    fp = FlatParams(model.parameters())
    broadcast_params(fp)
    red = GradReducer(fp, bucket_bytes=bucket_bytes)
    for step in range(2):                             # two steps: bookkeeping must reset between them
        # a backward OUTSIDE the armed window (e.g. discriminator parameters touched by the generator's backward) must
        # neither launch collectives nor disturb the bookkeeping of the step that follows
        fp.grad.zero_()
        _loss(model, shard_batch(_batch(), rank, world)).backward()
        assert not red._works and not any(red._ready)
        fp.grad.zero_()
        red.begin_step()                              # what FusedAdam.zero_grad() does through its zero_grad_hooks
        _loss(model, shard_batch(_batch(), rank, world)).backward()
        if bucket_bytes == 4096:
            assert len(red._works) >= 2, 'bucket all-reduces must be launched from the hooks, during backward'
        for i in range(len(fp.params)):
            red.param_ready(i)                        # a second report of the same parameter is a no-op
        red.wait()
    q.put((rank, fp.packed(fp.grad).detach().numpy().copy(), fp.packed(fp.flat).detach().numpy().copy(), len(red.buckets)))   # by value
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('bucket_bytes', [1 << 30, 4096])
def test_two_rank_gradient_mean_equals_sequential_shards(bucket_bytes):
    from scene_generation_amd.synthetic import shard_batch
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: loop over the shards, average the gradients
    model = _small_model()
    grads = []
    for r in range(world):
        model.zero_grad()
        _loss(model, shard_batch(_batch(), r, world)).backward()
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                for p in model.parameters()]))
    want = sum(grads) / world
    flat0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    res = [(r, torch.from_numpy(g), torch.from_numpy(f), nb) for r, g, f, nb in res]
    for rank, g, flat, nb in res:
        err = float((g - want).abs().max())          # workers run single-threaded: conv summation order differs
        assert err <= 1e-5 * float(want.abs().max()), 'rank %d: averaged gradient mismatch (%g)' % (rank, err)
        assert torch.equal(flat, flat0), 'rank %d: parameters were not broadcast from rank 0' % rank
        assert nb >= (2 if bucket_bytes == 4096 else 1)
    assert torch.equal(res[0][1], res[1][1]), 'ranks must hold identical averaged gradients'


def _order_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from scene_generation_amd.optim import FlatParams
    from scene_generation_amd.parallel import GradReducer, init_distributed, agree, broadcast_int, control_group
    init_distributed('gloo')
    assert control_group() is None                     # the default group already lives on the host
    params = [torch.nn.Parameter(torch.zeros(300 + 17 * i)) for i in range(9)]
    fp = FlatParams(params)
    red = GradReducer(fp, bucket_bytes=2048)           # several parameters per bucket, several buckets
    launched = []
    orig = red._launch
    red._launch = lambda b: (launched.append(b), orig(b))[1]
    g = torch.Generator().manual_seed(11 + rank)
    for step in range(2):
        fp.grad.zero_()
        red.begin_step()
        del launched[:]
        order = list(range(len(params))) if rank == 0 else torch.randperm(len(params), generator=g).tolist()
        skipped = 4 if rank == 1 else None             # rank 1 never reports parameter 4 (it holds zeros there)
        for i in order:
            if i == skipped:
                continue
            fp.grad_view(i).copy_(torch.full_like(params[i], float(rank + 1) * (i + 1)))
            red.param_ready(i)
            assert launched == sorted(launched) == list(range(len(launched))), 'buckets must be launched in bucket order'
        red.flush()
        assert launched == list(range(len(red.buckets)))
        red.flush()                                     # idempotent
        assert launched == list(range(len(red.buckets)))
        red.wait()
        assert not red.armed
    coin = broadcast_int(7 if rank == 0 else 3, 'cpu')
    flags = agree([float(rank == 0), 0.0, float(rank == 1)], dist.ReduceOp.MAX, 'cpu')
    q.put((rank, fp.packed(fp.grad).detach().numpy().copy(), coin, flags, len(red.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_launch_order_is_rank_independent():
    """Hooks fire in a different order on each rank and one rank never reports a parameter: the sequence of collectives
    (bucket 0, 1, 2, ...) must still be the same everywhere, flush() must issue the rest without waiting, and the mean
    must come out right.  Also the host-side agreement helpers (coin broadcast, flag OR)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [300 + 17 * i for i in range(9)]
    want = torch.cat([torch.full((n,), ((1.0 * (i + 1)) + (0.0 if i == 4 else 2.0 * (i + 1))) / 2) for i, n in enumerate(sizes)])
    for rank, g, coin, flags, nb in res:
        assert nb >= 3
        assert torch.equal(torch.from_numpy(g), want), 'rank %d' % rank
        assert coin == 7 and flags == [1.0, 0.0, 1.0]


def _ctrl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      SG_CTRL_GLOO='force')
    torch.set_num_threads(1)
    from scene_generation_amd.parallel import init_distributed, agree, broadcast_int, control_group
    init_distributed('gloo')
    g = control_group()
    assert g is not None and g is control_group()       # created once, on every rank, and agreed upon
    coin = broadcast_int(11 if rank == 0 else 5, 'cpu')
    flags = agree([float(rank), 1.0 - rank], dist.ReduceOp.MAX, 'cpu')
    q.put((rank, coin, flags))
    dist.barrier()
    dist.destroy_process_group()
    assert control_group() is None                      # a destroyed process group takes the control group with it


def test_host_side_control_group():
    """the separate gloo group the RCCL runs use for the per-step agreement values (created here next to a gloo default
    group: SG_CTRL_GLOO=force), its all-ranks availability check and the two helpers that travel over it"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ctrl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, coin, flags in res:
        assert coin == 11 and flags == [1.0, 1.0]
