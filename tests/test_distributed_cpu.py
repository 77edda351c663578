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


# ---- world size 8 at the configs[2] sizes: the bucket / shard logic of the first 8-GPU run ----------
def _generator_param_shapes():
    """shapes of the DEFAULT-width generator's parameters, in optimiser order (191 M parameters = 765 MB of fp32)"""
    from scene_generation_amd.args import parser
    from scene_generation_amd.model import Model
    from scene_generation_amd.synthetic import make_vocab
    args = parser.parse_args(['--output_dir', '/tmp/o'])
    with torch.device('meta'):
        m = Model(vocab=make_vocab(), image_size=args.image_size, embedding_dim=args.embedding_dim, gconv_dim=args.gconv_dim,
                  gconv_hidden_dim=args.gconv_hidden_dim, gconv_num_layers=args.gconv_num_layers,
                  mlp_normalization=args.mlp_normalization, appearance_normalization=args.appearance_normalization,
                  activation=args.activation, mask_size=args.mask_size, n_downsample_global=args.n_downsample_global,
                  box_dim=args.box_dim, use_attributes=args.use_attributes, box_noise_dim=args.box_noise_dim,
                  mask_noise_dim=args.mask_noise_dim, pool_size=args.pool_size, rep_size=args.rep_size)
    return [tuple(p.shape) for p in m.parameters()]


class _OptStub(object):
    """what GradReducer reads / writes on its owning optimiser (FusedAdam itself needs the GPU)"""

    def __init__(self, n):
        self._touched = [False] * n
        self.use_spill = True
        self.grad_scale = 1.0


def _world8_worker(rank, world, port, shapes, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from scene_generation_amd.optim import FlatParams
    from scene_generation_amd.parallel import GradReducer, init_distributed
    init_distributed('gloo')
    params = [torch.nn.Parameter(torch.empty(s)) for s in shapes]
    fp = FlatParams(params)
    opt = _OptStub(len(params))
    red = GradReducer(fp, optimizer=opt)                # the Trainer's generator reducer: 64 MB buckets, overlap
    assert opt.use_spill is False                       # (overlap: late contributions must not be parked in a spill buffer)
    launched = []
    orig = red._launch
    red._launch = lambda b: (launched.append(b), orig(b))[1]
    g = torch.Generator().manual_seed(100 + rank)
    n = len(params)
    # the parameter of the LAST bucket that rank 5 never touches (box_net on a use_gt == False step would look like this on a
    # rank whose coin disagreed): the OR of the flags must switch it on everywhere
    lone = red.buckets[-1][2][0]
    late_ok = late_raised = False
    for step in range(2):
        fp.grad.zero_()
        opt._touched = [False] * n
        red.begin_step()
        del launched[:]
        # gradients arrive roughly last-layer-first, with a rank-dependent jitter inside windows of 12 parameters
        order = list(reversed(range(n)))
        for w0 in range(0, n, 12):
            idx = torch.randperm(min(12, n - w0), generator=g).tolist()
            order[w0:w0 + 12] = [order[w0 + j] for j in idx]
        for i in order:
            if i == lone and rank == 5:
                continue
            fp.grad_view(i).fill_(float(rank + 1) * (1 + i % 7))
            opt._touched[i] = True
            red.param_ready(i)
            assert launched == list(range(len(launched))), 'buckets must leave in bucket order on every rank'
            if step == 0 and not late_ok and not red._launched[red.bucket_of[i]]:
                red.late_contribution(i)                # bucket still local: a second contribution is harmless
                late_ok = True
            if step == 0 and not late_raised and red._launched[red.bucket_of[i]]:
                try:
                    red.late_contribution(i)
                except RuntimeError:
                    late_raised = True
        assert len(launched) >= len(red.buckets) - 1, 'all but the last bucket leave from the hooks, during the backward'
        red.flush()
        assert launched == list(range(len(red.buckets)))
        red.wait(defer_scale=True)
        assert opt.grad_scale == 1.0 / world and all(opt._touched), 'flags are OR-ed across ranks'
        opt.grad_scale = 1.0                            # (what FusedAdam.step() does)
    assert late_ok and late_raised
    # the SUM over ranks sits in the buffer (the Adam kernel applies 1 / world): 36 * (1 + i % 7), 31 * ... for the lone one
    bad = 0
    for i in range(n):
        want = (sum(range(1, world + 1)) - (6 if i == lone else 0)) * (1 + i % 7)
        v = fp.grad_view(i)
        bad += int(not bool((v == float(want)).all()))
    sizes = [(e - s) * 4 for s, e, _ in red.buckets]
    q.put((rank, bad, len(red.buckets), min(sizes[:-1]), sum(sizes)))
    dist.barrier()
    dist.destroy_process_group()


def test_world8_generator_buckets_and_config3_shards():
    """configs[2] (batch 256 across 8 ranks, 32 images each) without the hardware: (1) ``shard_batch`` partitions the collated
    256-image batch into eight self-contained 32-image batches; (2) eight gloo ranks run the generator's GradReducer over a
    flat gradient buffer with the real parameter list (765 MB, >= 10 buckets of >= 64 MB): hooks fire in rank-dependent
    order, buckets leave strictly in order during the 'backward', a late contribution is accepted before and refused after
    its bucket left, one rank misses a parameter (flag OR), the 1 / world factor is handed to the optimiser."""
    from scene_generation_amd.synthetic import make_config_batch, shard_batch
    world = 8
    big = make_config_batch('c2', seed=31, N=256)
    seen_o = seen_t = 0
    for r in range(world):
        s = shard_batch(big, r, world)
        assert s.imgs.size(0) == 32 and torch.equal(s.imgs, big.imgs[32 * r:32 * r + 32])
        O, T = s.objs.numel(), s.triples.size(0)
        assert 32 * 4 <= O <= 32 * 9 and 32 * 6 <= T <= 32 * 16                     # 4..9 nodes, 6..16 triples per image
        assert int(s.obj_to_img.min()) == 0 and int(s.obj_to_img.max()) == 31
        assert bool((s.obj_to_img[1:] >= s.obj_to_img[:-1]).all()), 'nodes stay contiguous per image, images ascending'
        assert int(s.triples[:, [0, 2]].min()) >= 0 and int(s.triples[:, [0, 2]].max()) < O
        assert torch.equal(s.obj_to_img[s.triples[:, 0]], s.triple_to_img), 'a triple stays inside its image'
        assert torch.equal(s.objs, big.objs[seen_o:seen_o + O]) and torch.equal(s.boxes, big.boxes[seen_o:seen_o + O])
        assert torch.equal(s.triples[:, 1], big.triples[seen_t:seen_t + T, 1])
        assert torch.equal(s.triples[:, 0] + seen_o, big.triples[seen_t:seen_t + T, 0])
        seen_o, seen_t = seen_o + O, seen_t + T
    assert seen_o == big.objs.numel() and seen_t == big.triples.size(0)

    shapes = _generator_param_shapes()
    assert sum(int(torch.Size(s).numel()) for s in shapes) > 180e6
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, shapes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, bad, nb, min_bucket, total in res:
        assert bad == 0, 'rank %d: %d parameters hold a wrong sum' % (rank, bad)
        assert nb >= 10 and min_bucket >= (64 << 20) and total > 760e6
