"""Data-parallel training over the GPUs of one node: one process per GPU, torch.distributed 'nccl' (= RCCL over
xGMI on ROCm), gradients averaged with bucketed all-reduces over slices of each optimiser's FLAT gradient buffer
(optim.FlatParams), launched from autograd hooks while the rest of the backward is still running.

The batched scene graph is block-diagonal (node ids are offset per image, coco.py:527-529), so the minibatch shards
by image with no data-path exchange (synthetic.shard_batch); the only collective is the gradient mean.  What is
NOT exchanged -- BatchNorm batch statistics, VectorPool contents, the noise row -- is per rank, i.e. the semantics
are "the reference run independently on each shard, gradients averaged" (SURVEY 8e).  The use_gt coin (train.py:195)
decides which parameters receive gradients, so it is drawn once per step on rank 0 and broadcast (Trainer.draw_use_gt);
as a second line of defence the reducer ORs the optimisers' "received a gradient" flags across ranks.

Two planes: gradients travel over RCCL on the device; the tiny per-step AGREEMENT values (the coin, the flags) travel over
a gloo control group on the host (``control_group()``), so that agreeing never drains the GPU queue -- the host runs
~40 ms ahead of the device on this step and a blocking host-side exchange costs nothing, a ``.item()`` would cost a bubble.

xGMI note: collectives are per-link bound (7 links x ~153 GB/s), so buckets are LARGE (default 64 MB): a few big
reduce-scatter/all-gather rings amortise the per-collective latency; the 764.7 MB generator gradient is ~12 buckets.
"""
import os

import torch
import torch.distributed as dist

from . import streams


_CTRL = {'group': None, 'tried': False, 'default_pg': None}


def control_group():
    """Host-side (gloo) group for the per-step agreement values, or None (then they go through the default group on the
    device and cost one synchronisation each).  Collective: the first call must happen on every rank (init_distributed)."""
    if not dist.is_initialized():
        _CTRL['group'], _CTRL['tried'], _CTRL['default_pg'] = None, False, None     # (a later init_process_group starts over)
        return None
    pg = dist.distributed_c10d._get_default_group()
    if _CTRL['default_pg'] is not pg:
        # the default process group was destroyed and re-created since the control group was made: a group built on the
        # old one is stale (its ranks / store are gone) -- start over for THIS process group
        _CTRL['group'], _CTRL['tried'], _CTRL['default_pg'] = None, False, pg
    if not _CTRL['tried']:
        _CTRL['tried'] = True
        want = os.environ.get('SG_CTRL_GLOO', '1')       # '0': never; 'force': even when the default group is gloo (tests)
        if dist.get_world_size() > 1 and want != '0' and (dist.get_backend() != 'gloo' or want == 'force'):
            try:
                _CTRL['group'] = dist.new_group(backend='gloo')
            except Exception as e:                       # no usable interface: fall back to the device path
                print('scene_generation_amd.parallel: no gloo control group (%s); agreement values use the device group' % e)
                _CTRL['group'] = None
            # every rank must take the same route from here on: use the host group only if ALL ranks have one
            ok = torch.tensor([1.0 if _CTRL['group'] is not None else 0.0],
                              device='cuda' if torch.cuda.is_available() else 'cpu')
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                _CTRL['group'] = None
    return _CTRL['group']


def agree(values, op, device):
    """All-reduce a short list of numbers across ranks (op: dist.ReduceOp.MAX / SUM) and return them as Python floats.
    Host-side over the control group when there is one (no GPU synchronisation), else over the default group."""
    g = control_group()
    if g is not None or dist.get_backend() == 'gloo':
        t = torch.tensor(values, dtype=torch.float32)
        dist.all_reduce(t, op=op, group=g)
        return t.tolist()
    t = torch.tensor(values, dtype=torch.float32, device=device)
    dist.all_reduce(t, op=op)
    return t.tolist()


def broadcast_int(value, device, src=0):
    """rank ``src``'s integer on every rank (host-side when the control group exists)"""
    g = control_group()
    if g is not None or dist.get_backend() == 'gloo':
        t = torch.tensor([int(value)], dtype=torch.int64)
        dist.broadcast(t, src=src, group=g)
        return int(t.item())
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src)
    return int(t.item())


def first_contact(device, timeout_s=120.0):
    """The first collective of the default group, checked: a one-element SUM all-reduce whose result must equal the world size.
    RCCL sets its rings up lazily, so a broken transport (most often the IPC mode: this driver only supports dmabuf IPC,
    ``HSA_ENABLE_IPC_MODE_LEGACY=0``; the legacy mode fails with ``hipIpcGetMemHandle: invalid argument``) otherwise shows up as
    an opaque error -- or a hang -- inside the first gradient bucket.  Collective: call on every rank right after
    ``init_process_group``."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world, backend = dist.get_world_size(), dist.get_backend()
    t = torch.ones(1, device=device if backend != 'gloo' or torch.cuda.is_available() else 'cpu')
    try:
        work = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        import datetime
        try:
            work.wait(timeout=datetime.timedelta(seconds=timeout_s))
        except TypeError:                                 # (backends whose Work.wait takes no timeout)
            work.wait()
        got = float(t.item())
    except Exception as e:
        raise RuntimeError(
            'scene_generation_amd.parallel: the first %s collective failed on rank %d of %d: %r.  Environment: '
            'HSA_ENABLE_IPC_MODE_LEGACY=%s (must be 0 on hosts whose driver only supports dmabuf IPC), MASTER_ADDR=%s, '
            'device %s, visible devices %d.  Re-run with NCCL_DEBUG=INFO for the transport RCCL picked.'
            % (backend, dist.get_rank(), world, e, os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
               os.environ.get('MASTER_ADDR'), device, torch.cuda.device_count() if torch.cuda.is_available() else 0)) from e
    if got != float(world):
        raise RuntimeError('scene_generation_amd.parallel: first all-reduce returned %r, expected %d' % (got, world))


def init_distributed(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world)."""
    # dmabuf IPC (see first_contact).  ROCr reads the variable when the runtime initialises, so the default has to be in the
    # environment BEFORE the first torch.cuda call of the process -- the package's __init__ sets it at import too; a launcher
    # should export it (setting it after torch.cuda.is_available() is too late for this process)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1:
        return 0, 1
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if not dist.is_initialized():
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
        first_contact('cuda:%d' % torch.cuda.current_device() if torch.cuda.is_available() else 'cpu')
    control_group()
    return rank, world


class GradReducer:
    """Mean all-reduce of one FlatParams gradient buffer, bucketed + overlapped with backward.

    Buckets are contiguous slices of the flat gradient in REVERSE parameter order (gradients become ready roughly
    last-layer-first).  The reducer is ARMED by the owning optimiser's ``zero_grad()`` (``begin_step``) and disarmed by
    ``wait()``: gradient hooks that fire outside that window -- e.g. a discriminator parameter touched by the
    generator's backward -- are ignored, and a parameter counts once per step no matter how often its hook fires
    (per-parameter ready flags, not a counter).  Buckets are launched strictly IN BUCKET ORDER: bucket b's async
    all-reduce is issued as soon as every parameter of buckets 0..b has reported, so the sequence of collectives is the
    same on every rank whatever order (or subset) the hooks fire in.  ``flush()`` issues the rest without waiting (the
    Trainer calls it after the generator's backward and lets the reduce run under the discriminator steps);
    ``wait()`` (called before optimizer.step) flushes, makes the current stream wait for the collectives, and scales by
    1/world.  Parameters that received no gradient this step still hold zeros, which is what the all-reduce must see.

    ``touched`` (optional, a callable returning / accepting the optimiser's per-parameter "received a gradient" flags):
    after the reduce, the flags are OR-ed across ranks so that every rank updates the same set of parameters (Adam
    skips gradient-less parameters; ranks that disagree would silently diverge).
    """

    def __init__(self, flat_params, bucket_bytes=64 << 20, group=None, overlap=True, optimizer=None):
        self.fp = flat_params
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.optimizer = optimizer
        if overlap and optimizer is not None and hasattr(optimizer, 'use_spill'):
            # a spilled (2nd, 3rd, ...) contribution only reaches the gradient slice when the backward ends -- after this
            # reducer may have sent the slice's bucket away: with overlap the optimiser must add late contributions at once,
            # where late_contribution() can see them
            optimizer.use_spill = False
        self.buckets = []            # (start, end, [param indices])
        esz = self.fp.grad.element_size()
        cur, cur_end = [], None
        for i in reversed(range(len(self.fp.params))):
            o, n = self.fp.offsets[i], self.fp.params[i].numel()
            if cur_end is None:
                cur_end = o + n
            cur.append(i)
            if (cur_end - o) * esz >= bucket_bytes:
                self.buckets.append((o, cur_end, cur))
                cur, cur_end = [], None
        if cur:
            self.buckets.append((self.fp.offsets[cur[-1]], cur_end, cur))
        self.bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self.active = True
        self.armed = False
        # observability (bench.py --gpus N): with ``profile`` on, wait() brackets its blocking part with an event pair on the
        # compute stream -- the time the step is EXPOSED to the collectives (0 when they finished under other work)
        self.profile = False
        self._stall_events = []
        self._reset()
        if self.world > 1 and overlap:
            for i, p in enumerate(self.fp.params):
                if p.requires_grad:
                    p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _reset(self):
        self._ready = [False] * len(self.fp.params)
        self._pending = [len(idxs) for _, _, idxs in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._next = 0               # buckets [0, _next) have been launched
        self._works = []

    def begin_step(self):
        """arm the reducer for the backward that follows (called from the owning optimiser's zero_grad())"""
        self._reset()
        self.armed = True

    def _make_hook(self, i):
        def hook(param):
            self.param_ready(i)
        return hook

    def param_ready(self, i):
        """parameter ``i`` has its final gradient for this step (autograd hook, or ops.deliver_param_grad)"""
        if not (self.active and self.armed and self.world > 1 and self.overlap):
            return
        if self._ready[i]:
            return                   # a repeated REPORT (hooks may fire more than once); real late contributions: see below
        self._ready[i] = True
        self._pending[self.bucket_of[i]] -= 1
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)

    def late_contribution(self, i):
        """Called (ops.GradOut.finish, through FusedAdam.late_listeners) BEFORE a second contribution is added to the gradient
        slice of parameter ``i``.  "Final on first delivery" is an assumption about the owner (every parameter used once per
        step): while the bucket is still local a later contribution is harmless, but once its all-reduce is in flight it
        would be added to a slice that is being -- or has been -- reduced and the ranks would diverge silently."""
        if self.active and self.armed and self.world > 1 and self.overlap and self._launched[self.bucket_of[i]]:
            raise RuntimeError('GradReducer(overlap=True): parameter %d receives another gradient contribution after its '
                               'bucket was sent to the all-reduce (tied / re-used parameter?): build the reducer with '
                               'overlap=False for this optimiser' % i)

    def _launch(self, b):
        assert b == self._next
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        self._next = b + 1
        streams.join_all(self.fp.grad.device, getattr(self.optimizer, 'join_exclude', ()))      # gradients written by side-stream kernels (streams.py) must be complete
        self._works.append(dist.all_reduce(self.fp.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def flush(self):
        """Issue the all-reduce of every bucket not launched yet (the backward is over); does not wait."""
        if self.optimizer is not None and hasattr(self.optimizer, 'finalize_grads'):
            self.optimizer.finalize_grads()              # (a backward that delivered nothing queued no end-of-backward callback)
        if self.world > 1 and self.active and self.armed:
            while self._next < len(self.buckets):
                self._launch(self._next)

    def wait_deferred_scale(self):
        """``wait(defer_scale=True)``: the pre-step hook the Trainer registers on the owning FusedAdam"""
        self.wait(defer_scale=True)

    def wait(self, defer_scale=False):
        """Complete the mean all-reduce of every bucket; disarms the reducer until the next begin_step().
        ``defer_scale``: leave the SUM in the gradient buffer and hand the 1 / world factor to the owning optimiser's next
        step() (FusedAdam.grad_scale: the Adam kernel multiplies while it reads the gradient) -- saves one read + write pass
        over the flat gradient buffer and one launch per optimiser and step (1.5 GB of HBM traffic for the generator)."""
        if self.optimizer is not None and hasattr(self.optimizer, 'finalize_grads'):
            self.optimizer.finalize_grads()              # lazy_zero: slices nobody wrote are zero before they are reduced
        if self.world > 1 and self.active:
            while self._next < len(self.buckets):        # (also when nobody armed the reducer: hooks were ignored)
                self._launch(self._next)
            timed = self.profile and self.fp.grad.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for w in self._works:
                w.wait()
            if timed:
                e1.record()
                self._stall_events.append((e0, e1))
            opt = self.optimizer
            if defer_scale and opt is not None and hasattr(opt, 'grad_scale'):
                opt.grad_scale = 1.0 / self.world
            else:
                self.fp.grad.mul_(1.0 / self.world)
            if opt is not None:                          # every rank must update the same parameters
                flags = agree([1.0 if t else 0.0 for t in opt._touched], dist.ReduceOp.MAX, self.fp.grad.device) \
                    if self.group is None else self._agree_in_group(opt._touched)
                opt._touched = [bool(v) for v in flags]
        self.armed = False
        self._reset()

    def exposed_ms(self, reset=True):
        """(total, count): milliseconds the compute stream spent blocked on this reducer's collectives since the last call"""
        tot = 0.0
        for e0, e1 in self._stall_events:
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        n = len(self._stall_events)
        if reset:
            self._stall_events = []
        return tot, n

    def time_buckets(self, repeats=3):
        """isolated duration of each bucket's all-reduce (nothing else on the GPU): [(bytes, ms)] in launch order.  Collective:
        call on every rank.  The gradient buffer is scratch between steps (zero_grad() precedes every backward)."""
        out = []
        esz = self.fp.grad.element_size()
        for s_, e_, _ in self.buckets:
            buf = self.fp.grad[s_:e_]
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)      # warm-up (connection set-up on first use)
            if buf.is_cuda:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(repeats):
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / repeats
            else:
                import time
                t0 = time.perf_counter()
                for _ in range(repeats):
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                ms = 1e3 * (time.perf_counter() - t0) / repeats
            out.append(((e_ - s_) * esz, ms))
        return out

    def _agree_in_group(self, touched):
        flags = torch.tensor([1.0 if t else 0.0 for t in touched], dtype=torch.float32, device=self.fp.grad.device)
        dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
        return flags.tolist()


def broadcast_params(flat_params, src=0, group=None):
    """Make every rank start from rank ``src``'s parameters (one collective over the flat buffer)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params.flat, src=src, group=group)
