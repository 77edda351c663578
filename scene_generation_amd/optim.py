"""Flat-buffer Adam for MI355X.

All parameters of one optimiser live in ONE contiguous fp32 buffer (parameters become views into it), and so do
their gradients and both Adam moments.  Consequences:
  * optimizer.step()      = one HIP launch over 28 B/param of HBM traffic (sg_adam_step)
  * optimizer.zero_grad() = one fill launch -- or none (``lazy_zero``: the first contribution of a step overwrites its slice)
  * data-parallel reduce  = a handful of large RCCL all-reduces over slices of the flat gradient buffer
    (scene_generation_amd.parallel) instead of one small collective per tensor.
``state_dict()`` / ``load_state_dict()`` speak torch.optim.Adam's schema (trainer.py:138,186), so reference
checkpoints round-trip.
"""
import os

import torch

from . import ops, streams

# what the Trainer passes as FusedAdam(lazy_zero=...): SG_LAZY_ZERO=0 restores the fill launch of every zero_grad() (A/B switch)
LAZY_ZERO = os.environ.get('SG_LAZY_ZERO', '1') == '1'


class FlatParams:
    """Re-homes ``params`` into one flat buffer (+ a flat gradient buffer).  Device-agnostic host logic."""
    ALIGN = 64          # elements

    def __init__(self, params):
        self.params = [p for p in params]
        assert self.params, 'no parameters'
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            # 256-byte aligned slices: the conv kernels read weights with 16-byte vector loads only when aligned
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.flat = torch.zeros(off, dtype=dt, device=dev)     # alignment gaps stay zero under Adam (zero gradient)
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)
        self.attach_grads()

    def grad_view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def packed(self, buf):
        """``buf`` (the flat parameter / gradient / moment buffer) without the alignment gaps, in parameter order"""
        return torch.cat([buf[o:o + p.numel()] for p, o in zip(self.params, self.offsets)])

    def attach_grads(self):
        """(re)point every p.grad at its slice of the flat buffer; autograd then accumulates IN PLACE."""
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + self.offsets[i] * self.grad.element_size():
                p.grad = self.grad_view(i)


class FusedAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay=0, amsgrad=False) semantics on flat buffers.

    Like torch, a parameter that received NO gradient since the last zero_grad() is skipped entirely (no moment
    decay, no update, its step count does not advance) -- e.g. ``box_net`` in iterations with ``use_gt == False``
    (trainer.py:210-216).  "Received a gradient" is tracked with post-accumulate-grad hooks; step() launches the fused
    kernel once per maximal run of adjacent active parameters that share a step count (normally one launch)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, direct_grads=True, lazy_zero=False):
        self.fp = FlatParams(params)
        # lazy_zero (device buffers with gradient sinks only; the Trainer's four optimisers): zero_grad() does NOT fill the flat
        # gradient buffer (730 MB for the generator: 0.13 ms of HBM time per step).  Every backward kernel OVERWRITES the slice of
        # a parameter's first contribution (ops.GradOut mode 0) and Adam skips untouched parameters, so the fill only ever
        # mattered for parameters that receive nothing: their slices are zeroed when the first backward after zero_grad() ends
        # (or at step() / a reducer's wait()), whichever comes first.  Between zero_grad() and that point ``p.grad`` is None --
        # torch's ``set_to_none=True`` behaviour -- so a gradient autograd itself produces is assigned, never added to stale data.
        self.lazy_zero = bool(lazy_zero) and bool(direct_grads) and self.fp.flat.is_cuda
        self._stale, self._finalize_queued = False, False
        self._views = [self.fp.grad_view(i) for i in range(len(self.fp.params))] if self.lazy_zero else None
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        n = len(self.fp.params)
        self.steps = [0] * n
        self._touched = [False] * n
        # spill buffers: gradient contributions 2, 3, ... a parameter receives between zero_grad() and step() (ops.GradOut),
        # same layout as the gradient buffer, all-zero outside a step
        self._contrib = [0] * n
        self._spill, self._spill_used, self._fold_queued = [], 0, False
        self._spill_k, self._spill_dirty = [0] * n, False     # per parameter: spilled contributions since the last fold
        # False: later contributions are added into the gradient slice at once (temporary + axpy).  A reducer that sends
        # buckets to the all-reduce DURING the backward needs that: a spilled contribution only reaches the slice when the
        # backward ends, after the bucket may have left (parallel.GradReducer(overlap=True), Trainer)
        self.use_spill = True
        # the gradient enters the update as grad * grad_scale; a data-parallel reducer whose all-reduce SUMs sets 1 / world for
        # the step that follows (GradReducer.wait(defer_scale=True)) instead of scaling the flat buffer in place: after such a
        # step() ``p.grad`` holds the sum over ranks, not the mean.  Reset to 1 by step().
        self.grad_scale = 1.0
        # side-stream groups (streams.py) whose kernels never write this optimiser's buffers: not waited for in zero_grad() /
        # step() (the Trainer's discriminator optimisers: ('front',) -- the generator's object front runs beside their steps)
        self.join_exclude = ()
        self.pre_step_hooks = []          # e.g. GradReducer.wait
        self.zero_grad_hooks = []         # e.g. GradReducer.begin_step
        self.grad_listeners = []          # callables(i): parameter i just received (a contribution to) its gradient
        self.late_listeners = []          # callables(i): parameter i is ABOUT to receive a second contribution this step
        for i, p in enumerate(self.fp.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))
        if direct_grads and self.fp.flat.is_cuda:
            # backward kernels write parameter gradients straight into the flat buffer (ops.ParamSink) instead of
            # returning a tensor that autograd then adds into it with one ATen launch per parameter
            ops.register_param_sinks(self)

    def _make_hook(self, i):
        def hook(param):
            if self._stale:
                # lazy_zero: autograd ASSIGNED this gradient (p.grad was None): move it into the parameter's slice
                g, view = param.grad, self._views[i]
                if g is not None and g.data_ptr() != view.data_ptr():
                    if self._touched[i]:
                        view.add_(g)
                    else:
                        view.copy_(g)
                    param.grad = view
            self._on_grad(i)
        return hook

    def _on_grad(self, i):
        self._touched[i] = True
        self._contrib[i] += 1
        if self._stale and not self._finalize_queued:
            # p.grad is complete (untouched slices zero, every p.grad attached) as soon as the running ``.backward()`` returns
            try:
                from torch.autograd import Variable
                Variable._execution_engine.queue_callback(self.finalize_grads)
                self._finalize_queued = True
            except RuntimeError:                   # not inside a backward (a caller writing gradients by hand): step() finalizes
                pass
        for f in self.grad_listeners:
            f(i)

    def spill_view(self, i):
        """where the NEXT contribution to parameter ``i`` (which already has at least one) is written.  Only valid while a
        backward is running (ops.GradOut calls it from inside autograd Functions).  The index counts contributions since the
        last FOLD (every backward folds and clears the buffers when it ends), not since the last zero_grad(): gradient
        accumulation over micro-batches re-uses spill buffer 0 instead of growing one buffer per backward."""
        k = self._spill_k[i]
        self._spill_k[i] = k + 1
        self._spill_dirty = True
        while len(self._spill) <= k:
            self._new_spill_buffer()
        self._spill_used = max(self._spill_used, k + 1)
        if not self._fold_queued:
            # fold when the backward that is running right now ends: ``p.grad`` is complete as soon as ``.backward()`` returns
            # (gradient clipping, logging between backward and step see what torch would show them)
            from torch.autograd import Variable
            self._fold_queued = True
            Variable._execution_engine.queue_callback(self._fold_spill)
        p, o = self.fp.params[i], self.fp.offsets[i]
        return self._spill[k][o:o + p.numel()].view(p.shape)

    def finalize_grads(self):
        """lazy_zero: zero the slices of the parameters that received nothing since zero_grad() and re-attach every ``p.grad``.
        Idempotent; runs at the end of the first backward after zero_grad(), from step() and from GradReducer.wait()."""
        self._finalize_queued = False
        if not self._stale:
            return
        self._stale = False
        fp, t = self.fp, self._touched
        i, n = 0, len(fp.params)
        while i < n:
            if t[i]:
                i += 1
                continue
            j = i
            while j + 1 < n and not t[j + 1]:
                j += 1
            ops.fill_(fp.grad[fp.offsets[i]:fp.offsets[j] + fp.params[j].numel()], 0.0)
            i = j + 1
        for p, v in zip(fp.params, self._views):
            if p.grad is not v:
                p.grad = v

    def _new_spill_buffer(self):
        """A zeroed buffer that kernels of ANY stream may write slices of behind autograd's back.  With side streams on
        (streams.py) a zero fill still in flight on the allocating stream could wipe a slice another stream has already
        written, so the fill is completed before the buffer is handed out: one host synchronisation per buffer
        per optimiser, in the first step only (the buffers live as long as the optimiser)."""
        buf = torch.zeros_like(self.fp.grad)
        if (streams.ENABLED or streams.GROUPS) and buf.is_cuda:
            torch.cuda.current_stream(buf.device).synchronize()
        self._spill.append(buf)

    def _fold_spill(self):
        """grad = ((grad + spill[0]) + spill[1]) + ...: the order the per-parameter adds had; clears the spill buffers.
        Runs as an end-of-backward callback (and from step()): weight-gradient kernels of side streams write the gradient and
        spill slices behind autograd's back, so the reader joins them first (streams.py's rule for these buffers)."""
        self._fold_queued = False
        if self._spill_used:
            streams.join_all(self.fp.grad.device, self.join_exclude)
        for k in range(self._spill_used):
            ops.add_clear_(self.fp.grad, self._spill[k])
        self._spill_used = 0
        if self._spill_dirty:
            self._spill_k = [0] * len(self.fp.params)
            self._spill_dirty = False

    @property
    def step_count(self):
        return max(self.steps) if self.steps else 0

    @property
    def param_groups(self):
        return [dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False,
                     params=self.fp.params)]

    def zero_grad(self, set_to_none=False):
        streams.join_all(self.fp.grad.device, self.join_exclude)      # weight-gradient kernels of side streams write this buffer (streams.py)
        if self._spill_used:                        # contributions of a backward whose step never came: dropped with the rest
            for k in range(self._spill_used):
                ops.fill_(self._spill[k], 0.0)
            self._spill_used = 0
        if self.lazy_zero:
            self._stale, self._finalize_queued = True, False
            for p in self.fp.params:
                p.grad = None
        else:
            ops.fill_(self.fp.grad, 0.0)
            self.fp.attach_grads()
        self._touched = [False] * len(self.fp.params)
        self._contrib = [0] * len(self.fp.params)
        self._spill_k, self._spill_dirty = [0] * len(self.fp.params), False
        # a deferred 1 / world belongs to the gradients just cleared (a wait(defer_scale=True) whose step() never came, or a
        # step() that raised): it must not leak into the next step, which may run without a reduce
        self.grad_scale = 1.0
        for h in self.zero_grad_hooks:
            h()

    def mark_all_touched(self):
        """for callers that write gradients directly into the flat buffer (tests, custom reducers)"""
        self._touched = [True] * len(self.fp.params)
        self.finalize_grads()

    def scaled_grad(self, i=None):
        """The gradient the coming step() will apply -- ``grad * grad_scale`` -- of parameter ``i`` (None: the whole flat buffer),
        as a new tensor.  After a data-parallel reduce with a deferred scale (GradReducer.wait(defer_scale=True)) ``p.grad`` holds
        the SUM over ranks and ``grad_scale`` the 1 / world; a pre_step hook registered after the reducer's (gradient clipping,
        norm logging) reads the mean through this helper instead of the raw buffer."""
        self.finalize_grads()
        g = self.fp.grad if i is None else self.fp.grad_view(i)
        return g * self.grad_scale if self.grad_scale != 1.0 else g.clone()

    def step(self):
        try:
            streams.join_all(self.fp.grad.device, self.join_exclude)      # (see zero_grad)
            self._fold_spill()
            self.finalize_grads()
            for h in self.pre_step_hooks:
                h()
            self.fp.attach_grads()
            fp = self.fp
            i, n = 0, len(fp.params)
            while i < n:
                if not self._touched[i]:
                    i += 1
                    continue
                j, st = i, self.steps[i]
                while j + 1 < n and self._touched[j + 1] and self.steps[j + 1] == st:
                    j += 1
                lo, hi = fp.offsets[i], fp.offsets[j] + fp.params[j].numel()
                ops.adam_step(fp.flat[lo:hi], fp.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr,
                              self.betas[0], self.betas[1], self.eps, st + 1, self.grad_scale)
                for q in range(i, j + 1):
                    self.steps[q] = st + 1
                i = j + 1
        finally:
            self.grad_scale = 1.0          # also when a hook or a launch raised: the scale never outlives its step

    # ---- torch.optim.Adam-compatible (de)serialisation ----
    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.fp.params, self.fp.offsets)):
            if self.steps[i] == 0:
                continue
            n = p.numel()
            state[i] = {'step': torch.tensor(float(self.steps[i])),
                        'exp_avg': self.exp_avg[o:o + n].view(p.shape).clone(),
                        'exp_avg_sq': self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(self.fp.params))))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        self.lr, self.betas, self.eps = float(g['lr']), tuple(float(b) for b in g['betas']), float(g['eps'])
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.fp.params, self.fp.offsets)):
                st = sd['state'].get(i)
                n = p.numel()
                if st is None:
                    self.steps[i] = 0
                    self.exp_avg[o:o + n].zero_()
                    self.exp_avg_sq[o:o + n].zero_()
                    continue
                self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                self.steps[i] = int(float(st['step']))
