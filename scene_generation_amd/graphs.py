"""hipGraph replay of static-shape sub-networks.

The step is bound by the host as much as by the GPU: every kernel is issued through Python (autograd Function -> ctypes),
~35 us per launch.  Sub-networks whose shapes never change from iteration to iteration -- everything of the generator
behind its first convolution (N x 64 x H x W in, image out: ~110 launches forward, ~280 backward) and the VGG19 feature
extractor of the perceptual loss -- are therefore captured ONCE into a pair of hipGraphs (forward, backward) and replayed
with one launch each.  The capture runs the very same Python code (the ctypes launches land on the capturing stream), so a
replay executes exactly the kernels, in exactly the order, of the eager path: results are bit-identical.

Parameter gradients inside a captured backward go straight into the optimiser's flat gradient buffer (ops.GradOut); the
host-side notifications (touched flags, data-parallel reducer) that the eager path issues per parameter are recorded at
capture time and re-issued after every replay.  Object-level sub-networks (mask_net, encoders, object / mask
discriminators) see a different number of objects every iteration and stay eager.
"""
import os
import weakref

import torch
from torch.autograd import Function

from . import ops

ENABLED = os.environ.get('SG_GRAPHS', '1') != '0'
REPLAYS = [0]                    # hipGraph launches issued by this process
# Only the capturing thread is policed: a training process has other threads that legitimately touch the runtime while a
# segment is being captured (the DataLoader's pin-memory thread, RCCL's watchdog), and under the default 'global' mode any of
# their calls would invalidate the capture.
CAPTURE_MODE = os.environ.get('SG_GRAPH_CAPTURE_MODE', 'thread_local')
# debugging: synchronise after every replay (and name the segment on stderr), so that a fault inside a replayed graph is reported
# at the replay and not at the next host synchronisation
SYNC_REPLAYS = os.environ.get('SG_GRAPH_SYNC', '0') == '1'


def _after_replay(entry, what):
    if SYNC_REPLAYS:
        import sys
        print('scene_generation_amd.graphs: replayed %s of %s' % (what, getattr(entry, 'name', '?')), file=sys.stderr, flush=True)
        torch.cuda.synchronize()


def _flat(out):
    if isinstance(out, torch.Tensor):
        return [out], None
    out = list(out)
    assert all(isinstance(t, torch.Tensor) for t in out), 'a graphed segment returns a tensor or a flat list of tensors'
    return out, len(out)


class _Entry(object):
    pass


class _Token(object):
    """lives as long as the autograd node of one grad-mode replay: an entry whose token is alive has a backward pending"""
    __slots__ = ('__weakref__',)


class _GraphedFn(Function):
    @staticmethod
    def forward(ctx, entry, x):
        entry.static_in.copy_(x)
        entry.fwd.replay()
        REPLAYS[0] += 1
        _after_replay(entry, 'forward')
        ctx.entry = entry
        # the activations this replay saved live in the graph's private pool until the matching backward has run (or the
        # autograd node is dropped): GraphedSegment.__call__ refuses a second grad-mode replay of the entry until then
        ctx.token = _Token()
        entry.pending = weakref.ref(ctx.token)
        outs = [o.detach() for o in entry.static_out]
        if entry.clone_outputs:
            outs = [o.clone() for o in outs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        e = ctx.entry
        for buf, g in zip(e.static_gout, gys):
            if g is None:
                ops.fill_(buf, 0.0)
            else:
                buf.copy_(g)
        # A captured backward OVERWRITES the gradient slices of its parameters (sink mode 0: every parameter used once per
        # step, after zero_grad()).  If a slice already holds a contribution (gradient accumulation: a second backward
        # without zero_grad()), keep it and add it back after the replay -- the eager path would have accumulated.
        carried = []
        if not e.accumulate:
            for opt, i in e.deliveries:
                if opt._touched[i]:
                    view = opt.fp.grad_view(i)
                    carried.append((view, view.clone()))
        e.bwd.replay()
        REPLAYS[0] += 1
        _after_replay(e, 'backward')
        for view, old in carried:
            view.add_(old)
        for opt, i in e.deliveries:           # what ops.GradOut.finish() tells the optimiser on the eager path
            opt._on_grad(i)
        e.pending = None
        ctx.token = None
        return None, (e.static_gin.detach() if e.static_gin is not None else None)


class GraphedSegment(object):
    """``fn(x) -> tensor | list of tensors``: a pure function of one tensor and of parameters that are updated in place.

    ``accumulate``: parameter gradients of this segment may be contributions to slices that already hold one (a
    discriminator run several times per step): the captured backward adds; otherwise (every parameter used once per step,
    after zero_grad()) it overwrites.  ``clone_outputs``: hand out copies instead of views of the static output buffers
    (for outputs that callers keep across iterations)."""

    def __init__(self, fn, params=(), accumulate=False, warmup=2, clone_outputs=False, name='segment', modules=()):
        self.fn, self.params = fn, list(params)
        self.accumulate, self.warmup, self.clone_outputs, self.name = accumulate, warmup, clone_outputs, name
        self.modules = list(modules)           # their train / eval flags are part of the graph key (BatchNorm, dropout)
        self.entries, self.seen = {}, {}

    def __deepcopy__(self, memo):
        import copy
        return GraphedSegment(copy.deepcopy(self.fn, memo), copy.deepcopy(self.params, memo), self.accumulate, self.warmup,
                              self.clone_outputs, self.name, copy.deepcopy(self.modules, memo))   # the copy re-captures

    def __call__(self, x):
        if not (ENABLED and x.is_cuda) or ops.prof_is_enabled():
            return self.fn(x)
        grad = torch.is_grad_enabled()
        need_grad = grad and (x.requires_grad or any(p.requires_grad for p in self.params))
        key = (tuple(x.shape), x.dtype, need_grad, grad and x.requires_grad, tuple(p.requires_grad for p in self.params),
               ops.WINOGRAD, ops.FACTORED_LAYOUT, ops.skip_state_key(), tuple(m.training for m in self.modules))
        e = self.entries.get(key)
        if e is None:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if n < self.warmup:                # eager first: builds the shape tables and sizes the workspaces
                return self.fn(x)
            if need_grad and any(p.requires_grad and ops._sink_of(p) is None for p in self.params):
                e = self.entries[key] = False  # parameters outside a FusedAdam: their gradients are autograd's business
            else:
                try:
                    e = self._capture(x, need_grad)
                except Exception as exc:       # a failed capture must never take the training step down with it
                    import sys
                    print('scene_generation_amd.graphs: capture of %s failed (%r); staying eager' % (self.name, exc),
                          file=sys.stderr)
                    e = False
                self.entries[key] = e
        if e is False:
            return self.fn(x)
        if need_grad and e.pending is not None and e.pending() is not None:
            # a second grad-mode call before the first one's backward: a replay would overwrite the activations that
            # backward still needs (they live in the graph's pool) -- this call runs eager
            return self.fn(x)
        if need_grad:
            outs = _GraphedFn.apply(e, x)
        else:
            e.static_in.copy_(x)
            e.fwd.replay()
            REPLAYS[0] += 1
            _after_replay(e, 'forward (no grad)')
            outs = [o.detach() for o in e.static_out]
            if e.clone_outputs:
                outs = [o.clone() for o in outs]
        return outs[0] if e.n_out is None else list(outs)

    def _capture(self, x, need_grad):
        e = _Entry()
        e.name = self.name
        e.clone_outputs = self.clone_outputs
        e.accumulate = self.accumulate
        e.pending = None
        x_grad = torch.is_grad_enabled() and x.requires_grad
        e.static_in = x.detach().clone().requires_grad_(x_grad)
        torch.cuda.synchronize()
        e.fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(e.fwd, capture_error_mode=CAPTURE_MODE):
            out = self.fn(e.static_in)
        e.static_out, e.n_out = _flat(out)
        e.static_gin, e.static_gout, e.deliveries, e.bwd = None, [], [], None
        if need_grad:
            e.static_gout = [torch.zeros_like(o) for o in e.static_out]
            inputs = ([e.static_in] if x_grad else []) + [p for p in self.params if p.requires_grad]
            torch.cuda.synchronize()
            e.bwd = torch.cuda.CUDAGraph()
            with ops.capture_deliveries(1 if self.accumulate else 0) as deliveries:
                with torch.cuda.graph(e.bwd, pool=e.fwd.pool(), capture_error_mode=CAPTURE_MODE):
                    grads = torch.autograd.grad(e.static_out, inputs, e.static_gout, allow_unused=True, retain_graph=True)
            e.deliveries = list(deliveries)
            e.static_gin = grads[0] if x_grad else None
            late = [g for g in grads[(1 if x_grad else 0):] if g is not None]
            assert not late, ("%s: %d parameter gradients were returned as tensors instead of being written to an optimiser's "
                              "flat buffer -- graphed segments need FusedAdam-owned parameters" % (self.name, len(late)))
        return e
