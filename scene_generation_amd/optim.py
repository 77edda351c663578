"""Flat-buffer Adam for MI355X.

All parameters of one optimiser live in ONE contiguous fp32 buffer (parameters become views into it), and so do
their gradients and both Adam moments.  Consequences:
  * optimizer.step()      = one HIP launch over 28 B/param of HBM traffic (sg_adam_step)
  * optimizer.zero_grad() = one fill launch
  * data-parallel reduce  = a handful of large RCCL all-reduces over slices of the flat gradient buffer
    (scene_generation_amd.parallel) instead of one small collective per tensor.
``state_dict()`` / ``load_state_dict()`` speak torch.optim.Adam's schema (trainer.py:138,186), so reference
checkpoints round-trip.
"""
import torch

from . import ops


class FlatParams:
    """Re-homes ``params`` into one flat buffer (+ a flat gradient buffer).  Device-agnostic host logic."""

    def __init__(self, params):
        self.params = [p for p in params]
        assert self.params, 'no parameters'
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.numel = off
        self.flat = torch.empty(off, dtype=dt, device=dev)
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)
        self.attach_grads()

    def grad_view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def attach_grads(self):
        """(re)point every p.grad at its slice of the flat buffer; autograd then accumulates IN PLACE."""
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + self.offsets[i] * self.grad.element_size():
                p.grad = self.grad_view(i)


class FusedAdam:
    """torch.optim.Adam(params, lr, betas, eps, weight_decay=0, amsgrad=False) semantics on flat buffers."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.fp = FlatParams(params)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(self.fp.flat)
        self.exp_avg_sq = torch.zeros_like(self.fp.flat)
        self.step_count = 0
        self.pre_step_hooks = []          # e.g. GradReducer.wait

    @property
    def param_groups(self):
        return [dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False,
                     params=self.fp.params)]

    def zero_grad(self, set_to_none=False):
        ops.fill_(self.fp.grad, 0.0)
        self.fp.attach_grads()

    def step(self):
        for h in self.pre_step_hooks:
            h()
        self.fp.attach_grads()
        self.step_count += 1
        ops.adam_step(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.step_count)

    # ---- torch.optim.Adam-compatible (de)serialisation ----
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.fp.params, self.fp.offsets)):
                n = p.numel()
                state[i] = {'step': torch.tensor(float(self.step_count)),
                            'exp_avg': self.exp_avg[o:o + n].view(p.shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False, maximize=False,
                     foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(self.fp.params))))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        self.lr, self.betas, self.eps = float(g['lr']), tuple(float(b) for b in g['betas']), float(g['eps'])
        steps = set()
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.fp.params, self.fp.offsets)):
                st = sd['state'].get(i)
                if st is None:
                    continue
                n = p.numel()
                self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                steps.add(int(float(st['step'])))
        assert len(steps) <= 1, 'per-parameter step counts differ'
        self.step_count = steps.pop() if steps else 0
