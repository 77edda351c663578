"""Scalar losses, their one-launch weighted sum and cross-entropy (losses.py, trainer.py:331-340).
(Part of scene_generation_amd.ops: see ops/__init__.py.)"""
import ctypes

import torch
from torch.autograd import Function

from . import _core
from ._core import (LOSS_BCE_CONST, LOSS_BCE_PROB_CONST, LOSS_L1, LOSS_MEAN, LOSS_MSE, LOSS_MSE_CONST,
    LOSS_MSE_SIGMOID_CONST, WSUM_MAX, _L, _call, _f32, _i64, _p, _stream)


# =============================================================================================
# losses
# =============================================================================================

class ScalarLossFn(Function):
    """scale * sum_i l(a_i, b_i | target) as a 0-dim device tensor (no host sync)."""

    @staticmethod
    def forward(ctx, a, b, kind, target, scale):
        a = _f32(a, 'loss input')
        b = None if b is None else _f32(b, 'loss target')
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        wsb = _L().sg_loss_ws_bytes(a.numel())
        ws = torch.empty(wsb, dtype=torch.uint8, device=a.device)
        _call('sg_loss_fwd', kind, _p(a), _p(b), target, a.numel(), scale, _p(out), 0, _p(ws), wsb, _stream())
        ctx.cfg = (kind, target, scale)
        ctx.save_for_backward(a, b)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        kind, target, scale = ctx.cfg
        gout = _f32(gout.reshape(1))
        ga = torch.empty_like(a)
        _call('sg_loss_bwd', kind, _p(a), _p(b), target, a.numel(), scale, _p(gout), _p(ga), _stream())
        return ga, None, None, None, None


def mse_const(x, target):
    """nn.MSELoss()(x, full_like(x, target)) (losses.py:147-175)."""
    return ScalarLossFn.apply(x, None, LOSS_MSE_CONST, float(target), 1.0 / x.numel())


def mse(a, b):
    return ScalarLossFn.apply(a, b.detach(), LOSS_MSE, 0.0, 1.0 / a.numel())


def l1(a, b):
    return ScalarLossFn.apply(a, b.detach(), LOSS_L1, 0.0, 1.0 / a.numel())


def bce_logits_const(x, target):
    """bce_loss(x, full_like(x, target)) (losses.py:26-44)."""
    return ScalarLossFn.apply(x, None, LOSS_BCE_CONST, float(target), 1.0 / x.numel())


def mean(x):
    """x.mean() as a 0-dim device tensor (wgan losses, losses.py:93-112)"""
    return ScalarLossFn.apply(x, None, LOSS_MEAN, 0.0, 1.0 / x.numel())


def mse_sigmoid_const(x, target):
    """F.mse_loss(x.sigmoid(), full_like(x, target)) (lsgan losses, losses.py:115-132)"""
    return ScalarLossFn.apply(x, None, LOSS_MSE_SIGMOID_CONST, float(target), 1.0 / x.numel())


def bce_prob_const(x, target):
    """nn.BCELoss()(x, full_like(x, target)) on probabilities (GANLoss(use_lsgan=False), losses.py:147)"""
    return ScalarLossFn.apply(x, None, LOSS_BCE_PROB_CONST, float(target), 1.0 / x.numel())


class WeightedSumFn(Function):
    """sum_i w_i * t_i over 0-dim device tensors as ONE launch (and one for the backward): LossManager's running
    ``total_loss += loss * weight`` (utils.py:50-57) and the per-scale sums of GANLoss / calculate_features_loss."""

    @staticmethod
    def forward(ctx, weights, *terms):
        n = len(terms)
        terms = [_f32(t, 'loss term') for t in terms]
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
        w = (ctypes.c_float * n)(*weights)
        out = torch.empty(1, dtype=torch.float32, device=terms[0].device)
        _call('sg_weighted_sum_fwd', ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(w, ctypes.c_void_p), n, _p(out), _stream())
        ctx.w, ctx.n = w, n
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        gout = _f32(gout.reshape(1))
        g = torch.empty(ctx.n, dtype=torch.float32, device=gout.device)
        _call('sg_weighted_sum_bwd', ctypes.cast(ctx.w, ctypes.c_void_p), ctx.n, _p(gout), _p(g), _stream())
        return (None,) + tuple(g[i] if ctx.needs_input_grad[1 + i] else None for i in range(ctx.n))


def weighted_sum(tensors, weights):
    """sum_i weights[i] * tensors[i] for 0-dim device tensors (chunks of <= 32 terms per launch)"""
    tensors = [t.reshape(()) for t in tensors]
    weights = [float(w) for w in weights]
    while len(tensors) > WSUM_MAX:
        head = WeightedSumFn.apply(tuple(weights[:WSUM_MAX]), *tensors[:WSUM_MAX])
        tensors, weights = [head] + tensors[WSUM_MAX:], [1.0] + weights[WSUM_MAX:]
    return WeightedSumFn.apply(tuple(weights), *tensors)


class MultiLossFn(Function):
    """sum_t w_t * (scale_t * sum_i l(a_t[i], b_t[i] | target_t)) over several tensors in ONE launch (and one for the backward):
    the weighted sum of len(a) ScalarLossFn terms of one kind, same arithmetic in the same order (bit-identical), without the
    2 + 1 launches per term and the sum's own two (sg_multi_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, kind, targets, scales, weights, nt, *ab):
        a = [_f32(t, 'loss input') for t in ab[:nt]]
        b = [None if t is None else _f32(t, 'loss target') for t in ab[nt:]] if len(ab) > nt else [None] * nt
        dev = a[0].device
        pa = (ctypes.c_void_p * nt)(*[t.data_ptr() for t in a])
        pb = (ctypes.c_void_p * nt)(*[None if t is None else t.data_ptr() for t in b])
        n = (ctypes.c_int64 * nt)(*[t.numel() for t in a])
        sc, w, tg = (ctypes.c_float * nt)(*scales), (ctypes.c_float * nt)(*weights), (ctypes.c_float * nt)(*targets)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        wsb = _L().sg_multi_loss_ws_bytes(nt)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        cv = lambda x: ctypes.cast(x, ctypes.c_void_p)
        _call('sg_multi_loss_fwd', kind, nt, cv(pa), cv(pb), cv(n), cv(sc), cv(w), cv(tg), _p(out), None, _p(ws), wsb, _stream())
        ctx.cfg = (kind, nt, pa, pb, n, sc, w, tg)
        ctx.save_for_backward(*(a + [t for t in b if t is not None]))       # (keeps the operands the pointer arrays name alive)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        kind, nt, pa, pb, n, sc, w, tg = ctx.cfg
        a = ctx.saved_tensors[:nt]
        gout = _f32(gout.reshape(1))
        ga = [torch.empty_like(t) if ctx.needs_input_grad[5 + i] else None for i, t in enumerate(a)]
        pg = (ctypes.c_void_p * nt)(*[None if g is None else g.data_ptr() for g in ga])
        cv = lambda x: ctypes.cast(x, ctypes.c_void_p)
        _call('sg_multi_loss_bwd', kind, nt, cv(pa), cv(pb), cv(n), cv(sc), cv(w), cv(tg), _p(gout), cv(pg), _stream())
        return (None, None, None, None, None) + tuple(ga) + (None,) * (len(ctx.needs_input_grad) - 5 - nt)


def multi_loss(kind, a, b=None, targets=None, weights=None, mean=True):
    """sum_t weights[t] * l_kind(a[t], b[t] | targets[t]) with the mean over each tensor's elements (``mean``), as one launch;
    chunks of <= 32 terms.  ``b``: per-term second operands (detached) for the pair kinds, else None."""
    nt = len(a)
    weights = [1.0] * nt if weights is None else [float(x) for x in weights]
    targets = [0.0] * nt if targets is None else [float(x) for x in targets]
    if nt > WSUM_MAX:
        parts = [multi_loss(kind, a[i:i + WSUM_MAX], None if b is None else b[i:i + WSUM_MAX], targets[i:i + WSUM_MAX],
                            weights[i:i + WSUM_MAX], mean) for i in range(0, nt, WSUM_MAX)]
        return weighted_sum(parts, [1.0] * len(parts))
    scales = [1.0 / t.numel() if mean else 1.0 for t in a]
    extra = () if b is None else tuple(t.detach() for t in b)
    return MultiLossFn.apply(kind, tuple(targets), tuple(scales), tuple(weights), nt, *(tuple(a) + extra))


def l1_multi(a, b, weights):
    """sum_t weights[t] * nn.L1Loss()(a[t], b[t].detach())"""
    return multi_loss(LOSS_L1, a, b, None, weights)


class CrossEntropyFn(Function):
    @staticmethod
    def forward(ctx, logits, target):
        logits, target = _f32(logits), _i64(target)
        rows, classes = logits.shape
        row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        _call('sg_cross_entropy_fwd', _p(logits), _p(target), rows, classes, _p(row_loss), _p(out), _stream())
        ctx.save_for_backward(logits, target)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        logits, target = ctx.saved_tensors
        rows, classes = logits.shape
        gout = _f32(gout.reshape(1))
        gl = torch.empty_like(logits)
        _call('sg_cross_entropy_bwd', _p(logits), _p(target), rows, classes, _p(gout), _p(gl), _stream())
        return gl, None


def cross_entropy(logits, target):
    return CrossEntropyFn.apply(logits, target)
