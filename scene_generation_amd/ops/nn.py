"""Element-wise and normalisation operators: activations, residual add, dropout mask, Instance / BatchNorm, pooling, padding.
(Part of scene_generation_amd.ops: see ops/__init__.py.)"""

import torch
from torch.autograd import Function

from . import _core
from ._core import (ACT_NONE, GradOut, _L, _call, _f32, _p, _stream, _wants_grad, workspace)


class ActFn(Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        x = _f32(x)
        y = torch.empty_like(x)
        _call('sg_act_fwd', _p(x), _p(y), x.numel(), act, slope, _stream())
        ctx.cfg = (act, slope)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        gy = _f32(gy)
        gx = torch.empty_like(gy)
        _call('sg_act_bwd', _p(y), _p(gy), _p(gx), gy.numel(), ctx.cfg[0], ctx.cfg[1], _stream())
        return gx, None, None


def activation(x, act, slope=0.0):
    return ActFn.apply(x, act, float(slope))


# =============================================================================================
# normalisation / pooling
# =============================================================================================

class AddFn(Function):
    """a + b of two equally shaped tensors (the shortcut of build_cnn's residual blocks, layers.py:116)"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a, 'add lhs'), _f32(b, 'add rhs')
        assert a.shape == b.shape, 'add: shapes %s and %s differ' % (tuple(a.shape), tuple(b.shape))
        out = torch.empty_like(a)
        _call('sg_add', _p(a), _p(b), _p(out), a.numel(), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return AddFn.apply(a, b)


class MaskMulFn(Function):
    """x * mask * alpha with a constant mask (nn.Dropout: mask ~ Bernoulli(1 - p), alpha = 1 / (1 - p))"""

    @staticmethod
    def forward(ctx, x, mask, alpha):
        x, mask = _f32(x, 'dropout input'), _f32(mask, 'dropout mask')
        out = torch.empty_like(x)
        _call('sg_mul', _p(x), _p(mask), float(alpha), _p(out), x.numel(), _stream())
        ctx.save_for_backward(mask)
        ctx.alpha = float(alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        mask, = ctx.saved_tensors
        g = _f32(g, 'dropout gradient')
        gx = torch.empty_like(g)
        _call('sg_mul', _p(g), _p(mask), ctx.alpha, _p(gx), g.numel(), _stream())
        return gx, None, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    if p >= 1.0:
        return MaskMulFn.apply(x, torch.zeros_like(x), 0.0)
    mask = torch.empty_like(x).bernoulli_(1.0 - p)          # the framework's device RNG: plumbing, like torch.empty
    return MaskMulFn.apply(x, mask, 1.0 / (1.0 - p))


class InstanceNormFn(Function):
    """act(InstanceNorm2d(x)) [+ skip]  (affine=False, eps 1e-5: layers.py:296)."""

    @staticmethod
    def forward(ctx, x, skip, eps, act, slope):
        x = _f32(x, 'instance-norm input')
        skip = None if skip is None else _f32(skip)
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(N * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _call('sg_instnorm_fwd', _p(x), _p(skip), _p(y), _p(mean), _p(rstd), N * C, H * W, eps, act, slope, _stream())
        ctx.cfg = (act, slope, skip is not None)
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mean, rstd = ctx.saved_tensors
        act, slope, has_skip = ctx.cfg
        gy = _f32(gy)
        N, C, H, W = x.shape
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _call('sg_instnorm_bwd', _p(x), _p(gy), _p(mean), _p(rstd), _p(gx), N * C, H * W, act, slope, _stream())
        return gx, (gy if has_skip and ctx.needs_input_grad[1] else None), None, None, None


def instance_norm(x, skip=None, eps=1e-5, act=ACT_NONE, slope=0.0):
    return InstanceNormFn.apply(x, skip, float(eps), act, float(slope))


class BatchNormFn(Function):
    """act(BatchNorm(x)) over (N, HW) per channel, training or eval mode (generators.py:22; layers.py:23-31)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, nbt, training, momentum, eps, act, slope):
        x = _f32(x, 'batch-norm input')
        shp = x.shape
        N, C = shp[0], shp[1]
        HW = x.numel() // (N * C) if x.numel() else 1
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        wsb = _L().sg_batchnorm_ws_bytes(N, C, HW)
        _call('sg_batchnorm_fwd', _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(rmean), _p(rvar), _p(nbt),
              N, C, HW, eps, momentum, 1 if training else 0, act, slope, _p(workspace(wsb, x.device)), wsb, _stream())
        ctx.cfg = (N, C, HW, act, slope, 1 if training else 0)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 11
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        N, C, HW, act, slope, training = ctx.cfg
        gy = _f32(gy)
        gx = torch.empty_like(x)
        # the kernel produces both affine gradients in one pass; skipped parameters get scratch outputs that are dropped
        want_g = gamma is not None and ctx.needs_input_grad[1] and _wants_grad(gamma)
        want_b = beta is not None and ctx.needs_input_grad[2] and _wants_grad(beta)
        og = GradOut(gamma) if want_g else None
        ob = GradOut(beta) if want_b else None
        gg = og.buf if want_g else (torch.empty(C, dtype=torch.float32, device=x.device) if gamma is not None else None)
        gb = ob.buf if want_b else (torch.empty(C, dtype=torch.float32, device=x.device) if beta is not None else None)
        wsb = _L().sg_batchnorm_ws_bytes(N, C, HW)
        _call('sg_batchnorm_bwd', _p(x), _p(gy), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(gx), _p(gg), _p(gb), N, C, HW,
              training, act, slope, _p(workspace(wsb, x.device)), wsb, _stream())
        return (gx, og.finish() if want_g else None, ob.finish() if want_b else None, None, None, None, None, None, None,
                None, None)


def batch_norm(x, gamma, beta, rmean, rvar, nbt, training, momentum=0.1, eps=1e-5, act=ACT_NONE, slope=0.0):
    return BatchNormFn.apply(x, gamma, beta, rmean, rvar, nbt, training, float(momentum), float(eps), act, float(slope))


class AvgPool3s2Fn(Function):
    """nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False) (discriminators.py:100,186)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        N, C, H, W = x.shape
        OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty(N, C, OH, OW, dtype=torch.float32, device=x.device)
        _call('sg_avgpool3s2_fwd', _p(x), _p(y), N * C, H, W, OH, OW, _stream())
        ctx.shape = (N, C, H, W, OH, OW)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W, OH, OW = ctx.shape
        gy = _f32(gy)
        gx = torch.empty(N, C, H, W, dtype=torch.float32, device=gy.device)
        _call('sg_avgpool3s2_bwd', _p(gy), _p(gx), N * C, H, W, OH, OW, _stream())
        return gx


def avgpool3s2(x):
    return AvgPool3s2Fn.apply(x)


class MaxPool2Fn(Function):
    """nn.MaxPool2d(2, 2) (VGG19 feature extractor of VGGLoss, losses.py:183-198)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        N, C, H, W = x.shape
        y = torch.empty(N, C, H // 2, W // 2, dtype=torch.float32, device=x.device)
        _call('sg_maxpool2_fwd', _p(x), _p(y), N * C, H, W, _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        gy = _f32(gy)
        N, C, H, W = x.shape
        gx = torch.empty_like(x)
        _call('sg_maxpool2_bwd', _p(x), _p(gy), _p(gx), N * C, H, W, _stream())
        return gx


def maxpool2(x):
    return MaxPool2Fn.apply(x)


class Pool2dFn(Function):
    """nn.MaxPool2d(k, k) / nn.AvgPool2d(k, k) for any window k (build_cnn 'P<k>', layers.py:181-189)."""

    @staticmethod
    def forward(ctx, x, k, avg):
        x = _f32(x)
        N, C, H, W = x.shape
        y = torch.empty(N, C, H // k, W // k, dtype=torch.float32, device=x.device)
        _call('sg_pool2d_fwd', _p(x), _p(y), N * C, H, W, k, 1 if avg else 0, _stream())
        ctx.k, ctx.avg, ctx.shape = k, avg, (N, C, H, W)
        ctx.save_for_backward(*(() if avg else (x,)))
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gy = _f32(gy)
        gx = torch.empty(N, C, H, W, dtype=torch.float32, device=gy.device)
        x = None if ctx.avg else ctx.saved_tensors[0]
        _call('sg_pool2d_bwd', _p(x) if x is not None else None, _p(gy), _p(gx), N * C, H, W, ctx.k, 1 if ctx.avg else 0, _stream())
        return gx, None, None


def pool2d(x, k, avg=False):
    return Pool2dFn.apply(x, int(k), bool(avg))


class ReplicatePadFn(Function):
    """nn.ReplicationPad2d(pad) (ResnetBlock padding_type='replicate', layers.py:245-246)."""

    @staticmethod
    def forward(ctx, x, pad):
        x = _f32(x)
        N, C, H, W = x.shape
        y = torch.empty(N, C, H + 2 * pad, W + 2 * pad, dtype=torch.float32, device=x.device)
        _call('sg_replicate_pad_fwd', _p(x), _p(y), N * C, H, W, pad, _stream())
        ctx.pad = pad
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = _f32(gy)
        p = ctx.pad
        N, C, PH, PW = gy.shape
        gx = torch.empty(N, C, PH - 2 * p, PW - 2 * p, dtype=torch.float32, device=gy.device)
        _call('sg_replicate_pad_bwd', _p(gy), _p(gx), N * C, PH - 2 * p, PW - 2 * p, p, _stream())
        return gx, None


def replicate_pad(x, pad):
    return ReplicatePadFn.apply(x, int(pad))


class GapFn(Function):
    """GlobalAvgPool (layers.py:82-85)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        y = torch.empty(N, C, dtype=torch.float32, device=x.device)
        _call('sg_gap_fwd', _p(x), _p(y), N * C, HW, _stream())
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        shp = ctx.shape
        gy = _f32(gy)
        gx = torch.empty(shp, dtype=torch.float32, device=gy.device)
        NC = shp[0] * shp[1]
        _call('sg_gap_bwd', _p(gy), _p(gx), NC, gx.numel() // NC, _stream())
        return gx


def global_avg_pool(x):
    return GapFn.apply(x)


class Upsample2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        N, C, H, W = x.shape
        y = torch.empty(N, C, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
        _call('sg_upsample2_fwd', _p(x), _p(y), N * C, H, W, _stream())
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = _f32(gy)
        N, C, H2, W2 = gy.shape
        gx = torch.empty(N, C, H2 // 2, W2 // 2, dtype=torch.float32, device=gy.device)
        _call('sg_pad_upsample_bwd', _p(gy), _p(gx), N * C, H2 // 2, W2 // 2, 0, 2, _stream())
        return gx


def upsample2(x):
    return Upsample2Fn.apply(x)


class ReflectPadFn(Function):
    @staticmethod
    def forward(ctx, x, pad):
        x = _f32(x)
        N, C, H, W = x.shape
        y = torch.empty(N, C, H + 2 * pad, W + 2 * pad, dtype=torch.float32, device=x.device)
        _call('sg_reflect_pad_fwd', _p(x), _p(y), N * C, H, W, pad, _stream())
        ctx.pad = pad
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = _f32(gy)
        p = ctx.pad
        N, C, PH, PW = gy.shape
        gx = torch.empty(N, C, PH - 2 * p, PW - 2 * p, dtype=torch.float32, device=gy.device)
        _call('sg_pad_upsample_bwd', _p(gy), _p(gx), N * C, PH - 2 * p, PW - 2 * p, p, 1, _stream())
        return gx, None


def reflect_pad(x, pad):
    return ReflectPadFn.apply(x, pad)
