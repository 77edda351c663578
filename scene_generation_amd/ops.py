"""Autograd operators over the C ABI of libsg2im_hip.so (include/sg2im_hip.h).

PyTorch-ROCm is used here as plumbing only: device allocation (torch.empty), the autograd tape and
the current HIP stream.  Every forward/backward below is one or a few hand-written gfx950 kernel
launches through ctypes; there is no eager/CPU fallback -- a CPU tensor raises.
"""
import contextlib
import os
import ctypes

import torch
from torch.autograd import Function

from . import _hip
from ._hip import sgConvDesc

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
LOSS_MSE_CONST, LOSS_MSE, LOSS_L1, LOSS_BCE_CONST, LOSS_MEAN, LOSS_MSE_SIGMOID_CONST, LOSS_BCE_PROB_CONST = range(7)
WSUM_MAX = 32

_ws_cache = {}


def _L():
    return _hip.lib()


# raw handle of the current stream without building a torch.cuda.Stream object (this runs once per kernel launch: the
# handle and the tensor addresses are passed to ctypes as plain ints, no c_void_p objects)
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_dev = torch.cuda.current_device


def _stream_handle():
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


_stream = _stream_handle


def _p(t):
    return None if t is None else t.data_ptr()


def _dev(t, name='tensor'):
    if not t.is_cuda:
        raise RuntimeError('scene_generation_amd: %s is on %s -- the MI355X HIP path has no CPU fallback'
                           % (name, t.device))
    return t


def _f32(t, name='tensor'):
    _dev(t, name)
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32, got %s' % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _i64(t, name='index'):
    _dev(t, name)
    if t.dtype != torch.int64:
        raise TypeError('%s must be int64, got %s' % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def workspace(nbytes, device):
    """Per-device scratch, grown on demand.  Safe to share: every consumer is ordered on the current stream."""
    key = (device.index, _stream_handle())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_fn_cache = {}
CALLS = [0]                       # C-ABI calls issued by this process (bench.py reports calls per step)


def _call(name, *args):
    CALLS[0] += 1
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(_L(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError('%s failed (rc=%d): %s' % (name, rc, _hip.last_error()))


# =============================================================================================
# parameter-gradient sinks
# =============================================================================================
# A parameter owned by a FusedAdam lives in a flat buffer and so does its gradient (optim.FlatParams).  Returning a
# gradient tensor from backward makes autograd ADD it into that slice with one ATen launch per parameter per backward
# (~300 launches and 3 passes over the 765 MB generator gradient per step).  Instead the weight-gradient kernels write
# straight into the slice: the first contribution after zero_grad() overwrites (the slice is zero), the k-th goes to the same
# slice of the optimiser's spill buffer k-2 (folded in by optimizer.step(); inside a hipGraph capture: temporary + sg_axpy).
# backward then returns None for that input and notifies the optimiser (touched flag, DP reducer).
_SINKS = {}


class ParamSink(object):
    __slots__ = ('opt', 'i', 'view', 'ptr')

    def __init__(self, opt, i):
        import weakref
        self.opt, self.i = weakref.ref(opt), i
        self.view = opt.fp.grad_view(i)
        self.ptr = opt.fp.params[i].data_ptr()


def register_param_sinks(opt):
    for i, p in enumerate(opt.fp.params):
        _SINKS[p.data_ptr()] = ParamSink(opt, i)


def _sink_of(param):
    if not _SINKS or param is None:
        return None
    sk = _SINKS.get(param.data_ptr())
    if sk is None:
        return None
    opt = sk.opt()
    if opt is None or opt.fp.params[sk.i].data_ptr() != sk.ptr or tuple(sk.view.shape) != tuple(param.shape):
        del _SINKS[param.data_ptr()]           # the optimiser is gone (its flat buffer may have been re-used)
        return None
    return sk


_CAPTURE = None          # while a backward is being captured into a hipGraph: [forced sink mode, [(optimiser, index), ...]]


@contextlib.contextmanager
def capture_deliveries(mode):
    """Backward capture of a graphed segment (graphs.py): the overwrite-vs-add decision of the sinks is fixed to ``mode``
    (it would otherwise be taken from the optimiser's state at capture time and then frozen into the graph), and the
    notifications to the optimisers are collected instead of issued (the graph replays re-issue them)."""
    global _CAPTURE
    assert _CAPTURE is None
    _CAPTURE = [mode, []]
    try:
        yield _CAPTURE[1]
    finally:
        _CAPTURE = None


class GradOut(object):
    """where the gradient of ``param`` goes: ``buf`` is what the kernel writes; ``finish()`` is what backward returns"""
    __slots__ = ('sink', 'buf', 'mode')

    def __init__(self, param):
        sk = self.sink = _sink_of(param)
        if sk is None:
            self.buf, self.mode = torch.empty_like(param), 2
        elif (_CAPTURE[0] == 0) if _CAPTURE is not None else (not sk.opt()._touched[sk.i]):
            self.buf, self.mode = sk.view, 0               # first contribution since zero_grad(): write in place
        elif _CAPTURE is None and hasattr(sk.opt(), 'spill_view'):
            # k-th contribution (k >= 2: a discriminator's real / wrong-texture pass): written in place into the optimiser's
            # spill buffer k-2, which optimizer.step() folds into the gradient with ONE launch (optim.FusedAdam._fold_spill)
            self.buf, self.mode = sk.opt().spill_view(sk.i), 3
        else:
            self.buf, self.mode = torch.empty_like(param), 1

    def finish(self):
        if self.mode == 2:
            return self.buf
        sk = self.sink
        if self.mode in (1, 3):
            for f in getattr(sk.opt(), 'late_listeners', ()):      # e.g. GradReducer.late_contribution: may refuse
                f(sk.i)
        if self.mode == 1:
            _call('sg_axpy', _p(sk.view), _p(self.buf), 1.0, self.buf.numel(), _stream())
        if _CAPTURE is not None:
            _CAPTURE[1].append((sk.opt(), sk.i))
        else:
            sk.opt()._on_grad(sk.i)
        return None


# =============================================================================================
# convolution family
# =============================================================================================

def conv_out_size(size, k, stride, pad, upsample=1):
    return (size * upsample + 2 * pad - k) // stride + 1


_desc_cache = {}


def _conv_desc(N, C1, C2, H, W, Cout, KS, stride, pad, reflect, upsample, OH, OW, out_pad=0, x2_broadcast=0):
    """sgConvDesc for these sizes (one instance per shape: the ctypes reference and every shape-only query of the library
    -- workspace sizes, which specialised kernels apply -- are memoised on it; a training step repeats ~150 shapes)."""
    key = (N, C1, C2, H, W, Cout, KS, stride, pad, 1 if reflect else 0, upsample, OH, OW, out_pad, x2_broadcast)
    d = _desc_cache.get(key)
    if d is None:
        d = sgConvDesc(*key)
        d._ref = ctypes.byref(d)
        d._memo = {}
        _desc_cache[key] = d
    return d


def _q(d, name, *extra):
    """memoised shape-only query ``name(desc, *extra)`` of the library"""
    key = (name,) + extra
    v = d._memo.get(key)
    if v is None:
        v = d._memo[key] = getattr(_L(), name)(d._ref, *extra)
    return v


class Conv2dFn(Function):
    """act(conv2d([x1 ‖ x2]) + bias) with reflection padding / nearest-x2 upsampling / channel concat folded
    into the implicit-GEMM gather (nn.Conv2d call sites: generators.py:20-27,68-89; layers.py:160-180,251-270;
    discriminators.py:137-158,215-234)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, stride, pad, reflect, upsample, act, slope, grad_from, sparse=None):
        x1 = _f32(x1, 'conv input')
        x2 = None if x2 is None else _f32(x2, 'conv input 2')
        weight = _f32(weight, 'conv weight')
        N, C1, H, W = x1.shape
        C2 = 0 if x2 is None else x2.size(1)
        Cout, Cin, KS, KS2 = weight.shape
        assert KS == KS2 and Cin == C1 + C2, 'conv weight %s does not match input channels %d' % (tuple(weight.shape), C1 + C2)
        OH, OW = conv_out_size(H, KS, stride, pad, upsample), conv_out_size(W, KS, stride, pad, upsample)
        bcast = 1 if (x2 is not None and x2.dim() == 2) else 0      # [N, C2] broadcast over H x W
        d = _conv_desc(N, C1, C2, H, W, Cout, KS, stride, pad, reflect, upsample, OH, OW, 0, bcast)
        y = torch.empty(N, Cout, OH, OW, dtype=torch.float32, device=x1.device)
        ctx.smallm = x2 is None and sparse is None and bool(_q(d, 'sg_conv2d_smallm_supported'))
        ctx.wino = (x2 is None and sparse is None and WINOGRAD and bool(_q(d, 'sg_conv2d_wino_supported')))
        ctx.head = (x2 is None and sparse is None and HEADCONV and not ctx.wino and not ctx.smallm
                    and bool(_q(d, 'sg_conv2d_head_supported')))
        ctx.wino24 = (x2 is None and sparse is None and WINOGRAD24 and not ctx.head and not ctx.smallm
                      and bool(_q(d, 'sg_conv2d_wino24_supported')))
        if ctx.wino24:              # stride-1 4x4 convs of the PatchGANs: Winograd F(2x2,4x4), 25 batched dense GEMMs
            wsb = _q(d, 'sg_conv2d_wino24_ws_bytes')
            _call('sg_conv2d_wino24_fwd', d._ref, _p(x1), _p(weight), _p(bias), _p(y), act, slope,
                  _p(workspace(wsb, x1.device)), wsb, _stream())
        elif ctx.head:                # one output channel (PatchGAN score maps, mask_net's 1x1 head): vector-ALU reduction
            wsb = _q(d, 'sg_conv2d_head_ws_bytes')
            _call('sg_conv2d_head_fwd', d._ref, _p(x1), _p(weight), _p(bias), _p(y), act, slope,
                  _p(workspace(wsb, x1.device)), wsb, _stream())
        elif ctx.wino:              # ResnetBlock convs: Winograd F(2x2,3x3), 16 batched dense GEMMs
            wsb = _q(d, 'sg_conv2d_wino_ws_bytes')
            # the data gradient of the same conv multiplies with the transposed filter transform: build it now, in the same
            # pass over the weights (they do not change between this forward and its backward)
            utn = _q(d, 'sg_conv2d_wino_ut_floats') if ctx.needs_input_grad[0] else 0
            ctx.wino_ut = torch.empty(utn, dtype=torch.float32, device=x1.device) if utn else None
            _call('sg_conv2d_wino_fwd', d._ref, _p(x1), _p(weight), _p(bias), _p(y), act, slope, _p(ctx.wino_ut),
                  _p(workspace(wsb, x1.device)), wsb, _stream())
        elif ctx.smallm:              # <= 4 output channels (the RGB head): direct vector-ALU kernel, no MFMA tile waste
            _call('sg_conv2d_smallm_fwd', d._ref, _p(x1), _p(weight), _p(bias), _p(y), act, slope, _stream())
        elif sparse is not None:    # (chan_list [N, L] int32, chan_cnt [N] int32): see sg_conv2d_fwd_sparse
            clist, ccnt = sparse
            assert clist.dtype == torch.int32 and ccnt.dtype == torch.int32 and clist.size(0) == N == ccnt.numel()
            L = int(clist.size(1))
            wsb = _q(d, 'sg_conv2d_sparse_ws_bytes', L, 0)
            ws = workspace(wsb, x1.device)
            _call('sg_conv2d_fwd_sparse', d._ref, _p(x1), _p(x2), _p(weight), _p(bias), _p(clist), _p(ccnt), L,
                  _p(y), act, slope, _p(ws), wsb, _stream())
        else:
            wsb = _q(d, 'sg_conv2d_ws_bytes', 0)
            ws = workspace(wsb, x1.device)
            _call('sg_conv2d_fwd', d._ref, _p(x1), _p(x2), _p(weight), _p(bias), _p(y), act, slope, _p(ws), wsb,
                  _stream())
        ctx.desc = d
        ctx.sparse = sparse
        ctx.bias_ref = bias          # only its identity is used (gradient sink / skip list), never its values
        ctx.set_materialize_grads(False)
        ctx.cfg = (act, slope, bias is not None, int(grad_from))
        ctx.save_for_backward(x1, x2, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 12
        x1, x2, weight, y = ctx.saved_tensors
        d = ctx.desc
        act, slope, has_bias, grad_from = ctx.cfg
        gy = _f32(gy)
        s = _stream()
        if act != ACT_NONE:
            g2 = torch.empty_like(gy)
            _call('sg_act_bwd', _p(y), _p(gy), _p(g2), gy.numel(), act, slope, s)
            gy = g2
        need_x1, need_x2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and x2 is not None
        need_w = ctx.needs_input_grad[2] and _wants_grad(weight)
        need_b = has_bias and ctx.needs_input_grad[3] and _wants_grad(ctx.bias_ref)
        gx1 = gx2 = gw = gb = None
        dev = gy.device
        if need_x1 or need_x2:
            wsb = _q(d, 'sg_conv2d_ws_bytes', 1)
            ws = workspace(wsb, dev)
            fold = d.pad_reflect or d.upsample == 2
            GH = d.H * d.upsample + (2 * d.pad if d.pad_reflect else 0)
            GW = d.W * d.upsample + (2 * d.pad if d.pad_reflect else 0)

            folded = x2 is None and _q(d, 'sg_conv2d_dgrad_folded_supported')

            def dgrad(c0, c1):
                if ctx.head and c0 == 0 and c1 == d.C1:
                    out = torch.empty(d.N, d.C1, d.H, d.W, dtype=torch.float32, device=dev)
                    _call('sg_conv2d_head_dgrad', d._ref, _p(gy), _p(weight), _p(out), s)
                    return out
                if ctx.wino24 and c0 == 0 and c1 == d.C1:   # Winograd F(2x2,4x4) on gy with the rotated, transposed filter
                    out = torch.empty(d.N, d.C1, d.H, d.W, dtype=torch.float32, device=dev)
                    fb = _q(d, 'sg_conv2d_wino24_ws_bytes')
                    _call('sg_conv2d_wino24_dgrad', d._ref, _p(gy), _p(weight), _p(out), _p(workspace(fb, dev)), fb, s)
                    return out
                if ctx.wino and c0 == 0 and c1 == d.C1:     # Winograd on the padded gradient grid + reflection fold
                    out = torch.empty(d.N, d.C1, d.H, d.W, dtype=torch.float32, device=dev)
                    fb = _q(d, 'sg_conv2d_wino_ws_bytes')
                    _call('sg_conv2d_wino_dgrad', d._ref, _p(gy), _p(weight), _p(out), _p(getattr(ctx, 'wino_ut', None)),
                          _p(workspace(fb, dev)), fb, s)
                    return out
                if folded:       # ReflectionPad(1)+3x3: gradient straight on the H x W grid (no padded grid, no fold pass)
                    out = torch.empty(d.N, c1 - c0, d.H, d.W, dtype=torch.float32, device=dev)
                    fb = _q(d, 'sg_conv2d_dgrad_folded_ws_bytes')
                    _call('sg_conv2d_dgrad_folded', d._ref, _p(gy), _p(weight), _p(out), c0, c1,
                          _p(workspace(fb, dev)), fb, s)
                    return out
                g = torch.empty(d.N, c1 - c0, GH, GW, dtype=torch.float32, device=dev)
                _call('sg_conv2d_dgrad', d._ref, _p(gy), _p(weight), _p(g), c0, c1, _p(ws), wsb, s)
                if fold:
                    out = torch.empty(d.N, c1 - c0, d.H, d.W, dtype=torch.float32, device=dev)
                    _call('sg_pad_upsample_bwd', _p(g), _p(out), d.N * (c1 - c0), d.H, d.W,
                          d.pad if d.pad_reflect else 0, d.upsample, s)
                    return out
                return g
            if need_x1:
                if grad_from > 0:      # channels [0, grad_from) of x1 are constants of the graph (one-hot layout block)
                    gx1 = torch.zeros(d.N, d.C1, d.H, d.W, dtype=torch.float32, device=dev)
                    gx1[:, grad_from:] = dgrad(grad_from, d.C1)
                else:
                    gx1 = dgrad(0, d.C1)
            if need_x2:
                gx2 = dgrad(d.C1, d.C1 + d.C2)
                if d.x2_broadcast:          # [N, C2] source broadcast over H x W: reduce the map gradient
                    red = torch.empty(d.N, d.C2, dtype=torch.float32, device=dev)
                    _call('sg_gap_fwd', _p(gx2), _p(red), d.N * d.C2, d.H * d.W, s)
                    gx2 = scale_(red, float(d.H * d.W))
        if need_w or need_b:
            ow = GradOut(weight) if need_w else None
            ob = GradOut(ctx.bias_ref) if need_b else None
            if need_w:
                gw = ow.buf
                gb = ob.buf if need_b else None
                if ctx.wino24:
                    wsb = max(_q(d, 'sg_conv2d_wino24_ws_bytes'), _L().sg_channel_sum_ws_bytes(d.Cout))
                    ws = workspace(wsb, dev)
                    _call('sg_conv2d_wino24_wgrad', d._ref, _p(gy), _p(x1), _p(gw), _p(ws), wsb, s)
                    if gb is not None:
                        _call('sg_channel_sum', _p(gy), _p(gb), d.N, d.Cout, d.OH * d.OW, _p(ws), wsb, s)
                elif ctx.head:
                    wsb = max(_q(d, 'sg_conv2d_head_ws_bytes'), _L().sg_channel_sum_ws_bytes(d.Cout))
                    ws = workspace(wsb, dev)
                    _call('sg_conv2d_head_wgrad', d._ref, _p(gy), _p(x1), _p(gw), _p(ws), wsb, s)
                    if gb is not None:
                        _call('sg_channel_sum', _p(gy), _p(gb), d.N, d.Cout, d.OH * d.OW, _p(ws), wsb, s)
                elif ctx.wino:
                    wsb = max(_q(d, 'sg_conv2d_wino_ws_bytes'), _L().sg_channel_sum_ws_bytes(d.Cout))
                    ws = workspace(wsb, dev)
                    _call('sg_conv2d_wino_wgrad', d._ref, _p(gy), _p(x1), _p(gw), _p(ws), wsb, s)
                    if gb is not None:
                        _call('sg_channel_sum', _p(gy), _p(gb), d.N, d.Cout, d.OH * d.OW, _p(ws), wsb, s)
                elif ctx.smallm:
                    wsb = max(_q(d, 'sg_conv2d_smallm_ws_bytes'), _L().sg_channel_sum_ws_bytes(d.Cout))
                    ws = workspace(wsb, dev)
                    _call('sg_conv2d_smallm_wgrad', d._ref, _p(gy), _p(x1), _p(gw), _p(ws), wsb, s)
                    if gb is not None:
                        _call('sg_channel_sum', _p(gy), _p(gb), d.N, d.Cout, d.OH * d.OW, _p(ws), wsb, s)
                elif ctx.sparse is not None:
                    clist, ccnt = ctx.sparse
                    L = int(clist.size(1))
                    wsb = _q(d, 'sg_conv2d_sparse_ws_bytes', L, 2)
                    ws = workspace(wsb, dev)
                    _call('sg_conv2d_wgrad_sparse', d._ref, _p(gy), _p(x1), _p(x2), _p(clist), _p(ccnt), L,
                          _p(gw), _p(gb), _p(ws), wsb, s)
                else:
                    wsb = _q(d, 'sg_conv2d_ws_bytes', 2)
                    ws = workspace(wsb, dev)
                    _call('sg_conv2d_wgrad', d._ref, _p(gy), _p(x1), _p(x2), _p(gw), _p(gb), _p(ws), wsb, s)
            else:
                gb = ob.buf
                wsb = _L().sg_channel_sum_ws_bytes(d.Cout)
                ws = workspace(wsb, dev)
                _call('sg_channel_sum', _p(gy), _p(gb), d.N, d.Cout, d.OH * d.OW, _p(ws), wsb, s)
            gw = ow.finish() if need_w else None
            gb = ob.finish() if need_b else None
        return gx1, gx2, gw, gb, None, None, None, None, None, None, None, None


# single-output-channel convs on the vector ALUs (SG_HEADCONV=0: the 32x128 MFMA tile with one live row)
HEADCONV = os.environ.get('SG_HEADCONV', '1') != '0'
# Winograd F(2x2,3x3) for the ResnetBlock convs (SG_WINOGRAD=0 keeps them on the direct implicit-GEMM kernels)
WINOGRAD = os.environ.get('SG_WINOGRAD', '1') != '0'
# Winograd F(2x2,4x4) for the stride-1 4x4 convs of the PatchGANs (SG_WINOGRAD24=0 keeps them on the direct kernels)
WINOGRAD24 = os.environ.get('SG_WINOGRAD24', '1') != '0'
# convs over a masks_to_layout() layout computed from its factored form (SG_FACTORED_LAYOUT=0: channel-sparse path instead)
FACTORED_LAYOUT = os.environ.get('SG_FACTORED_LAYOUT', '1') != '0'

_SKIP_PARAM_GRADS = set()


@contextlib.contextmanager
def skip_param_grads(params):
    """While active, backward passes do not compute (nor accumulate) gradients of ``params`` even though they require
    grad: lets a discriminator forward recorded during the generator step be re-used by the discriminator step
    (the reference re-runs the identical forward, trainer.py:302-325) without paying for weight gradients twice."""
    ids = {p.data_ptr() for p in params}
    _SKIP_PARAM_GRADS.update(ids)
    try:
        yield
    finally:
        _SKIP_PARAM_GRADS.difference_update(ids)


def _wants_grad(t):
    return t is not None and t.data_ptr() not in _SKIP_PARAM_GRADS


def skip_state_key():
    """hashable summary of the skip set (part of the key of captured graphs: it changes what a backward computes)"""
    return len(_SKIP_PARAM_GRADS)


# ---- layout hints -------------------------------------------------------------------------------------
# What the model knows about a masks_to_layout() result -- which channels can be non-zero per image ('sparse',
# 'sparse_cat'), that the one-hot block is a constant of the graph ('grad_from'), its factored form ('factored'), whether
# the dense tensor has been written yet ('pending') -- lives in a side table keyed by the tensor's storage address, NOT in
# Python attributes: the reference's training loop passes ``layout.detach()`` around (train.py:208-215) and a plain
# ``.detach()`` keeps the storage but drops attributes.  An entry holds a strong reference to its tensor, so the address
# cannot be recycled while the entry exists; Model.forward clears the table at the start of every iteration.
_HINT_TABLE = {}
_HINT_KEYS = ('sparse', 'sparse_cat', 'grad_from', 'factored', 'keep_grad', 'pending')


def clear_hints():
    _HINT_TABLE.clear()


def set_hints(t, **kw):
    e = _HINT_TABLE.get(t.data_ptr())
    if e is None or tuple(e[0].shape) != tuple(t.shape):
        e = _HINT_TABLE[t.data_ptr()] = (t, {})
    e[1].update(kw)
    return t


def hints_of(t):
    if not _HINT_TABLE:
        return None
    e = _HINT_TABLE.get(t.data_ptr())
    if e is None or tuple(e[0].shape) != tuple(t.shape) or e[0].stride() != t.stride():
        return None
    return e[1]


def hint(t, key, default=None):
    h = hints_of(t)
    return default if h is None else h.get(key, default)


def carry_hints(src, dst, grad_from=False):
    """copy the layout hints (per-image active channels, constant channel block) to a tensor derived from ``src`` by an
    op that keeps all-zero channels all-zero (average pooling)"""
    h = hints_of(src)
    if h:
        set_hints(dst, **{k: v for k, v in h.items() if k != 'pending' and (k != 'grad_from' or grad_from)})
    return dst


def detach_keep(t):
    """``t.detach()``: the hints follow the storage (kept for callers of the round-1 API)"""
    return t.detach()


def ensure_dense(t):
    """run the deferred masks_to_layout launch of a lazily built layout (Model.lazy_layouts) before a dense read"""
    h = hints_of(t)
    if h and h.get('pending') is not None:
        fill = h.pop('pending')
        fill()
    return t


def _upconv_prefers_winograd(x, weight):
    """the folded-upsample Winograd path (>= 128 channels in multiples of 128) keeps its convs"""
    if not WINOGRAD:
        return False
    N, C, H, W = x.shape
    d = _conv_desc(N, C, 0, H, W, weight.size(0), 3, 1, 1, False, 2, 2 * H, 2 * W, 0, 0)
    return bool(_q(d, 'sg_conv2d_wino_supported'))


def conv2d(x, weight, bias=None, stride=1, pad=0, reflect=False, upsample=1, act=ACT_NONE, slope=0.0, x2=None):
    """Layout hints (see the table above): 'grad_from' = c promises that nobody needs d/dx[:, :c] (the data gradient is
    then only computed for channels >= c, the rest is returned as zeros); 'sparse' = (chan_list, chan_cnt) promises that,
    per image, every channel outside the list is all-zero (forward and weight gradient then only visit the listed
    channels, sg_conv2d_*_sparse); 'factored' = the layout as planes x per-object vectors (factored_layout_conv)."""
    h = hints_of(x)
    if (UPCONV and upsample == 2 and stride == 1 and pad == 1 and not reflect and x2 is None and h is None
            and act == ACT_NONE and weight.size(2) == 3 and weight.size(3) == 3 and not _upconv_prefers_winograd(x, weight)):
        return UpConv3Fn.apply(x, weight, bias)
    if h is None:
        return Conv2dFn.apply(x, x2, weight, bias, stride, pad, reflect, upsample, act, float(slope), 0, None)
    f = h.get('factored') if FACTORED_LAYOUT else None
    if f is not None and upsample == 1 and (x2 is None or (x2.dim() == 4 and not reflect)):
        if not (x.requires_grad or h.get('keep_grad')):     # a detached layout: no gradient reaches the appearance
            f = f.detached()                                # vectors through its factored form either
        return factored_layout_conv(f, weight, bias, stride, pad, reflect, act, slope, x2)
    if h.get('pending') is not None and x.requires_grad:
        raise NotImplementedError('a lazily built layout (Model.lazy_layouts) only carries gradients through its factored '
                                  'form; this convolution needs the dense tensor -- build the model output densely')
    ensure_dense(x)
    grad_from = int(h.get('grad_from', 0)) if x.requires_grad else 0
    if not (0 < grad_from < x.size(1)):
        grad_from = 0
    if x2 is None:
        sparse = h.get('sparse')
    else:          # channel-concatenated second source (the image next to the layout): lists that include its channels
        sparse = h.get('sparse_cat', {}).get(x2.size(1)) if x2.dim() == 4 else None
    if sparse is not None and not (2 * sparse[0].size(1) <= x.size(1)):
        sparse = None                      # not sparse enough to pay for the per-image weight compaction
    return Conv2dFn.apply(x, x2, weight, bias, stride, pad, reflect, upsample, act, float(slope), grad_from, sparse)


class ConvTranspose2dFn(Function):
    """nn.ConvTranspose2d(k3, s2, p1, output_padding=1) of the generator's up path (generators.py:83-87)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_pad):
        x = _f32(x, 'convT input')
        weight = _f32(weight, 'convT weight')
        N, Cin, H, W = x.shape
        Cin2, Cout, KS, _ = weight.shape
        assert Cin == Cin2
        OH = (H - 1) * stride - 2 * pad + KS + out_pad
        OW = (W - 1) * stride - 2 * pad + KS + out_pad
        d = _conv_desc(N, Cin, 0, H, W, Cout, KS, stride, pad, False, 1, OH, OW, out_pad)
        y = torch.empty(N, Cout, OH, OW, dtype=torch.float32, device=x.device)
        wsb = _q(d, 'sg_conv2d_ws_bytes', 0)
        ws = workspace(wsb, x.device)
        _call('sg_convT2d_fwd', d._ref, _p(x), _p(weight), _p(bias), _p(y), _p(ws), wsb, _stream())
        ctx.desc = d
        ctx.bias_ref = bias
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 6
        x, weight = ctx.saved_tensors
        d = ctx.desc
        gy = _f32(gy)
        s = _stream()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            wsb = _q(d, 'sg_conv2d_ws_bytes', 1)
            ws = workspace(wsb, gy.device)
            _call('sg_convT2d_dgrad', d._ref, _p(gy), _p(weight), _p(gx), _p(ws), wsb, s)
        need_w = ctx.needs_input_grad[1] and _wants_grad(weight)
        need_b = ctx.bias_ref is not None and ctx.needs_input_grad[2] and _wants_grad(ctx.bias_ref)
        ow = GradOut(weight) if need_w else None
        ob = GradOut(ctx.bias_ref) if need_b else None
        if need_w:
            wsb = _q(d, 'sg_conv2d_ws_bytes', 2)
            ws = workspace(wsb, gy.device)
            _call('sg_convT2d_wgrad', d._ref, _p(gy), _p(x), _p(ow.buf), _p(ob.buf) if need_b else None, _p(ws), wsb, s)
        elif need_b:
            wsb = _L().sg_channel_sum_ws_bytes(d.Cout)
            ws = workspace(wsb, gy.device)
            _call('sg_channel_sum', _p(gy), _p(ob.buf), d.N, d.Cout, d.OH * d.OW, _p(ws), wsb, s)
        gw = ow.finish() if need_w else None
        gb = ob.finish() if need_b else None
        return gx, gw, gb, None, None, None


def conv_transpose2d(x, weight, bias=None, stride=2, pad=1, out_pad=1):
    return ConvTranspose2dFn.apply(x, weight, bias, stride, pad, out_pad)


# Interpolate(x2, nearest) + Conv2d(3, padding=1) as a sub-pixel transposed convolution (SG_UPCONV=0: 3x3 gather over the
# folded upsample instead): 16 instead of 36 multiply-adds per input pixel and channel pair (mask_net, generators.py:20-21)
UPCONV = os.environ.get('SG_UPCONV', '1') != '0'


class UpConv3Fn(Function):
    """conv3x3(pad 1)(nearest_up2(x)) == convT(k4, s2, p1)(x; wt), wt = the 3x3 taps summed per source pixel
    (sg_upconv3_fold_weights); backward = the transposed conv's data / weight gradients + the adjoint of the fold."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _f32(x, 'conv input')
        weight = _f32(weight, 'conv weight')
        N, Cin, H, W = x.shape
        Cout = weight.size(0)
        assert tuple(weight.shape) == (Cout, Cin, 3, 3)
        s = _stream()
        d = _conv_desc(N, Cin, 0, H, W, Cout, 4, 2, 1, False, 1, 2 * H, 2 * W, 0)
        wt = torch.empty(Cin, Cout, 4, 4, dtype=torch.float32, device=x.device)
        _call('sg_upconv3_fold_weights', _p(weight), _p(wt), Cout, Cin, s)
        y = torch.empty(N, Cout, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
        wsb = _q(d, 'sg_conv2d_ws_bytes', 0)
        _call('sg_convT2d_fwd', d._ref, _p(x), _p(wt), _p(bias), _p(y), _p(workspace(wsb, x.device)), wsb, s)
        ctx.desc = d
        ctx.bias_ref = bias
        ctx.weight_ref = weight
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, wt)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return None, None, None
        x, wt = ctx.saved_tensors
        d, weight = ctx.desc, ctx.weight_ref
        gy = _f32(gy)
        s, dev = _stream(), gy.device
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            wsb = _q(d, 'sg_conv2d_ws_bytes', 1)
            _call('sg_convT2d_dgrad', d._ref, _p(gy), _p(wt), _p(gx), _p(workspace(wsb, dev)), wsb, s)
        need_w = ctx.needs_input_grad[1] and _wants_grad(weight)
        need_b = ctx.bias_ref is not None and ctx.needs_input_grad[2] and _wants_grad(ctx.bias_ref)
        ow = GradOut(weight) if need_w else None
        ob = GradOut(ctx.bias_ref) if need_b else None
        if need_w:
            gwt = torch.empty_like(wt)
            wsb = _q(d, 'sg_conv2d_ws_bytes', 2)
            _call('sg_convT2d_wgrad', d._ref, _p(gy), _p(x), _p(gwt), _p(ob.buf) if need_b else None,
                  _p(workspace(wsb, dev)), wsb, s)
            _call('sg_upconv3_unfold_wgrad', _p(gwt), _p(ow.buf), d.Cout, d.C1, s)
        elif need_b:
            wsb = _L().sg_channel_sum_ws_bytes(d.Cout)
            _call('sg_channel_sum', _p(gy), _p(ob.buf), d.N, d.Cout, d.OH * d.OW, _p(workspace(wsb, dev)), wsb, s)
        gw = ow.finish() if need_w else None
        gb = ob.finish() if need_b else None
        return gx, gw, gb


# =============================================================================================
# dense layers
# =============================================================================================

class LinearFn(Function):
    """act(x W^T + b): nn.Linear (+ the ReLU build_mlp appends, layers.py:215-231)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        x = _f32(x, 'linear input')
        weight = _f32(weight, 'linear weight')
        rows, in_f = x.shape
        out_f = weight.size(0)
        assert weight.size(1) == in_f
        y = torch.empty(rows, out_f, dtype=torch.float32, device=x.device)
        if rows > 0:
            _call('sg_linear_fwd', _p(x), _p(weight), _p(bias), _p(y), rows, in_f, out_f, act, slope, _stream())
        ctx.cfg = (act, slope, bias is not None)
        ctx.bias_ref = bias
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 5
        x, weight, y = ctx.saved_tensors
        act, slope, has_bias = ctx.cfg
        gy = _f32(gy)
        rows, in_f = x.shape
        out_f = weight.size(0)
        s = _stream()
        if rows == 0:
            return (torch.zeros_like(x), torch.zeros_like(weight),
                    torch.zeros(out_f, device=x.device) if has_bias else None, None, None)
        if act != ACT_NONE:
            g2 = torch.empty_like(gy)
            _call('sg_act_bwd', _p(y), _p(gy), _p(g2), gy.numel(), act, slope, s)
            gy = g2
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _call('sg_linear_bwd_data', _p(gy), _p(weight), _p(gx), rows, in_f, out_f, s)
        need_w = ctx.needs_input_grad[1] and _wants_grad(weight)
        need_b = has_bias and ctx.needs_input_grad[2] and _wants_grad(ctx.bias_ref)
        ow = GradOut(weight) if need_w else None
        ob = GradOut(ctx.bias_ref) if need_b else None
        if need_w:
            _call('sg_linear_bwd_weight', _p(gy), _p(x), _p(ow.buf), _p(ob.buf) if need_b else None, rows, in_f, out_f, s)
        elif need_b:
            _call('sg_channel_sum', _p(gy), _p(ob.buf), rows, out_f, 1, None, 0, s)
        gw = ow.finish() if need_w else None
        gb = ob.finish() if need_b else None
        return gx, gw, gb, None, None


def linear(x, weight, bias=None, act=ACT_NONE, slope=0.0):
    return LinearFn.apply(x, weight, bias, act, float(slope))


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


# =============================================================================================
# graph convolution
# =============================================================================================
_csr_cache = {}


def build_csr(edges, O):
    """Destination-major CSR of the (pass, t) entries (device-side; cached per edges tensor version)."""
    edges = _i64(edges, 'edges')
    key = (edges.data_ptr(), edges._version, edges.size(0), O)
    hit = _csr_cache.get('k')
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    T = edges.size(0)
    off = torch.empty(O + 1, dtype=torch.int32, device=edges.device)
    ent = torch.empty(max(2 * T, 1), dtype=torch.int32, device=edges.device)
    _call('sg_build_csr', _p(edges), T, O, _p(off), _p(ent), _stream())
    _csr_cache['k'] = (key, off, ent, edges)      # keep edges alive so the data_ptr key stays unique
    return off, ent


class GatherConcatFn(Function):
    """cur_t = [obj[s], pred, obj[o]] (graph.py:79-84); backward = deterministic segmented sums."""

    @staticmethod
    def forward(ctx, obj, pred, edges, off, ent):
        obj, pred = _f32(obj), _f32(pred)
        T, Do, Dp = edges.size(0), obj.size(1), pred.size(1)
        out = torch.empty(T, 2 * Do + Dp, dtype=torch.float32, device=obj.device)
        _call('sg_gather_concat_fwd', _p(obj), _p(pred), _p(edges), _p(out), T, Do, Dp, _stream())
        ctx.dims = (obj.size(0), T, Do, Dp)
        ctx.save_for_backward(off, ent)
        return out

    @staticmethod
    def backward(ctx, g):
        off, ent = ctx.saved_tensors
        O, T, Do, Dp = ctx.dims
        g = _f32(g)
        s = _stream()
        g_obj = g_pred = None
        if ctx.needs_input_grad[0]:
            g_obj = torch.empty(O, Do, dtype=torch.float32, device=g.device)
            _call('sg_segment_sum', _p(g), 2 * Do + Dp, 0, Do + Dp, Do, _p(off), _p(ent), _p(g_obj), O, 0, s)
        if ctx.needs_input_grad[1]:
            g_pred = torch.empty(T, Dp, dtype=torch.float32, device=g.device)
            _call('sg_copy_cols', _p(g), 2 * Do + Dp, Do, _p(g_pred), Dp, 0, T, Dp, s)
        return g_obj, g_pred, None, None, None


class TriplePoolFn(Function):
    """new_t -> (pooled object vectors, new predicate vectors): the split + scatter_add + avg of graph.py:89-116,
    accumulated in the reference's CPU order (bit-exact)."""

    @staticmethod
    def forward(ctx, new_t, edges, off, ent, O, H, Dout, avg):
        new_t = _f32(new_t)
        T = new_t.size(0)
        ld = 2 * H + Dout
        pooled = torch.empty(O, H, dtype=torch.float32, device=new_t.device)
        new_p = torch.empty(T, Dout, dtype=torch.float32, device=new_t.device)
        s = _stream()
        _call('sg_segment_sum', _p(new_t), ld, 0, H + Dout, H, _p(off), _p(ent), _p(pooled), O, 1 if avg else 0, s)
        _call('sg_copy_cols', _p(new_t), ld, H, _p(new_p), Dout, 0, T, Dout, s)
        ctx.dims = (T, O, H, Dout, avg)
        ctx.save_for_backward(edges, off)
        return pooled, new_p

    @staticmethod
    def backward(ctx, g_pooled, g_new_p):
        edges, off = ctx.saved_tensors
        T, O, H, Dout, avg = ctx.dims
        dev = edges.device
        g_pooled = _f32(g_pooled) if g_pooled is not None else torch.zeros(O, H, device=dev)
        g_new_p = None if g_new_p is None else _f32(g_new_p)
        g = torch.empty(T, 2 * H + Dout, dtype=torch.float32, device=dev)
        _call('sg_pool_bwd', _p(g_pooled), _p(g_new_p), _p(edges), _p(off), _p(g), T, H, Dout, 1 if avg else 0, _stream())
        return g, None, None, None, None, None, None, None


class EmbeddingFn(Function):
    @staticmethod
    def forward(ctx, table, idx):
        table, idx = _f32(table), _i64(idx)
        out = torch.empty(idx.numel(), table.size(1), dtype=torch.float32, device=table.device)
        _call('sg_embedding_fwd', _p(table), _p(idx), _p(out), idx.numel(), table.size(1), _stream())
        ctx.rows = table.size(0)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(idx, table)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, table = ctx.saved_tensors
        if g is None or not (ctx.needs_input_grad[0] and _wants_grad(table)):
            return None, None
        g = _f32(g)
        ot = GradOut(table)
        _call('sg_embedding_bwd', _p(g), _p(idx), _p(ot.buf), idx.numel(), ctx.rows, g.size(1), _stream())
        return ot.finish(), None


def embedding(table, idx):
    return EmbeddingFn.apply(table, idx)


class ConcatColsFn(Function):
    """torch.cat(tensors, dim=1) for 2-D fp32 tensors (model.py:134,152,168,171)."""

    @staticmethod
    def forward(ctx, *ts):
        ts = [_f32(t) for t in ts]
        rows = ts[0].size(0)
        widths = [t.size(1) for t in ts]
        out = torch.empty(rows, sum(widths), dtype=torch.float32, device=ts[0].device)
        s, off = _stream(), 0
        for t, w in zip(ts, widths):
            _call('sg_copy_cols', _p(t), w, 0, _p(out), out.size(1), off, rows, w, s)
            off += w
        ctx.widths = widths
        return out

    @staticmethod
    def backward(ctx, g):
        g = _f32(g)
        rows, ld = g.shape
        outs, off, s = [], 0, _stream()
        for i, w in enumerate(ctx.widths):
            if ctx.needs_input_grad[i]:
                gi = torch.empty(rows, w, dtype=torch.float32, device=g.device)
                _call('sg_copy_cols', _p(g), ld, off, _p(gi), w, 0, rows, w, s)
                outs.append(gi)
            else:
                outs.append(None)
            off += w
        return tuple(outs)


def concat_cols(*ts):
    return ConcatColsFn.apply(*ts)


def one_hot(idx, classes, dtype=torch.float32):
    idx = _i64(idx)
    out = torch.empty(idx.numel(), classes, dtype=torch.float32, device=idx.device)
    _call('sg_one_hot', _p(idx), _p(out), idx.numel(), classes, classes, 0, _stream())
    return out


class ConcatChannelsFn(Function):
    """materialised torch.cat((a, b), dim=1) on NCHW (API completeness; the training path folds it into conv2d)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32(a), _f32(b)
        N, Ca, H, W = a.shape
        Cb = b.size(1)
        out = torch.empty(N, Ca + Cb, H, W, dtype=torch.float32, device=a.device)
        _call('sg_concat_channels', _p(a), _p(b), _p(out), N, Ca, Cb, H * W, _stream())
        ctx.ca = Ca
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.ca].contiguous(), g[:, ctx.ca:].contiguous()


# =============================================================================================
# layout + crops + vector pool
# =============================================================================================

def segment_offsets(obj_to_img, N):
    obj_to_img = _i64(obj_to_img, 'obj_to_img')
    off = torch.empty(N + 1, dtype=torch.int32, device=obj_to_img.device)
    _call('sg_segment_offsets', _p(obj_to_img), obj_to_img.numel(), N, _p(off), _stream())
    return off


class MasksToLayoutFn(Function):
    """masks_to_layout (layout.py:64-93) train branch; gradients w.r.t. vecs (columns >= grad_from), float masks, boxes."""

    @staticmethod
    def forward(ctx, vecs, boxes, masks, seg_off, N, H, W, avg, grad_from, max_per_image, obj_to_img=None):
        vecs, boxes = _f32(vecs, 'vecs'), _f32(boxes, 'boxes')
        _dev(masks, 'masks')
        if masks.dtype == torch.int64:
            i64 = 1
        elif masks.dtype == torch.float32:
            i64 = 0
        else:
            raise TypeError('masks must be int64 or float32')
        masks = masks if masks.is_contiguous() else masks.contiguous()
        O, D = vecs.shape
        M = masks.size(1)
        out = torch.empty(N, D, H, W, dtype=torch.float32, device=vecs.device)
        _call('sg_masks_to_layout_fwd', _p(vecs), _p(boxes), _p(masks), i64, _p(seg_off), _p(out), N, O, D, M, H, W,
              1 if avg else 0, max_per_image, _stream())
        ctx.cfg = (N, O, D, M, H, W, avg, grad_from, i64)
        geom = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        ctx.save_for_backward(boxes, masks, seg_off, vecs if geom else None, obj_to_img if geom else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        boxes, masks, seg_off, vecs, obj_to_img = ctx.saved_tensors
        N, O, D, M, H, W, avg, grad_from, i64 = ctx.cfg
        gv = gb = gm = None
        gout = _f32(gout)
        if ctx.needs_input_grad[0]:
            gv = torch.empty(O, D, dtype=torch.float32, device=gout.device)
            _call('sg_masks_to_layout_bwd_vecs', _p(gout), _p(boxes), _p(masks), i64, None, _p(seg_off), _p(gv), N, O, D, M,
                  H, W, 1 if avg else 0, grad_from, _stream())
        want_b, want_m = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if want_b or want_m:
            if obj_to_img is None:
                raise RuntimeError('masks_to_layout: gradients w.r.t. boxes / masks need obj_to_img (pass it to the Function)')
            if want_m and i64:
                raise RuntimeError('masks_to_layout: integer masks have no gradient')
            gm = torch.empty(O, M, M, dtype=torch.float32, device=gout.device) if want_m else None
            gb = torch.empty(O, 4, dtype=torch.float32, device=gout.device) if want_b else None
            wsb = _L().sg_masks_to_layout_bwd_geom_ws_bytes(O, H, W)
            _call('sg_masks_to_layout_bwd_geom', _p(gout), _p(vecs), _p(boxes), _p(masks), i64, _p(_i64(obj_to_img)), _p(seg_off),
                  _p(gm), _p(gb), _p(workspace(wsb, gout.device)), wsb, N, O, D, M, H, W, 1 if avg else 0, _stream())
        return gv, gb, gm, None, None, None, None, None, None, None, None


def masks_to_layout_deferred(vecs, boxes, masks, seg_off, N, H, W, avg, max_per_image, differentiable=False):
    """An (N, D, H, W) layout whose sg_masks_to_layout_fwd launch is DEFERRED until somebody reads it densely
    (ensure_dense): with the factored layout convs nothing on the training step does -- the three dense 428 MB layouts of
    model.py:119-121 are outputs for logging only (train.py:201-203,219).  No autograd history: gradients reach the
    appearance vectors through the factored form.  ``differentiable`` marks the result as requiring grad (like the
    reference's gt_layout) so that consumers can tell it from its ``.detach()``."""
    vecs, boxes = _f32(vecs.detach(), 'vecs'), _f32(boxes.detach(), 'boxes')
    masks = _dev(masks.detach(), 'masks')
    if masks.dtype not in (torch.int64, torch.float32):
        raise TypeError('masks must be int64 or float32')
    masks = masks if masks.is_contiguous() else masks.contiguous()
    O, D = vecs.shape
    out = torch.empty(N, D, H, W, dtype=torch.float32, device=vecs.device)

    def fill():
        _call('sg_masks_to_layout_fwd', _p(vecs), _p(boxes), _p(masks), 1 if masks.dtype == torch.int64 else 0, _p(seg_off),
              _p(out), N, O, D, masks.size(1), H, W, 1 if avg else 0, max_per_image, _stream())
    if differentiable:
        out.requires_grad_(True)
    return set_hints(out, pending=fill)


def masks_to_layout_test(vecs, boxes, masks, seg_off, N, H, W, avg):
    """test-mode compositing (layout.py:87-92,157-169); inference only, returns a tensor without history."""
    if torch.is_grad_enabled() and (vecs.requires_grad or masks.requires_grad or boxes.requires_grad):
        raise NotImplementedError('masks_to_layout(test_mode=True) is inference-only (the reference composites through '
                                  '.item()/numpy argsort, layout.py:161-162): run it under torch.no_grad()')
    vecs, boxes = _f32(vecs.detach(), 'vecs'), _f32(boxes.detach(), 'boxes')
    masks = _dev(masks.detach(), 'masks')
    if masks.dtype not in (torch.int64, torch.float32):
        raise TypeError('masks must be int64 or float32')
    masks = masks if masks.is_contiguous() else masks.contiguous()
    O, D = vecs.shape
    out = torch.empty(N, D, H, W, dtype=torch.float32, device=vecs.device)
    wsb = _L().sg_masks_to_layout_test_ws_bytes(O)
    ws = workspace(wsb, vecs.device)
    _call('sg_masks_to_layout_test_fwd', _p(vecs), _p(boxes), _p(masks), 1 if masks.dtype == torch.int64 else 0, _p(seg_off),
          _p(out), _p(ws), wsb, N, O, D, masks.size(1), H, W, 1 if avg else 0, _stream())
    return out


# ------------------------------------------------------------------------------------------
# factored layout convolutions
# ------------------------------------------------------------------------------------------
def layout_planes(boxes, masks, seg_off, plane_idx, N, J, H, W):
    """Z [N, J, H, W]: the sampled mask S_o of the j-th object of every image (the spatial factor of masks_to_layout) =
    masks_to_layout with the vectors one_hot(plane index of o): the same fused kernel, D = J channels."""
    with torch.no_grad():
        sel = one_hot(plane_idx, J)
        return MasksToLayoutFn.apply(sel, boxes.detach(), masks.detach(), seg_off, N, H, W, False, 0, J, None)


class FactoredLayout(object):
    """layout = sum_o vecs[o] (x) S_o with vecs[o] = [one_hot(class_o) | repr_o] (model.py:165-168, layout.py:85-86), kept in
    factored form: Z [N, J, H, W] (planes S_o per image), the class ids, the appearance vectors and where object o sits
    (image, plane).  A conv over the layout is then a conv over <= J planes with per-image weights
    W_eff[o] = W[:, class_o] + sum_d repr[o, d] W[:, num_objs + d] -- 204 -> <= 9 "channels"."""

    def __init__(self, Z, objs, repr_vecs, num_objs, img_idx, plane_idx, counts_host, seg=None):
        self.Z, self.objs, self.repr, self.num_objs = Z, objs, repr_vecs, int(num_objs)
        self.img_idx, self.plane_idx = img_idx, plane_idx          # int64 [O] on the device
        self.counts_host = list(counts_host)                       # objects per image (host ints)
        if seg is None:                                            # int32 [N + 1] object offsets per image
            off = [0]
            for c in self.counts_host:
                off.append(off[-1] + c)
            from .utils import to_device_async
            seg = to_device_async(torch.tensor(off, dtype=torch.int32), Z.device)
        self.seg = seg
        self._lists = {}

    def detached(self):
        if not self.repr.requires_grad:
            return self
        f = FactoredLayout(self.Z, self.objs, self.repr.detach(), self.num_objs, self.img_idx, self.plane_idx,
                           self.counts_host, self.seg)
        f._lists = self._lists
        return f

    def with_planes(self, Z):
        f = FactoredLayout(Z, self.objs, self.repr, self.num_objs, self.img_idx, self.plane_idx, self.counts_host, self.seg)
        f._lists = self._lists
        return f

    def lists(self, extra):
        """(chan_list [N, L], chan_cnt [N], extra_pos [N, extra]) for the gather: the image's planes, then ``extra``
        channels of a concatenated second source (they sit at channel ids J.. and list positions cnt[n]..)"""
        key = int(extra)
        if key not in self._lists:
            import numpy as np
            N, J = self.Z.size(0), self.Z.size(1)
            L = max(self.counts_host) + key
            cl = np.zeros((N, L), dtype=np.int32)
            cc = np.zeros((N,), dtype=np.int32)
            ep = np.zeros((N, max(key, 1)), dtype=np.int64)
            for n, c in enumerate(self.counts_host):
                ch = list(range(c)) + [J + i for i in range(key)]
                cl[n, :len(ch)] = ch
                cl[n, len(ch):] = ch[0]
                cc[n] = len(ch)
                ep[n, :key] = [c + i for i in range(key)]
            dev = self.Z.device
            from .utils import to_device_async
            self._lists[key] = (to_device_async(torch.from_numpy(cl), dev), to_device_async(torch.from_numpy(cc), dev),
                                to_device_async(torch.from_numpy(ep), dev), L)
        return self._lists[key]


class PerImageConvFn(Function):
    """conv over planes with per-image weights wimg [N, Cout, L, KS, KS] (sg_conv2d_fwd_perimage / _wgrad_perimage).
    ``planes`` = the layout's mask planes, followed by the channels of a concatenated second source ``x2`` when there
    is one (the image next to the layout in the image discriminator).  The mask planes are constants; ``x2`` gets its data
    gradient from the shared weights ``w_full`` [Cout, cfull + C2, KS, KS] (it does not depend on the per-image part)."""

    @staticmethod
    def forward(ctx, planes, x2, wimg, bias, clist, ccnt, w_full, cfull, stride, pad, reflect, act, slope):
        planes = _f32(planes, 'planes')
        wimg = _f32(wimg, 'per-image weights')
        N, J, H, W = planes.shape
        Cout, L, KS = wimg.size(1), wimg.size(2), wimg.size(3)
        OH, OW = conv_out_size(H, KS, stride, pad, 1), conv_out_size(W, KS, stride, pad, 1)
        d = _conv_desc(N, J, 0, H, W, Cout, KS, stride, pad, reflect, 1, OH, OW, 0, 0)
        y = torch.empty(N, Cout, OH, OW, dtype=torch.float32, device=planes.device)
        wsb = _q(d, 'sg_conv2d_sparse_ws_bytes', L, 0)
        _call('sg_conv2d_fwd_perimage', d._ref, _p(planes), None, _p(wimg), _p(bias), _p(clist), _p(ccnt), L, _p(y),
              act, slope, _p(workspace(wsb, planes.device)), wsb, _stream())
        ctx.desc, ctx.L = d, L
        ctx.bias_ref = bias
        ctx.set_materialize_grads(False)
        ctx.cfg = (act, slope, bias is not None, int(cfull), 0 if x2 is None else x2.size(1), stride, pad, reflect)
        ctx.save_for_backward(planes, clist, ccnt, w_full, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 13
        planes, clist, ccnt, w_full, y = ctx.saved_tensors
        d, L = ctx.desc, ctx.L
        act, slope, has_bias, cfull, C2, stride, pad, reflect = ctx.cfg
        gy = _f32(gy)
        s, dev = _stream(), gy.device
        if act != ACT_NONE:
            g2 = torch.empty_like(gy)
            _call('sg_act_bwd', _p(y), _p(gy), _p(g2), gy.numel(), act, slope, s)
            gy = g2
        gx2 = gwimg = gb = None
        # skip_param_grads (the discriminators inside the generator step): the per-image weights only carry parameter
        # gradients when the appearance vectors are detached, which is the case for every discriminator input
        want_w = _wants_grad(w_full)
        if ctx.needs_input_grad[2] and want_w:
            gwimg = torch.empty(d.N, d.Cout, L, d.KS, d.KS, dtype=torch.float32, device=dev)
            wsb = _q(d, 'sg_conv2d_sparse_ws_bytes', L, 2)
            _call('sg_conv2d_wgrad_perimage', d._ref, _p(gy), _p(planes), None, _p(clist), _p(ccnt), L, _p(gwimg),
                  _p(workspace(wsb, dev)), wsb, s)
        if has_bias and ctx.needs_input_grad[3] and _wants_grad(ctx.bias_ref):
            ob = GradOut(ctx.bias_ref)
            wsb = _L().sg_channel_sum_ws_bytes(d.Cout)
            _call('sg_channel_sum', _p(gy), _p(ob.buf), d.N, d.Cout, d.OH * d.OW, _p(workspace(wsb, dev)), wsb, s)
            gb = ob.finish()
        if C2 and ctx.needs_input_grad[1]:
            # same conv seen with its full channel layout [layout channels | x2]: only the x2 slice is differentiated
            if reflect:
                raise NotImplementedError('factored layout conv: x2 gradient with reflection padding')
            df = _conv_desc(d.N, cfull, C2, d.H, d.W, d.Cout, d.KS, stride, pad, reflect, 1, d.OH, d.OW, 0, 0)
            wsb = _q(df, 'sg_conv2d_ws_bytes', 1)
            gx2 = torch.empty(d.N, C2, d.H, d.W, dtype=torch.float32, device=dev)
            _call('sg_conv2d_dgrad', df._ref, _p(gy), _p(_f32(w_full)), _p(gx2), cfull, cfull + C2,
                  _p(workspace(wsb, dev)), wsb, s)
        return None, gx2, gwimg, gb, None, None, None, None, None, None, None, None, None


class FactoredWeightsFn(Function):
    """per-image filters of a factored layout conv and their gradients w.r.t. the conv weight and the appearance
    vectors (sg_factored_weights_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, weight, repr_vecs, objs, seg, img_idx, N, L, C, C2):
        weight, repr_vecs = _f32(weight, 'conv weight'), _f32(repr_vecs, 'appearance vectors')
        M, Ct, KS, _ = weight.shape
        O, R = repr_vecs.shape
        assert Ct == C + R + C2 and seg.dtype == torch.int32 and seg.numel() == N + 1
        wimg = torch.empty(N, M, L, KS, KS, dtype=torch.float32, device=weight.device)
        _call('sg_factored_weights_fwd', _p(weight), _p(repr_vecs), _p(_i64(objs)), _p(seg), _p(wimg), N, O, M, L, KS * KS, C, R,
              C2, _stream())
        ctx.cfg = (N, O, M, L, KS * KS, C, R, C2)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(weight, repr_vecs, objs, seg, img_idx)
        return wimg

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 9
        weight, repr_vecs, objs, seg, img_idx = ctx.saved_tensors
        N, O, M, L, KS2, C, R, C2 = ctx.cfg
        need_w = ctx.needs_input_grad[0] and _wants_grad(weight)
        need_r = ctx.needs_input_grad[1]
        if not (need_w or need_r):
            return (None,) * 9
        g = _f32(g)
        ow = GradOut(weight) if need_w else None
        grepr = torch.empty_like(repr_vecs) if need_r else None
        _call('sg_factored_weights_bwd', _p(g), _p(weight), _p(repr_vecs), _p(objs), _p(seg), _p(_i64(img_idx)),
              _p(ow.buf) if need_w else None, _p(grepr), N, O, M, L, KS2, C, R, C2, _stream())
        return (ow.finish() if need_w else None, grepr) + (None,) * 7


def factored_layout_conv(f, weight, bias, stride, pad, reflect, act, slope, x2):
    """conv2d([layout | x2], weight) computed from the factored layout ``f`` (see FactoredLayout)."""
    M, Ctot, KS, _ = weight.shape
    KS2 = KS * KS
    R = f.repr.size(1)
    cfull = f.num_objs + R
    C2 = 0 if x2 is None else x2.size(1)
    assert Ctot == cfull + C2, 'weight has %d input channels, layout %d + second source %d' % (Ctot, cfull, C2)
    N = f.Z.size(0)
    clist, ccnt, extra_pos, L = f.lists(C2)
    # per-image filters (sg_factored_weights_fwd):  W_eff[o] = W[:, class_o] + sum_d repr[o, d] W[:, num_objs + d] for the
    # objects of the image, then the filters of the second source's channels.  (A discriminator inside the generator step
    # runs under skip_param_grads: PerImageConvFn then returns no gradient for the filters and the parameter's
    # AccumulateGrad never fires; the SAME recorded forward still yields the weight gradient when the discriminator
    # step differentiates it.)
    w_full = weight
    wimg = FactoredWeightsFn.apply(weight, f.repr, f.objs, f.seg, f.img_idx, N, L, f.num_objs, C2)
    planes = f.Z if x2 is None else torch.cat([f.Z, x2.detach()], 1)
    return PerImageConvFn.apply(planes, x2, wimg, bias, clist, ccnt, w_full.detach(), cfull, stride, pad, reflect, act,
                                float(slope))


class CropBBoxFn(Function):
    """crop_bbox_batch (bilinear.py:26-41,67-130): gather forward, scatter-add backward w.r.t. feats."""

    @staticmethod
    def forward(ctx, feats, boxes, idx, HH, WW):
        feats, boxes, idx = _f32(feats, 'feats'), _f32(boxes, 'bbox'), _i64(idx, 'bbox_to_feats')
        N, C, H, W = feats.shape
        B = boxes.size(0)
        out = torch.empty(B, C, HH, WW, dtype=torch.float32, device=feats.device)
        _call('sg_crop_bbox_fwd', _p(feats), _p(boxes), _p(idx), _p(out), N, C, H, W, B, HH, WW, _stream())
        ctx.cfg = (N, C, H, W, B, HH, WW)
        ctx.save_for_backward(boxes, idx)
        return out

    @staticmethod
    def backward(ctx, g):
        boxes, idx = ctx.saved_tensors
        N, C, H, W, B, HH, WW = ctx.cfg
        gf = None
        if ctx.needs_input_grad[0]:
            g = _f32(g)
            gf = torch.empty(N, C, H, W, dtype=torch.float32, device=g.device)     # the gather writes every element
            _call('sg_crop_bbox_bwd', _p(g), _p(boxes), _p(idx), _p(gf), N, C, H, W, B, HH, WW, _stream())
        return gf, None, None, None, None


def vector_pool_exchange(pool, vectors, plan):
    vectors = _f32(vectors)
    O, R = vectors.shape
    out = torch.empty_like(vectors)
    _call('sg_vector_pool_exchange', _p(pool), _p(vectors), _p(plan), _p(out), O, R, pool.size(1), _stream())
    return out


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


# =============================================================================================
# optimiser / misc
# =============================================================================================

def adam_step(p, g, m, v, lr, beta1, beta2, eps, step):
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
    _call('sg_adam_step', _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, bc1, bc2_sqrt, _stream())


def fill_(t, value):
    _call('sg_fill', _p(t), float(value), t.numel(), _stream())
    return t


def add_clear_(y, x):
    """y += x ; x = 0 (optim.FusedAdam._fold_spill)"""
    assert y.numel() == x.numel()
    _call('sg_add_clear', _p(y), _p(x), y.numel(), _stream())
    return y


def scale_(t, alpha):
    _call('sg_scale', _p(t), float(alpha), t.numel(), _stream())
    return t


# ---- profiler ----
_PROF_ON = False


def prof_enable(on=True):
    global _PROF_ON
    _PROF_ON = bool(on)
    _L().sg_prof_enable(1 if on else 0)


def prof_is_enabled():
    return _PROF_ON


def prof_reset():
    _L().sg_prof_reset()


def prof_read():
    """-> {kind: dict(ms, launches, flops, bytes)} since the last reset (synchronises the recorded events)."""
    L = _L()
    out = {}
    for k in range(L.sg_prof_num_kinds()):
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        n = ctypes.c_int64()
        L.sg_prof_read(k, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        out[L.sg_prof_kind_name(k).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out
