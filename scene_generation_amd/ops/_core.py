"""Autograd operators over the C ABI of libsg2im_hip.so (include/sg2im_hip.h).

PyTorch-ROCm is used here as plumbing only: device allocation (torch.empty), the autograd tape and
the current HIP stream.  Every forward/backward below is one or a few hand-written gfx950 kernel
launches through ctypes; there is no eager/CPU fallback -- a CPU tensor raises.


Binding helpers of the C ABI (ctypes calls, stream handle, workspace), the parameter-gradient sinks of FusedAdam, the path
switches, layout hints, flat-buffer updates and the profiler front end: everything the operator modules share.
"""
import contextlib
import os
import ctypes

import torch
from torch.autograd import Function

from .. import _hip
from .._hip import sgConvDesc

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


def _al16(t):
    """``t`` (contiguous) at a 16-byte aligned address: itself, or a copy when a view starts mid-vector.  The F(4x4,3x3) Winograd
    entry points take float4 loads of their operands and reject unaligned pointers (the form is a function of the conv desc alone:
    the saved-operand buffers are sized from it); torch allocations and FlatParams slices are 256-byte aligned, so the copy is for
    odd views only."""
    return t if t is None or (t.data_ptr() & 15) == 0 else t.clone()


def _i64(t, name='index'):
    _dev(t, name)
    if t.dtype != torch.int64:
        raise TypeError('%s must be int64, got %s' % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def workspace(nbytes, device):
    """Per-(device, stream) scratch, grown on demand.  Safe to share: every consumer is ordered on the current stream.

    While the current stream is being CAPTURED into a hipGraph the scratch is a fresh allocation of the capturing graph's private
    pool instead, dropped when the call returns (the pool hands the block to later captured allocations, which run later in the
    graph's order): a cached buffer would be baked into the graph by address and then (a) replaced -- i.e. freed -- by a larger
    request later in the same capture, or (b) belong to the private pool of an EARLIER graph whose owner is long gone (the cache
    outlives every Trainer): replays of the new graph then touch memory the allocator may have handed out again or returned to
    the driver.  Round 6: a memory access fault in the generator segment captured by the 170th test of one process."""
    if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
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
        if rc == SG_ERR_INDEX:          # an index operand outside its range: what the reference's indexing raises
            raise IndexError('%s: %s' % (name, _hip.last_error()))
        raise RuntimeError('%s failed (rc=%d): %s' % (name, rc, _hip.last_error()))


SG_ERR_INDEX = -2


def check_indices(idx, lo, hi, what):
    """With the library option ``check_indices`` on (``_hip.set_option('check_indices', 1)`` / SG_CHECK_INDICES=1): range-check
    an int64 index operand on the device and raise IndexError like the reference's indexing does (graph.py:79-80,
    model.py:131, bilinear.py:36).  The kernels themselves read indices unchecked.  One stream synchronisation per check."""
    if idx is not None and idx.is_cuda and _hip.option_cached('check_indices'):
        idx = idx if idx.is_contiguous() else idx.contiguous()
        _call('sg_check_indices', _p(idx), idx.numel(), int(lo), int(hi), what.encode(), _stream())


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
        elif _CAPTURE is None and getattr(sk.opt(), 'use_spill', False):
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
    # (the answers depend on run-time options -- wino_reuse, wino24, w24_pmin, ...: the generation counter of
    # _hip.set_option is part of the key, so a changed option is never answered from a stale memo: ADVICE r4)
    key = (name, _hip.OPTION_GENERATION) + extra
    v = d._memo.get(key)
    if v is None:
        v = d._memo[key] = getattr(_L(), name)(d._ref, *extra)
    return v


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
_HINT_KEYS = ('sparse', 'sparse_cat', 'grad_from', 'factored', 'keep_grad', 'pending', 'wrong_twin')


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


# Interpolate(x2, nearest) + Conv2d(3, padding=1) as a sub-pixel transposed convolution (SG_UPCONV=0: 3x3 gather over the
# folded upsample instead): 16 instead of 36 multiply-adds per input pixel and channel pair (mask_net, generators.py:20-21)
UPCONV = os.environ.get('SG_UPCONV', '1') != '0'
# A conv over [feature map || per-sample row expanded over the grid] (the class one-hot of the mask discriminator,
# discriminators.py:107-110) with the constant channels folded into a per-(sample, output channel, tap) term: only the feature
# map's channels are gathered (SG_COND_FOLD=0: the row is a broadcast second gather source of the full-width conv)
COND_FOLD = os.environ.get('SG_COND_FOLD', '1') != '0'


# =============================================================================================
# optimiser / misc
# =============================================================================================

def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = (1.0 - beta2 ** step) ** 0.5
    _call('sg_adam_step', _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, bc1, bc2_sqrt, float(grad_scale),
          _stream())


def fill_(t, value):
    _call('sg_fill', _p(t), float(value), t.numel(), _stream())
    return t


def stage_copy(dst, src_pinned, nbytes):
    """dst (uint8, device) <- the first ``nbytes`` of ``src_pinned`` (uint8, page-locked host memory) by ONE kernel on the current
    stream that reads the host buffer through its device mapping (pipeline.DeviceBatchPrefetcher)."""
    assert dst.is_cuda and dst.dtype == torch.uint8 and src_pinned.dtype == torch.uint8 and not src_pinned.is_cuda
    assert dst.numel() >= nbytes and src_pinned.numel() >= nbytes
    with torch.cuda.device(dst.device):
        _call('sg_stage_copy', ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src_pinned.data_ptr()), int(nbytes), _stream())
    return dst


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
