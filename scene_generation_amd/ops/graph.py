"""Scene-graph operators (graph.py:58-122, model.py:131-134): CSR build, row gather + concat, the bit-exact triple pool, embeddings.
(Part of scene_generation_amd.ops: see ops/__init__.py.)"""

import torch
from torch.autograd import Function

from . import _core
from ._core import (ACT_NONE, GradOut, _L, _call, _f32, _i64, _p, _stream, _wants_grad, workspace)


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
        _core.check_indices(edges, 0, obj.size(0), 'edges (subject / object node ids)')      # graph.py:79-80
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


class GatherLinearFn(Function):
    """act([obj[s], pred, obj[o]] W^T + b): the row gather of graph.py:79-84 and the first Linear (+ReLU) of net1 (graph.py:58-60,86)
    as ONE launch -- the A loader of the register-streaming GEMM reads the node / edge feature rows, the (T, 2 Do + Dp) matrix is
    never written in the forward (sg_gconv_gather_linear_fwd).  The backward materialises the rows once for the weight gradient
    (they are the GEMM's second operand there) and scatters the data gradient with the deterministic segmented sums of
    GatherConcatFn."""

    @staticmethod
    def forward(ctx, obj, pred, edges, off, ent, weight, bias, act, slope):
        obj, pred, weight = _f32(obj), _f32(pred), _f32(weight, 'linear weight')
        T, Do, Dp = edges.size(0), obj.size(1), pred.size(1)
        out_f = weight.size(0)
        assert weight.size(1) == 2 * Do + Dp
        _core.check_indices(edges, 0, obj.size(0), 'edges (subject / object node ids)')      # graph.py:79-80
        y = torch.empty(T, out_f, dtype=torch.float32, device=obj.device)
        wsb = _L().sg_gconv_gather_linear_ws_bytes(T, Do, Dp, out_f)
        ws = workspace(wsb, obj.device) if wsb else None
        _call('sg_gconv_gather_linear_fwd', _p(obj), _p(pred), _p(edges), _p(weight), _p(bias), _p(y), T, Do, Dp, out_f, act,
              slope, _p(ws), wsb, _stream())
        ctx.cfg = (act, slope, bias is not None, obj.size(0), T, Do, Dp, out_f)
        ctx.bias_ref = bias
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(obj, pred, edges, off, ent, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 9
        obj, pred, edges, off, ent, weight, y = ctx.saved_tensors
        act, slope, has_bias, O, T, Do, Dp, out_f = ctx.cfg
        K = 2 * Do + Dp
        gy = _f32(gy)
        s = _stream()
        dev = gy.device
        if T == 0:
            return (torch.zeros_like(obj), torch.zeros_like(pred), None, None, None, torch.zeros_like(weight),
                    torch.zeros(out_f, device=dev) if has_bias else None, None, None)
        if act != ACT_NONE:
            g2 = torch.empty_like(gy)
            _call('sg_act_bwd', _p(y), _p(gy), _p(g2), gy.numel(), act, slope, s)
            gy = g2
        need_w = ctx.needs_input_grad[5] and _wants_grad(weight)
        need_b = has_bias and ctx.needs_input_grad[6] and _wants_grad(ctx.bias_ref)
        ow = GradOut(weight) if need_w else None
        ob = GradOut(ctx.bias_ref) if need_b else None
        if need_w:
            cur_t = torch.empty(T, K, dtype=torch.float32, device=dev)
            _call('sg_gather_concat_fwd', _p(obj), _p(pred), _p(edges), _p(cur_t), T, Do, Dp, s)
            _call('sg_linear_bwd_weight', _p(gy), _p(cur_t), _p(ow.buf), _p(ob.buf) if need_b else None, T, K, out_f, s)
        elif need_b:
            _call('sg_channel_sum', _p(gy), _p(ob.buf), T, out_f, 1, None, 0, s)
        g_obj = g_pred = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            g = torch.empty(T, K, dtype=torch.float32, device=dev)
            _call('sg_linear_bwd_data', _p(gy), _p(weight), _p(g), T, K, out_f, s)
            if ctx.needs_input_grad[0]:
                g_obj = torch.empty(O, Do, dtype=torch.float32, device=dev)
                _call('sg_segment_sum', _p(g), K, 0, Do + Dp, Do, _p(off), _p(ent), _p(g_obj), O, 0, s)
            if ctx.needs_input_grad[1]:
                g_pred = torch.empty(T, Dp, dtype=torch.float32, device=dev)
                _call('sg_copy_cols', _p(g), K, Do, _p(g_pred), Dp, 0, T, Dp, s)
        gw = ow.finish() if need_w else None
        gb = ob.finish() if need_b else None
        return g_obj, g_pred, None, None, None, gw, gb, None, None


def gather_linear(obj, pred, edges, off, ent, weight, bias, act=ACT_NONE, slope=0.0):
    return GatherLinearFn.apply(obj, pred, edges, off, ent, weight, bias, act, float(slope))


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
        _core.check_indices(idx, 0, table.size(0), 'embedding index')                         # model.py:131-132
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
