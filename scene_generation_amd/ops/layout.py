"""Layout scatter and crops (layout.py:64-184, bilinear.py:67-130): masks_to_layout in its dense, deferred, test-mode and factored
forms, the per-image-weight convs over the factored layout, bilinear crops, VectorPool exchange.
(Part of scene_generation_amd.ops: see ops/__init__.py.)"""

import torch
from torch.autograd import Function

from . import _core
from ._core import (ACT_NONE, GradOut, _L, _call, _conv_desc, _dev, _f32, _i64, _p, _q, _stream, _wants_grad,
    conv_out_size, set_hints, workspace)
from .graph import (one_hot)


# =============================================================================================
# layout + crops + vector pool
# =============================================================================================

def segment_offsets(obj_to_img, N):
    obj_to_img = _i64(obj_to_img, 'obj_to_img')
    _core.check_indices(obj_to_img, 0, N, 'obj_to_img')
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
            from ..utils import to_device_async
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

    @staticmethod
    def stacked(a, b):
        """The batch concatenation [a ; b] of two factored layouts over the SAME planes and objects (the ground-truth layout and its
        wrong-texture twin, model.py:119-124: only the appearance vectors differ): images N.. are b's.  Appearance vectors enter
        detached (the only consumer is a discriminator pass that must not reach the generator)."""
        assert a.Z is b.Z and a.objs is b.objs and a.counts_host == b.counts_host and a.num_objs == b.num_objs
        N = a.Z.size(0)
        f = FactoredLayout(torch.cat([a.Z, a.Z], 0), torch.cat([a.objs, a.objs], 0),
                           torch.cat([a.repr.detach(), b.repr.detach()], 0), a.num_objs,
                           torch.cat([a.img_idx, a.img_idx + N], 0), torch.cat([a.plane_idx, a.plane_idx], 0),
                           a.counts_host + a.counts_host)
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
            from ..utils import to_device_async
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


class CropBBoxJJFn(Function):
    """crop_bbox(feats, bbox, HH, WW, backend='jj'): the ``bilinear_sample`` geometry (bilinear.py:127-128,188-243), one box per image;
    gradient w.r.t. feats."""

    @staticmethod
    def forward(ctx, feats, boxes, HH, WW):
        feats, boxes = _f32(feats, 'feats'), _f32(boxes, 'bbox')
        N, C, H, W = feats.shape
        out = torch.empty(N, C, HH, WW, dtype=torch.float32, device=feats.device)
        _call('sg_crop_bbox_jj_fwd', _p(feats), _p(boxes), _p(out), N, C, H, W, HH, WW, _stream())
        ctx.dims = (N, C, H, W, HH, WW)
        ctx.save_for_backward(boxes)
        return out

    @staticmethod
    def backward(ctx, g):
        boxes, = ctx.saved_tensors
        N, C, H, W, HH, WW = ctx.dims
        gf = None
        if ctx.needs_input_grad[0]:
            g = _f32(g)
            gf = torch.empty(N, C, H, W, dtype=torch.float32, device=g.device)
            _call('sg_crop_bbox_jj_bwd', _p(g), _p(boxes), _p(gf), N, C, H, W, HH, WW, _stream())
        return gf, None, None, None


class CropBBoxFn(Function):
    """crop_bbox_batch (bilinear.py:26-41,67-130): gather forward, scatter-add backward w.r.t. feats."""

    @staticmethod
    def forward(ctx, feats, boxes, idx, HH, WW):
        feats, boxes, idx = _f32(feats, 'feats'), _f32(boxes, 'bbox'), _i64(idx, 'bbox_to_feats')
        _core.check_indices(idx, 0, feats.size(0), 'bbox_to_feats')                           # bilinear.py:36
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
