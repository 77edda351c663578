"""Convolution family: nn.Conv2d / nn.ConvTranspose2d / the sub-pixel form of Interpolate(x2)+Conv3x3 / nn.Linear as autograd
Functions over the implicit-GEMM, Winograd, head and skinny kernels.
(Part of scene_generation_amd.ops: see ops/__init__.py.)"""

import torch
from torch.autograd import Function

from . import _core
from ._core import (ACT_NONE, GradOut, _L, _al16, _call, _conv_desc, _f32, _p, _q, _stream, _wants_grad, conv_out_size,
    ensure_dense, hints_of, scale_, workspace)
from .layout import (factored_layout_conv)


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
        ctx.wino = (x2 is None and sparse is None and _core.WINOGRAD and bool(_q(d, 'sg_conv2d_wino_supported')))
        if ctx.wino:                  # (float4 operand loads: a view that starts mid-vector is copied, see _al16)
            x1, weight = _al16(x1), _al16(weight)
        ctx.head = (x2 is None and sparse is None and _core.HEADCONV and not ctx.wino and not ctx.smallm
                    and bool(_q(d, 'sg_conv2d_head_supported')))
        ctx.wino24 = (x2 is None and sparse is None and _core.WINOGRAD24 and not ctx.head and not ctx.smallm
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
            # ... and the weight gradient multiplies with the input transform this forward builds anyway: keep it (33.5 MB per
            # ResnetBlock conv at the benchmark shape) instead of transforming x a second time in the backward
            vn = _q(d, 'sg_conv2d_wino_v_floats') if (ctx.needs_input_grad[0] and ctx.needs_input_grad[2]) else 0
            ctx.wino_v = torch.empty(vn, dtype=torch.float32, device=x1.device) if vn else None
            _call('sg_conv2d_wino_fwd', d._ref, _p(x1), _p(weight), _p(bias), _p(y), act, slope, _p(ctx.wino_ut),
                  _p(ctx.wino_v), _p(workspace(wsb, x1.device)), wsb, _stream())
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
        if ctx.wino:
            gy = _al16(gy)
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
        wino_keep = {'ytp': None}
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
                    # the gradient transform is the other operand of this conv's weight gradient: keep it when that follows
                    yn = _q(d, 'sg_conv2d_wino_ytp_floats') if (need_w and getattr(ctx, 'wino_v', None) is not None) else 0
                    wino_keep['ytp'] = torch.empty(yn, dtype=torch.float32, device=dev) if yn else None
                    _call('sg_conv2d_wino_dgrad', d._ref, _p(gy), _p(weight), _p(out), _p(getattr(ctx, 'wino_ut', None)),
                          _p(wino_keep['ytp']), _p(workspace(fb, dev)), fb, s)
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
                    ytp = wino_keep['ytp']
                    _call('sg_conv2d_wino_wgrad', d._ref, _p(gy), _p(x1), _p(gw),
                          _p(ctx.wino_v) if ytp is not None else None, _p(ytp), _p(ws), wsb, s)
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


class ConvInstNormFn(Function):
    """act(InstanceNorm2d(Conv2d(ReflectionPad2d(1)(x)))) [+ skip] of a ResnetBlock (layers.py:251-270) on the Winograd F(4x4,3x3)
    path with the normalisation fused at both ends of the batched GEMMs (sg_conv2d_wino_fwd_instnorm /
    sg_conv2d_wino_dgrad_instnorm): one launch less in each direction and one pass less over the conv result than ConvFn +
    InstanceNormFn, same arithmetic up to the order of the per-plane sums."""

    @staticmethod
    def forward(ctx, x, weight, bias, skip, eps, act, slope):
        x = _al16(_f32(x, 'conv input'))               # (the fused entry points reject operands that start mid-vector: ADVICE r5)
        weight = _al16(_f32(weight, 'conv weight'))
        skip = None if skip is None else _al16(_f32(skip))
        N, C, H, W = x.shape
        Cout = weight.size(0)
        d = _conv_desc(N, C, 0, H, W, Cout, 3, 1, 1, True, 1, H, W, 0, 0)
        dev = x.device
        ypre = torch.empty(N, Cout, H, W, dtype=torch.float32, device=dev)
        out = torch.empty_like(ypre)
        mean = torch.empty(N * Cout, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        utn = _q(d, 'sg_conv2d_wino_ut_floats') if need_x else 0
        ut = torch.empty(utn, dtype=torch.float32, device=dev) if utn else None
        vn = _q(d, 'sg_conv2d_wino_v_floats') if need_w else 0
        v = torch.empty(vn, dtype=torch.float32, device=dev) if vn else None
        wsb = _q(d, 'sg_conv2d_wino_ws_bytes')
        _call('sg_conv2d_wino_fwd_instnorm', d._ref, _p(x), _p(weight), _p(bias), _p(skip), _p(ypre), _p(out), _p(mean), _p(rstd),
              eps, act, slope, _p(ut), _p(v), _p(workspace(wsb, dev)), wsb, _stream())
        ctx.desc = d
        ctx.bias_ref = bias
        ctx.cfg = (act, slope, bias is not None, skip is not None)
        ctx.wino_ut, ctx.wino_v = ut, v
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, weight, ypre, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, gout):
        if gout is None:
            return (None,) * 7
        x, weight, ypre, mean, rstd = ctx.saved_tensors
        d = ctx.desc
        act, slope, has_bias, has_skip = ctx.cfg
        gout = _al16(_f32(gout))
        s = _stream()
        dev = gout.device
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1] and _wants_grad(weight)
        need_b = has_bias and ctx.needs_input_grad[2] and _wants_grad(ctx.bias_ref)
        gconv = torch.empty_like(ypre)
        gx = torch.empty(d.N, d.C1, d.H, d.W, dtype=torch.float32, device=dev) if need_x else None
        yn = _q(d, 'sg_conv2d_wino_ytp_floats') if (need_w and ctx.wino_v is not None) else 0
        ytp = torch.empty(yn, dtype=torch.float32, device=dev) if yn else None
        wsb = _q(d, 'sg_conv2d_wino_ws_bytes')
        # the bias gradient comes out of the same launch (per-image plane sums of the conv's gy + one small launch over the images)
        ob = GradOut(ctx.bias_ref) if need_b else None
        _call('sg_conv2d_wino_dgrad_instnorm', d._ref, _p(gout), _p(ypre), _p(mean), _p(rstd), act, slope, _p(weight), _p(gconv),
              _p(gx), _p(ob.buf) if need_b else None, _p(ctx.wino_ut), _p(ytp), _p(workspace(wsb, dev)), wsb, s)
        gw = gb = None
        if need_w or need_b:
            ow = GradOut(weight) if need_w else None
            wsb2 = max(wsb, _L().sg_channel_sum_ws_bytes(d.Cout))
            ws = workspace(wsb2, dev)
            if need_w:
                _call('sg_conv2d_wino_wgrad', d._ref, _p(gconv), _p(x), _p(ow.buf), _p(ctx.wino_v) if ytp is not None else None,
                      _p(ytp), _p(ws), wsb2, s)
            gw = ow.finish() if need_w else None
            gb = ob.finish() if need_b else None
        return gx, gw, gb, (gout if has_skip and ctx.needs_input_grad[3] else None), None, None, None


def conv_instnorm_fusable(x, weight, reflect_pad, stride, pad):
    """ReflectionPad2d(1) + Conv2d(3, stride 1) + InstanceNorm2d on the fused Winograd F(4x4,3x3) path?"""
    if not (_core.WINOGRAD and x.is_cuda and x.dim() == 4 and reflect_pad == 1 and stride == 1 and pad == 0
            and weight.size(2) == 3 and weight.size(3) == 3 and weight.size(1) == x.size(1)):
        return False
    N, C, H, W = x.shape
    d = _conv_desc(N, C, 0, H, W, weight.size(0), 3, 1, 1, True, 1, H, W, 0, 0)
    return bool(_q(d, 'sg_conv2d_wino_in_supported'))


def conv2d_instnorm(x, weight, bias, skip=None, eps=1e-5, act=ACT_NONE, slope=0.0):
    return ConvInstNormFn.apply(x, weight, bias, skip, float(eps), act, float(slope))


class CondConv2dFn(Function):
    """act(conv2d([x1 || cond[:, :, None, None].expand(H, W)]) + bias) for a per-sample row ``cond`` [N, C2], zero padding: the
    constant channels never enter the gather.  With W = [W1 | W2] along the input channels,
        y[n, m, oh, ow] = conv(x1, W1)[n, m, oh, ow] + sum_{taps (kh, kw) that land inside the plane at (oh, ow)} P[n, m, kh, kw],
        P[n, m, t] = sum_c2 cond[n, c2] W2[m, c2, t]                     (one dense layer, [N x C2] x [C2 x Cout*KS*KS])
    -- C1 instead of C1 + C2 gathered channels in the forward and the weight-gradient GEMMs (128 of 300 in the mask
    discriminator, discriminators.py:107-110,147-154).  Backward: the data gradient and dW1 are the C1-channel conv's; dP is the
    per-tap window sum of the output gradient, dW2 = dP^T cond and dcond = dP W2 dense layers; dW1 and dW2 are interleaved into
    the parameter's gradient by one kernel.  Whether the weight gradient is wanted is decided when the backward runs
    (ops.skip_param_grads: a forward recorded during the generator step is re-used by the discriminator step).
    The C1-channel conv always runs the implicit-GEMM gather: at 128 channels Winograd F(2x2,3x3) measured slower (round 6:
    1007-1009 against 1010-1011 images/s at configs[1], 525 against 531 at configs[4])."""

    @staticmethod
    def forward(ctx, x1, cond, weight, bias, stride, pad, act, slope):
        x1, cond, weight = _f32(x1, 'conv input'), _f32(cond, 'conv condition row'), _f32(weight, 'conv weight')
        N, C1, H, W = x1.shape
        C2 = cond.size(1)
        Cout, Cin, KS, KS2 = weight.shape
        assert KS == KS2 and Cin == C1 + C2 and cond.size(0) == N and N > 0
        R = KS * KS
        OH, OW = conv_out_size(H, KS, stride, pad), conv_out_size(W, KS, stride, pad)
        d = _conv_desc(N, C1, 0, H, W, Cout, KS, stride, pad, False, 1, OH, OW, 0, 0)
        dev, s = x1.device, _stream()
        w1 = torch.empty(Cout, C1, KS, KS, dtype=torch.float32, device=dev)
        w2r = torch.empty(Cout * R, C2, dtype=torch.float32, device=dev)
        _call('sg_cond_conv_split_w', _p(weight), _p(w1), _p(w2r), Cout, C1, C2, R, s)
        proj = torch.empty(N, Cout * R, dtype=torch.float32, device=dev)
        _call('sg_linear_fwd', _p(cond), _p(w2r), None, _p(proj), N, C2, Cout * R, ACT_NONE, 0.0, s)
        y = torch.empty(N, Cout, OH, OW, dtype=torch.float32, device=dev)
        wsb = _q(d, 'sg_conv2d_ws_bytes', 0)
        _call('sg_conv2d_fwd', d._ref, _p(x1), None, _p(w1), _p(bias), _p(y), ACT_NONE, 0.0, _p(workspace(wsb, dev)), wsb, s)
        _call('sg_cond_conv_bias_act', _p(y), _p(proj), N * Cout, OH, OW, H, W, KS, stride, pad, act, slope, s)
        ctx.desc, ctx.geom = d, (C2, R, KS, stride, pad)
        ctx.weight_ref, ctx.bias_ref = weight, bias          # identities only (gradient sinks / skip list)
        ctx.cfg = (act, slope, bias is not None)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x1, cond, w1, w2r, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 8
        x1, cond, w1, w2r, y = ctx.saved_tensors
        d = ctx.desc
        C2, R, KS, stride, pad = ctx.geom
        act, slope, has_bias = ctx.cfg
        gy = _f32(gy)
        dev, s = gy.device, _stream()
        if act != ACT_NONE:
            g2 = torch.empty_like(gy)
            _call('sg_act_bwd', _p(y), _p(gy), _p(g2), gy.numel(), act, slope, s)
            gy = g2
        need_x1, need_c = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_w = ctx.needs_input_grad[2] and _wants_grad(ctx.weight_ref)
        need_b = has_bias and ctx.needs_input_grad[3] and _wants_grad(ctx.bias_ref)
        gx1 = gc = gw = gb = None
        if need_x1:
            gx1 = torch.empty(d.N, d.C1, d.H, d.W, dtype=torch.float32, device=dev)
            wsb = _q(d, 'sg_conv2d_ws_bytes', 1)
            _call('sg_conv2d_dgrad', d._ref, _p(gy), _p(w1), _p(gx1), 0, d.C1, _p(workspace(wsb, dev)), wsb, s)
        gproj = None
        if need_c or need_w:
            gproj = torch.empty(d.N, d.Cout * R, dtype=torch.float32, device=dev)
            _call('sg_cond_conv_window_sums', _p(gy), _p(gproj), d.N * d.Cout, d.OH, d.OW, d.H, d.W, KS, stride, pad, s)
        if need_c:
            gc = torch.empty_like(cond)
            _call('sg_linear_bwd_data', _p(gproj), _p(w2r), _p(gc), d.N, C2, d.Cout * R, s)
        if need_w or need_b:
            ow = GradOut(ctx.weight_ref) if need_w else None
            ob = GradOut(ctx.bias_ref) if need_b else None
            if need_w:
                gw1, gw2r = torch.empty_like(w1), torch.empty_like(w2r)
                wsb = _q(d, 'sg_conv2d_ws_bytes', 2)
                _call('sg_conv2d_wgrad', d._ref, _p(gy), _p(x1), None, _p(gw1), _p(ob.buf) if need_b else None,
                      _p(workspace(wsb, dev)), wsb, s)
                _call('sg_linear_bwd_weight', _p(gproj), _p(cond), _p(gw2r), None, d.N, C2, d.Cout * R, s)
                _call('sg_cond_conv_merge_w', _p(gw1), _p(gw2r), _p(ow.buf), d.Cout, d.C1, C2, R, s)
            else:
                wsb = _L().sg_channel_sum_ws_bytes(d.Cout)
                _call('sg_channel_sum', _p(gy), _p(ob.buf), d.N, d.Cout, d.OH * d.OW, _p(workspace(wsb, dev)), wsb, s)
            gw = ow.finish() if need_w else None
            gb = ob.finish() if need_b else None
        return gx1, gc, gw, gb, None, None, None, None


def _cond_fold_applies(x, weight, stride, pad, reflect, upsample, x2):
    """a [N, C2] row as second source of a zero-padded conv with a kernel size the window-sum kernel has (1, 3, 4) and more than
    four output channels (below that the dense conv runs the vector-ALU kernels, which have no two-source form to fold)"""
    if not (_core.COND_FOLD and x2 is not None and x2.dim() == 2 and not reflect and upsample == 1 and x.size(0) > 0):
        return False
    KS = weight.size(2)
    if KS != weight.size(3) or KS not in (1, 3, 4) or weight.size(1) != x.size(1) + x2.size(1) or weight.size(0) <= 4:
        return False
    return conv_out_size(x.size(2), KS, stride, pad) > 0 and conv_out_size(x.size(3), KS, stride, pad) > 0


def _upconv_prefers_winograd(x, weight):
    """the folded-upsample Winograd path (>= 128 channels in multiples of 128) keeps its convs"""
    if not _core.WINOGRAD:
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
    if (_core.UPCONV and upsample == 2 and stride == 1 and pad == 1 and not reflect and x2 is None and h is None
            and act == ACT_NONE and weight.size(2) == 3 and weight.size(3) == 3 and not _upconv_prefers_winograd(x, weight)):
        return UpConv3Fn.apply(x, weight, bias)
    if h is None:
        if x2 is not None and _cond_fold_applies(x, weight, stride, pad, reflect, upsample, x2):
            return CondConv2dFn.apply(x, x2, weight, bias, stride, pad, act, float(slope))
        return Conv2dFn.apply(x, x2, weight, bias, stride, pad, reflect, upsample, act, float(slope), 0, None)
    f = h.get('factored') if _core.FACTORED_LAYOUT else None
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
