"""ORACLE (test infrastructure, NOT product code).

A from-scratch PyTorch-CPU restatement of the scene-graph -> image G+D training step of
ashual/scene_generation, written against the reference's behaviour (file:line citations are
relative to /root/reference/).  It exists only so that

  * tests/ can check the hand-written HIP path against it on identical seeded inputs,
  * __graft_entry__.smoke() can check one small invocation,
  * bench.py's ``cpu_baseline`` leg can time the same workload on the host cores.

Nothing under scene_generation_amd/ imports this file.  Parity pin: tests/test_oracle_golden.py
compares every function here with golden vectors captured by importing the reference itself in
the build container (tools/make_golden.py -> tests/golden/*.npz).

The fused formulations used by the HIP kernels are restated here in plain fp32 torch:
  * masks_to_layout : factored  out[n,d] = sum_o vecs[o,d] * S_o, S_o = bilinear(mask_o)   (layout.py:64-93)
  * crop_bbox_batch : direct per-box bilinear gather, no expand/cat/invperm              (bilinear.py:67-130)
  * GraphTripleConv : sequential s-pass then o-pass pooling                              (graph.py:94-116)
All state_dict key names equal the reference's.
"""
import random

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------
# bilinear sampling primitive: F.grid_sample(bilinear, zeros padding, align_corners=False) as
# executed by torch>=1.3, written out explicitly.
# ------------------------------------------------------------------------------------------

def _bilinear_taps(px, size):
    """px: float pixel coordinate (already un-normalised).  Returns (i0, i1, w0, w1, in0, in1)."""
    f = torch.floor(px)
    w1 = px - f
    w0 = 1.0 - w1
    i0 = f.long()
    i1 = i0 + 1
    in0 = ((i0 >= 0) & (i0 < size)).to(px.dtype)
    in1 = ((i1 >= 0) & (i1 < size)).to(px.dtype)
    return i0.clamp(0, size - 1), i1.clamp(0, size - 1), w0 * in0, w1 * in1


# The authors ran PyTorch 1.0 (requirements.txt:8), where F.grid_sample without an ``align_corners`` argument meant
# align_corners=True; their released checkpoints were trained with that geometry.  LEGACY_ALIGN_CORNERS switches the whole
# oracle to it (the HIP path has the same switch: scene_generation_amd.set_legacy_align_corners).
LEGACY_ALIGN_CORNERS = False


def bilinear_sample(src, gx, gy):
    """src (B,C,Hs,Ws); gx (B,Wo) / gy (B,Ho) separable normalised grids in [-1,1] space.
    Returns (B,C,Ho,Wo).  Un-normalisation: p = ((g+1)/2)*size - 0.5  (align_corners=False) or, in legacy mode,
    p = ((g+1)/2)*(size-1)  (align_corners=True)."""
    B, C, Hs, Ws = src.shape
    if LEGACY_ALIGN_CORNERS:
        px = (gx + 1) / 2 * (Ws - 1)
        py = (gy + 1) / 2 * (Hs - 1)
    else:
        px = ((gx + 1) * Ws - 1) / 2
        py = ((gy + 1) * Hs - 1) / 2
    x0, x1, wx0, wx1 = _bilinear_taps(px, Ws)
    y0, y1, wy0, wy1 = _bilinear_taps(py, Hs)
    bidx = torch.arange(B)[:, None, None]

    def tap(yi, xi):
        return src[bidx, :, yi[:, :, None], xi[:, None, :]].permute(0, 3, 1, 2)  # (B,C,Ho,Wo)

    out = tap(y0, x0) * (wy0[:, None, :, None] * wx0[:, None, None, :])
    out = out + tap(y0, x1) * (wy0[:, None, :, None] * wx1[:, None, None, :])
    out = out + tap(y1, x0) * (wy1[:, None, :, None] * wx0[:, None, None, :])
    out = out + tap(y1, x1) * (wy1[:, None, :, None] * wx1[:, None, None, :])
    return out


# ------------------------------------------------------------------------------------------
# layout.py
# ------------------------------------------------------------------------------------------

def _segments(obj_to_img):
    """Contiguous, gap-free image segments (layout.py:149-155 relies on this)."""
    N = int(obj_to_img.max()) + 1
    counts = torch.bincount(obj_to_img, minlength=N)
    if (counts == 0).any() or not bool((obj_to_img[1:] >= obj_to_img[:-1]).all()):
        raise ValueError('obj_to_img must be sorted and cover every image')
    ends = torch.cumsum(counts, 0)
    return N, ends - counts, ends, counts


def _box_grid(boxes, H, W):
    """layout.py:96-128: X=linspace(0,1,W); gx=((X-x0)/(x1-x0))*2-1 (same for y)."""
    x0, y0, x1, y1 = boxes[:, 0:1], boxes[:, 1:2], boxes[:, 2:3], boxes[:, 3:4]
    X = torch.linspace(0, 1, steps=W).view(1, W).to(boxes)
    Y = torch.linspace(0, 1, steps=H).view(1, H).to(boxes)
    gx = ((X - x0) / (x1 - x0)).mul(2).sub(1)
    gy = ((Y - y0) / (y1 - y0)).mul(2).sub(1)
    return gx, gy


def masks_to_layout(vecs, boxes, masks, obj_to_img, H, W=None, pooling='sum', test_mode=False):
    """layout.py:64-93 (both branches of _pool_samples, layout.py:131-183)."""
    O, D = vecs.shape
    M = masks.size(1)
    assert masks.shape == (O, M, M)
    W = H if W is None else W
    gx, gy = _box_grid(boxes, H, W)
    S = bilinear_sample(masks.to(vecs.dtype).view(O, 1, M, M), gx, gy)[:, 0]      # (O,H,W); masks.float() in fp32
    N, starts, ends, counts = _segments(obj_to_img)
    out = vecs.new_zeros(N, D, H, W)
    if test_mode:
        # layout.py:87-92,157-169: per image the objects are visited in ascending "mass" (sum of the sampled
        # vecs (x) mask tensor); a pixel belongs to the first visited object whose sampled mask exceeds 0.5
        import numpy as np
        for n in range(N):
            lo, hi = int(starts[n]), int(ends[n])
            samples = [vecs[o].view(D, 1, 1) * S[o].view(1, H, W) for o in range(lo, hi)]
            mass = [float(t.sum()) for t in samples]
            taken = vecs.new_zeros(H, W)
            for j in np.argsort(mass):
                m = (taken == 0).to(vecs.dtype) * (S[lo + j] > 0.5).to(vecs.dtype)
                taken = taken + m
                out[n] = out[n] + samples[j] * m
        if pooling == 'avg':
            out = out / counts.clamp(min=1).to(out).view(N, 1, 1, 1)
        elif pooling != 'sum':
            raise ValueError('Invalid pooling "%s"' % pooling)
        return out
    for n in range(N):
        acc = None
        for o in range(int(starts[n]), int(ends[n])):                        # ascending o
            term = vecs[o].view(D, 1, 1) * S[o].view(1, H, W)
            acc = term if acc is None else acc + term
        out[n] = acc
    if pooling == 'avg':
        out = out / counts.clamp(min=1).to(out).view(N, 1, 1, 1)
    elif pooling != 'sum':
        raise ValueError('Invalid pooling "%s"' % pooling)
    return out


def boxes_to_layout(vecs, boxes, obj_to_img, H, W=None, pooling='sum'):
    """layout.py:28-61 INTENDED semantics (the reference raises TypeError at :59 -- parity unpinned):
    masks_to_layout with an all-ones 8x8 mask (layout.py:50)."""
    ones = torch.ones(vecs.size(0), 8, 8, dtype=vecs.dtype, device=vecs.device)
    return masks_to_layout(vecs, boxes, ones, obj_to_img, H, W, pooling)


# ------------------------------------------------------------------------------------------
# bilinear.py
# ------------------------------------------------------------------------------------------

def crop_bbox_batch(feats, bbox, bbox_to_feats, HH, WW=None, backend='cudnn'):
    """bilinear.py:26-41,67-98,101-130,246-275.  grid x[j] = (1-j/(WW-1))*x0' + (j/(WW-1))*x1',
    x' = 2x-1, computed as start_w*start + end_w*end with linspace(1,0) / linspace(0,1)."""
    WW = HH if WW is None else WW
    b = 2 * bbox - 1
    sw_x = torch.linspace(1, 0, steps=WW).to(feats)
    ew_x = torch.linspace(0, 1, steps=WW).to(feats)
    sw_y = torch.linspace(1, 0, steps=HH).to(feats)
    ew_y = torch.linspace(0, 1, steps=HH).to(feats)
    gx = sw_x[None] * b[:, 0:1] + ew_x[None] * b[:, 2:3]
    gy = sw_y[None] * b[:, 1:2] + ew_y[None] * b[:, 3:4]
    return bilinear_sample(feats[bbox_to_feats], gx, gy)


def crop_bbox_jj(feats, bbox, HH, WW=None):
    """crop_bbox(feats, bbox, HH, WW, backend='jj') = bilinear_sample on the box grid (bilinear.py:101-130,188-243): one box per
    image, grid x[j] = start_w[j] * x0 + end_w[j] * x1 in [0, 1] (tensor_linspace, bilinear.py:246-275), pixel coordinate
    X = x * W (no half-pixel shift), taps floor(X) and floor(X) + 1 clamped to [0, W - 1], weights (x1 - X)(y1 - Y) etc. taken
    from the CLAMPED tap positions."""
    WW = HH if WW is None else WW
    N, C, H, W = feats.shape
    X = (torch.linspace(1, 0, steps=WW).to(feats)[None] * bbox[:, 0:1] + torch.linspace(0, 1, steps=WW).to(feats)[None] * bbox[:, 2:3]) * W
    Y = (torch.linspace(1, 0, steps=HH).to(feats)[None] * bbox[:, 1:2] + torch.linspace(0, 1, steps=HH).to(feats)[None] * bbox[:, 3:4]) * H
    x0 = X.floor().clamp(min=0, max=W - 1)
    x1 = (x0 + 1).clamp(min=0, max=W - 1)
    y0 = Y.floor().clamp(min=0, max=H - 1)
    y1 = (y0 + 1).clamp(min=0, max=H - 1)
    n = torch.arange(N)[:, None, None]

    def tap(yi, xi):                                   # (N, C, HH, WW)
        return feats[n, :, yi.long()[:, :, None], xi.long()[:, None, :]].permute(0, 3, 1, 2)

    wx0, wx1 = (x1 - X)[:, None, None, :], (X - x0)[:, None, None, :]
    wy0, wy1 = (y1 - Y)[:, None, :, None], (Y - y0)[:, None, :, None]
    return (wx0 * wy0) * tap(y0, x0) + (wx0 * wy1) * tap(y1, x0) + (wx1 * wy0) * tap(y0, x1) + (wx1 * wy1) * tap(y1, x1)


# ------------------------------------------------------------------------------------------
# layers.py
# ------------------------------------------------------------------------------------------

def build_mlp(dim_list, activation='relu', batch_norm='none', dropout=0, final_nonlinearity=True):
    """layers.py:215-231 -- note the non-linearity also follows the LAST Linear."""
    mods = []
    last = len(dim_list) - 2
    for i in range(last + 1):
        mods.append(nn.Linear(dim_list[i], dim_list[i + 1]))
        if i != last or final_nonlinearity:
            if batch_norm == 'batch':
                mods.append(nn.BatchNorm1d(dim_list[i + 1]))
            if activation == 'relu':
                mods.append(nn.ReLU())
            elif activation == 'leakyrelu':
                mods.append(nn.LeakyReLU())
        if dropout > 0:
            mods.append(nn.Dropout(p=dropout))
    return nn.Sequential(*mods)


def get_activation(name):
    """layers.py:34-47 -- ALWAYS LeakyReLU (line 40 overwrites the name); only '-slope' matters."""
    slope = 0.01
    if name.lower().startswith('leakyrelu') and '-' in name:
        slope = float(name.split('-')[1])
    return nn.LeakyReLU(negative_slope=slope)


def get_normalization_2d(channels, normalization):
    """layers.py:23-31."""
    if normalization == 'instance':
        return nn.InstanceNorm2d(channels)
    if normalization == 'batch':
        return nn.BatchNorm2d(channels)
    if normalization == 'none':
        return None
    raise ValueError('Unrecognized normalization type "%s"' % normalization)


class GlobalAvgPool(nn.Module):
    """layers.py:82-85."""

    def forward(self, x):
        return x.flatten(2).mean(dim=2)


class Interpolate(nn.Module):
    """layers.py:304-314 (only nearest x2 is used: generators.py:20)."""

    def __init__(self, size=None, scale_factor=None, mode='nearest', align_corners=None):
        super().__init__()
        self.size, self.scale_factor, self.mode, self.align_corners = size, scale_factor, mode, align_corners

    def forward(self, x):
        return F.interpolate(x, size=self.size, scale_factor=self.scale_factor, mode=self.mode,
                             align_corners=self.align_corners)


class ResidualBlock(nn.Module):
    """layers.py:84-118: x + branch(x), branch = [norm, act, conv, norm, act, conv] ('same' padding).  The reference runs the
    branch twice and keeps the second result (:114-115) -- visible in BatchNorm's buffers, which advance twice -- and cannot run
    with padding 0 (its shortcut slice is empty, :111-113)."""

    def __init__(self, channels, normalization='batch', activation='relu', padding='same', kernel_size=3, init='default'):
        super().__init__()
        pad = 0 if padding == 'valid' else (kernel_size - 1) // 2
        if pad == 0:
            raise ValueError('ResidualBlock without padding: the reference shortcut is empty')
        branch = []
        for _ in range(2):
            nrm = get_normalization_2d(channels, normalization)
            branch += ([nrm] if nrm is not None else []) + [get_activation(activation),
                                                            nn.Conv2d(channels, channels, kernel_size, padding=pad)]
        self.net = nn.Sequential(*branch)

    def forward(self, x):
        self.net(x)
        return x + self.net(x)


def build_cnn(arch, normalization='batch', activation='relu', padding='same', pooling='max', init='default'):
    """layers.py:128-212: layer kinds I, C, R, U, P."""
    specs = arch.split(',') if isinstance(arch, str) else list(arch)
    chans = 3
    if specs and specs[0][0] == 'I':
        chans = int(specs[0][1:])
        specs = specs[1:]
    mods, seen_conv = [], False
    for s in specs:
        if s[0] == 'C':
            if seen_conv:
                nrm = get_normalization_2d(chans, normalization)
                if nrm is not None:
                    mods.append(nrm)
                mods.append(get_activation(activation))
            seen_conv = True
            v = [int(t) for t in s[1:].split('-')]
            k, oc, st = (v + [1])[:3] if len(v) == 2 else v
            pad = 0 if padding == 'valid' else (k - 1) // 2
            mods.append(nn.Conv2d(chans, oc, kernel_size=k, padding=pad, stride=st))
            chans = oc
        elif s[0] == 'R':          # layers.py:172-177
            mods.append(ResidualBlock(chans, normalization=normalization if seen_conv else 'none', activation=activation,
                                      padding=padding))
            seen_conv = True
        elif s[0] == 'U':
            mods.append(Interpolate(scale_factor=int(s[1:]), mode='nearest'))
        elif s[0] == 'P':
            f = int(s[1:])
            mods.append(nn.MaxPool2d(f, f) if pooling == 'max' else nn.AvgPool2d(f, f))
        else:
            raise ValueError('Invalid layer "%s"' % s)
    return nn.Sequential(*mods), chans


class ResnetBlock(nn.Module):
    """layers.py:234-273 with padding_type='reflect', no dropout: x + [pad,conv,norm,relu,pad,conv,norm](x)."""

    def __init__(self, dim, padding_type, norm_layer, activation=None, use_dropout=False):
        super().__init__()
        assert padding_type == 'reflect' and not use_dropout
        activation = nn.ReLU(True) if activation is None else activation
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), norm_layer(dim), activation,
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), norm_layer(dim))

    def forward(self, x):
        return x + self.conv_block(x)


def get_norm_layer(norm_type='instance'):
    """layers.py:292-301."""
    if norm_type == 'instance':
        return lambda c: nn.InstanceNorm2d(c, affine=False)
    if norm_type == 'batch':
        return lambda c: nn.BatchNorm2d(c, affine=True)
    raise NotImplementedError(norm_type)


# ------------------------------------------------------------------------------------------
# graph.py
# ------------------------------------------------------------------------------------------

def _kaiming(seq):
    for m in seq:
        if isinstance(m, nn.Linear):
            nn.init.kaiming_normal_(m.weight)          # graph.py:27-30


class GraphTripleConv(nn.Module):
    """graph.py:33-122."""

    def __init__(self, input_dim, attributes_dim=0, output_dim=None, hidden_dim=512, pooling='avg',
                 mlp_normalization='none'):
        super().__init__()
        output_dim = input_dim if output_dim is None else output_dim
        assert pooling in ('sum', 'avg'), 'Invalid pooling "%s"' % pooling
        self.input_dim, self.output_dim, self.hidden_dim, self.pooling = input_dim, output_dim, hidden_dim, pooling
        self.net1 = build_mlp([3 * input_dim + 2 * attributes_dim, hidden_dim, 2 * hidden_dim + output_dim],
                              batch_norm=mlp_normalization)
        self.net2 = build_mlp([hidden_dim, hidden_dim, output_dim], batch_norm=mlp_normalization)
        _kaiming(self.net1)
        _kaiming(self.net2)

    def forward(self, obj_vecs, pred_vecs, edges):
        O, T, H, Dout = obj_vecs.size(0), pred_vecs.size(0), self.hidden_dim, self.output_dim
        s, o = edges[:, 0].contiguous(), edges[:, 1].contiguous()
        t_in = torch.cat([obj_vecs[s], pred_vecs, obj_vecs[o]], dim=1)            # graph.py:79-84
        t_out = self.net1(t_in)
        new_s, new_p, new_o = t_out[:, :H], t_out[:, H:H + Dout], t_out[:, H + Dout:2 * H + Dout]
        pooled = pool_triples(new_s, new_o, s, o, O, self.pooling)
        return self.net2(pooled), new_p


def pool_triples(new_s, new_o, s, o, O, pooling='avg'):
    """graph.py:94-116: scatter_add s-pass THEN o-pass, t ascending (CPU scatter_add is sequential
    along dim 0: SURVEY appendix D.1); degree count clamp(min=1); true division."""
    H = new_s.size(1)
    pooled = new_s.new_zeros(O, H)
    pooled = pooled.index_add(0, s, new_s)
    pooled = pooled.index_add(0, o, new_o)
    if pooling == 'avg':
        ones = new_s.new_ones(s.numel())
        cnt = new_s.new_zeros(O).index_add(0, s, ones).index_add(0, o, ones).clamp(min=1)
        pooled = pooled / cnt.view(-1, 1)
    return pooled


class GraphTripleConvNet(nn.Module):
    """graph.py:125-147."""

    def __init__(self, input_dim, num_layers=5, hidden_dim=512, pooling='avg', mlp_normalization='none'):
        super().__init__()
        self.num_layers = num_layers
        self.gconvs = nn.ModuleList([
            GraphTripleConv(input_dim, hidden_dim=hidden_dim, pooling=pooling, mlp_normalization=mlp_normalization)
            for _ in range(num_layers)])

    def forward(self, obj_vecs, pred_vecs, edges):
        for g in self.gconvs:
            obj_vecs, pred_vecs = g(obj_vecs, pred_vecs, edges)
        return obj_vecs, pred_vecs


# ------------------------------------------------------------------------------------------
# generators.py
# ------------------------------------------------------------------------------------------

def conv_weights_init(m):
    """generators.py:7-13 / discriminators.py:57-63: N(0,0.02) for every class whose name contains 'Conv'."""
    name = m.__class__.__name__
    if 'Conv' in name:
        m.weight.data.normal_(0.0, 0.02)
    elif 'BatchNorm2d' in name:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def mask_net(dim, mask_size):
    """generators.py:16-28."""
    mods, cur = [], 1
    while cur < mask_size:
        mods += [Interpolate(scale_factor=2, mode='nearest'), nn.Conv2d(dim, dim, 3, padding=1),
                 nn.BatchNorm2d(dim), nn.ReLU()]
        cur *= 2
    if cur != mask_size:
        raise ValueError('Mask size must be a power of 2')
    mods.append(nn.Conv2d(dim, 1, 1))
    return nn.Sequential(*mods)


class AppearanceEncoder(nn.Module):
    """generators.py:31-48."""

    def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', vecs_size=1024,
                 pooling='avg'):
        super().__init__()
        self.vocab = vocab
        cnn, ch = build_cnn(arch=arch, normalization=normalization, activation=activation, pooling=pooling,
                            padding=padding)
        self.cnn = nn.Sequential(cnn, GlobalAvgPool(), nn.Linear(ch, vecs_size))

    def forward(self, crops):
        return self.cnn(crops)


class GlobalGenerator(nn.Module):
    """generators.py:62-91 (pix2pixHD global generator)."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=nn.BatchNorm2d,
                 padding_type='reflect'):
        super().__init__()
        act = nn.ReLU(True)
        m = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, 7), norm_layer(ngf), act]
        c = ngf
        for _ in range(n_downsampling):
            m += [nn.Conv2d(c, 2 * c, 3, stride=2, padding=1), norm_layer(2 * c), act]
            c *= 2
        for _ in range(n_blocks):
            m.append(ResnetBlock(c, padding_type=padding_type, activation=act, norm_layer=norm_layer))
        for _ in range(n_downsampling):
            m += [nn.ConvTranspose2d(c, c // 2, 3, stride=2, padding=1, output_padding=1), norm_layer(c // 2), act]
            c //= 2
        m += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Tanh()]
        self.model = nn.Sequential(*m)

    def forward(self, x):
        return self.model(x)


def define_G(input_nc, output_nc, ngf, n_downsample_global=3, n_blocks_global=9, norm='instance'):
    """generators.py:51-57 minus the CUDA assert/.cuda()."""
    g = GlobalGenerator(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global, get_norm_layer(norm))
    g.apply(conv_weights_init)
    return g


# ------------------------------------------------------------------------------------------
# discriminators.py
# ------------------------------------------------------------------------------------------

class AcDiscriminator(nn.Module):
    """discriminators.py:10-36."""

    def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', pooling='avg'):
        super().__init__()
        self.vocab = vocab
        cnn, D = build_cnn(arch=arch, normalization=normalization, activation=activation, pooling=pooling,
                           padding=padding)
        self.cnn = nn.Sequential(cnn, GlobalAvgPool(), nn.Linear(D, 1024))
        self.real_classifier = nn.Linear(1024, 1)
        self.obj_classifier = nn.Linear(1024, len(vocab['object_to_idx']))

    def forward(self, x, y):
        if x.dim() == 3:
            x = x[:, None]
        v = self.cnn(x)
        return self.real_classifier(v), F.cross_entropy(self.obj_classifier(v), y)


class AcCropDiscriminator(nn.Module):
    """discriminators.py:39-51."""

    def __init__(self, vocab, arch, normalization='none', activation='relu', object_size=64, padding='same',
                 pooling='avg'):
        super().__init__()
        self.vocab = vocab
        self.discriminator = AcDiscriminator(vocab, arch, normalization, activation, padding, pooling)
        self.object_size = object_size

    def forward(self, imgs, objs, boxes, obj_to_img):
        crops = crop_bbox_batch(imgs, boxes, obj_to_img, self.object_size)
        real_scores, ac_loss = self.discriminator(crops, objs)
        return real_scores, ac_loss, crops


def _patch_layers(input_nc, ndf, n_layers, norm_layer, kw, use_sigmoid, extra_in=0):
    """Shared layer plan of NLayerDiscriminator (kw=4, discriminators.py:206-245) and
    NLayerMaskDiscriminator (kw=3, extra one-hot channels before the second-last conv, :128-169)."""
    pad = (kw - 1 + 1) // 2                       # int(ceil((kw-1)/2))
    seq = [[nn.Conv2d(input_nc, ndf, kw, stride=2, padding=pad), nn.LeakyReLU(0.2, True)]]
    nf = ndf
    for _ in range(1, n_layers):
        prev, nf = nf, min(nf * 2, 512)
        seq.append([nn.Conv2d(prev, nf, kw, stride=2, padding=pad), norm_layer(nf), nn.LeakyReLU(0.2, True)])
    prev, nf = nf, min(nf * 2, 512)
    seq.append([nn.Conv2d(prev + extra_in, nf, kw, stride=1, padding=pad), norm_layer(nf), nn.LeakyReLU(0.2, True)])
    seq.append([nn.Conv2d(nf, 1, kw, stride=1, padding=pad)])
    if use_sigmoid:
        seq.append([nn.Sigmoid()])
    return [nn.Sequential(*s) for s in seq]


class MultiscaleDiscriminator(nn.Module):
    """discriminators.py:172-202: num_D PatchGANs, finest scale first, all intermediate features returned."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False, num_D=3):
        super().__init__()
        self.num_D, self.n_layers = num_D, n_layers
        for i in range(num_D):
            for j, blk in enumerate(_patch_layers(input_nc, ndf, n_layers, norm_layer, 4, use_sigmoid)[:n_layers + 2]):
                setattr(self, 'scale%d_layer%d' % (i, j), blk)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, x):
        out = []
        for i in range(self.num_D):
            feats, h = [], x
            for j in range(self.n_layers + 2):
                h = getattr(self, 'scale%d_layer%d' % (self.num_D - 1 - i, j))(h)
                feats.append(h)
            out.append(feats)
            if i != self.num_D - 1:
                x = self.downsample(x)
        return out


class MultiscaleMaskDiscriminator(nn.Module):
    """discriminators.py:87-124: as above with k=3 and the one-hot class map concatenated before layer[-2]."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False, num_D=3,
                 num_objects=None):
        super().__init__()
        self.num_D, self.n_layers = num_D, n_layers
        for i in range(num_D):
            blks = _patch_layers(input_nc, ndf, n_layers, norm_layer, 3, use_sigmoid, extra_in=num_objects)
            for j, blk in enumerate(blks[:n_layers + 2]):
                setattr(self, 'scale%d_layer%d' % (i, j), blk)
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, x, cond):
        out = []
        L = self.n_layers + 2
        for i in range(self.num_D):
            layers = [getattr(self, 'scale%d_layer%d' % (self.num_D - 1 - i, j)) for j in range(L)]
            feats, h = [], x
            for j in range(L - 2):
                h = layers[j](h)
                feats.append(h)
            a, _, c, d = h.shape
            h = layers[L - 2](torch.cat([h, cond.view(a, -1, 1, 1).expand(-1, -1, c, d)], dim=1))
            feats.append(h)
            h = layers[L - 1](h)
            feats.append(h)
            out.append(feats)
            if i != self.num_D - 1:
                x = self.downsample(x)
        return out


def define_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1):
    d = MultiscaleDiscriminator(input_nc, ndf, n_layers_D, get_norm_layer(norm), use_sigmoid, num_D)
    d.apply(conv_weights_init)
    return d


def define_mask_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1, num_objects=None):
    d = MultiscaleMaskDiscriminator(input_nc, ndf, n_layers_D, get_norm_layer(norm), use_sigmoid, num_D, num_objects)
    d.apply(conv_weights_init)
    return d


# ------------------------------------------------------------------------------------------
# losses.py
# ------------------------------------------------------------------------------------------

def bce_loss(x, target):
    """losses.py:26-44: max(x,0) - x*t + log(1+exp(-|x|)), mean."""
    return (x.clamp(min=0) - x * target + (1 + (-x.abs()).exp()).log()).mean()


def gan_g_loss(scores_fake):
    """losses.py:59-70."""
    s = scores_fake.reshape(-1)
    return bce_loss(s, torch.ones_like(s))


def gan_d_loss(scores_real, scores_fake):
    """losses.py:73-90."""
    assert scores_real.size() == scores_fake.size()
    r, f = scores_real.reshape(-1), scores_fake.reshape(-1)
    return bce_loss(r, torch.ones_like(r)) + bce_loss(f, torch.zeros_like(f))


def wgan_g_loss(scores_fake):
    """losses.py:93-101."""
    return -scores_fake.mean()


def wgan_d_loss(scores_real, scores_fake):
    """losses.py:104-112."""
    return scores_fake.mean() - scores_real.mean()


def lsgan_g_loss(scores_fake):
    """losses.py:115-119: MSE of sigmoid(scores) against 1."""
    s = scores_fake.reshape(-1)
    return F.mse_loss(s.sigmoid(), torch.ones_like(s))


def lsgan_d_loss(scores_real, scores_fake):
    """losses.py:122-132."""
    assert scores_real.size() == scores_fake.size()
    r, f = scores_real.reshape(-1), scores_fake.reshape(-1)
    return F.mse_loss(r.sigmoid(), torch.ones_like(r)) + F.mse_loss(f.sigmoid(), torch.zeros_like(f))


def get_gan_losses(gan_type):
    """losses.py:8-23 ('gan' is the type the default flags select, args.py:95)."""
    if gan_type == 'gan':
        return gan_g_loss, gan_d_loss
    if gan_type == 'wgan':
        return wgan_g_loss, wgan_d_loss
    if gan_type == 'lsgan':
        return lsgan_g_loss, lsgan_d_loss
    raise ValueError('Unrecognized GAN type "%s"' % gan_type)


class GANLoss(nn.Module):
    """losses.py:135-175: MSE (LSGAN, default) or nn.BCELoss against a constant 1/0 target, SUMMED over scales
    (:166-172)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=None):
        super().__init__()
        self.use_lsgan = use_lsgan
        self.real_label, self.fake_label = target_real_label, target_fake_label

    def _one(self, pred, target_is_real):
        t = self.real_label if target_is_real else self.fake_label
        if self.use_lsgan:
            return F.mse_loss(pred, torch.full_like(pred, t))
        return F.binary_cross_entropy(pred, torch.full_like(pred, t))

    def __call__(self, preds, target_is_real):
        if isinstance(preds[0], list):
            total = 0
            for p in preds:
                total = total + self._one(p[-1], target_is_real)
            return total
        return self._one(preds[-1], target_is_real)


def features_loss(pred_fake, pred_real):
    """trainer.py:331-340: sum_i sum_{j<last} (1/num_D)(4/len) L1(fake_ij, real_ij.detach())."""
    num_d = len(pred_fake)
    w = (1.0 / num_d) * (4.0 / len(pred_fake[0]))
    total = 0
    for i in range(num_d):
        for j in range(len(pred_fake[i]) - 1):
            total = total + w * F.l1_loss(pred_fake[i][j], pred_real[i][j].detach())
    return total


# VGG feature matching (losses.py:179-224).  The reference slices torchvision's vgg19(pretrained=True).features; torchvision
# (and its ImageNet weights) are absent here, so the ARCHITECTURE below restates torchvision's published configuration 'E'
# (conv3x3 pad 1 + ReLU(inplace), MaxPool2d(2, 2); features indices 0..29) and the weights are whatever the caller loads.
# tools/make_golden.py runs the reference's own Vgg19 / VGGLoss classes on a torchvision shim with this configuration,
# which pins the reference-side part (slice boundaries, L1 weights, detach) -- the ImageNet weights stay unpinned.
VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')


def vgg19_features():
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


class Vgg19(nn.Module):
    """losses.py:179-209."""

    def __init__(self, requires_grad=False):
        super().__init__()
        feats = vgg19_features()
        bounds = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]
        for k, (lo, hi) in enumerate(bounds):
            seq = nn.Sequential()
            for x in range(lo, hi):
                seq.add_module(str(x), feats[x])
            setattr(self, 'slice%d' % (k + 1), seq)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, X):
        h1 = self.slice1(X)
        h2 = self.slice2(h1)
        h3 = self.slice3(h2)
        h4 = self.slice4(h3)
        h5 = self.slice5(h4)
        return [h1, h2, h3, h4, h5]


class VGGLoss(nn.Module):
    """losses.py:212-224."""

    def __init__(self):
        super().__init__()
        self.vgg = Vgg19()
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        x_vgg, y_vgg = self.vgg(x), self.vgg(y)
        loss = 0
        for i in range(len(x_vgg)):
            loss = loss + self.weights[i] * F.l1_loss(x_vgg[i], y_vgg[i].detach())
        return loss


# ------------------------------------------------------------------------------------------
# utils.py
# ------------------------------------------------------------------------------------------

class LossManager(object):
    """utils.py:43-59."""

    def __init__(self):
        self.total_loss = None
        self.all_losses = {}

    def add_loss(self, loss, name, weight=1.0, use_loss=True):
        cur = loss * weight
        if use_loss:
            self.total_loss = cur if self.total_loss is None else self.total_loss + cur
        self.all_losses[name] = cur.item()

    def items(self):
        return self.all_losses.items()


class VectorPool:
    """utils.py:62-90: per-class replay pool on the host; Python ``random.randint`` draws."""

    def __init__(self, pool_size):
        self.pool_size = pool_size
        self.vectors = {}

    def query(self, objs, vectors):
        if self.pool_size == 0:
            return vectors
        out = []
        for cls, vec in zip(objs.tolist(), vectors):
            vec = vec.detach().clone()
            pool = self.vectors.setdefault(cls, [])
            n = len(pool)
            if n == 0:
                out.append(vec)
                pool.append(vec)
            elif n < self.pool_size:
                r = random.randint(0, n - 1)
                pool.append(vec)
                out.append(pool[r])
            else:
                r = random.randint(0, n - 1)
                out.append(pool[r])
                pool[r] = vec
        return torch.stack(out).to(vectors.device)


# ------------------------------------------------------------------------------------------
# model.py
# ------------------------------------------------------------------------------------------

class Model(nn.Module):
    """model.py:12-172 (training branch)."""

    def __init__(self, vocab, image_size=(64, 64), embedding_dim=128, gconv_dim=128, gconv_hidden_dim=512,
                 gconv_pooling='avg', gconv_num_layers=5, mask_size=32, mlp_normalization='none',
                 appearance_normalization='', activation='', n_downsample_global=4, box_dim=128,
                 use_attributes=False, box_noise_dim=64, mask_noise_dim=64, pool_size=100, rep_size=32,
                 ngf=64, n_blocks_global=9):
        super().__init__()
        self.vocab, self.image_size, self.use_attributes = vocab, image_size, use_attributes
        self.mask_noise_dim, self.object_size = mask_noise_dim, 64
        self.fake_pool = VectorPool(pool_size)
        self.num_objs = len(vocab['object_to_idx'])
        self.num_preds = len(vocab['pred_idx_to_name'])
        self.obj_embeddings = nn.Embedding(self.num_objs, embedding_dim)
        self.pred_embeddings = nn.Embedding(self.num_preds, embedding_dim)
        adim = vocab['num_attributes'] if use_attributes else 0
        assert gconv_num_layers >= 1
        self.gconv = GraphTripleConv(embedding_dim, attributes_dim=adim, output_dim=gconv_dim,
                                     hidden_dim=gconv_hidden_dim, pooling=gconv_pooling,
                                     mlp_normalization=mlp_normalization)
        self.gconv_net = None
        if gconv_num_layers > 1:
            self.gconv_net = GraphTripleConvNet(gconv_dim, num_layers=gconv_num_layers - 1,
                                                hidden_dim=gconv_hidden_dim, pooling=gconv_pooling,
                                                mlp_normalization=mlp_normalization)
        self.box_net = build_mlp([box_dim, gconv_hidden_dim, 4], batch_norm=mlp_normalization)
        self.g_mask_dim = gconv_dim + mask_noise_dim
        self.mask_net = mask_net(self.g_mask_dim, mask_size)
        self.repr_net = build_mlp([self.g_mask_dim, 64, rep_size], batch_norm=mlp_normalization)
        self.image_encoder = AppearanceEncoder(vocab=vocab, arch='C4-64-2,C4-128-2,C4-256-2',
                                               normalization=appearance_normalization, activation=activation,
                                               padding='valid', vecs_size=self.g_mask_dim)
        self.layout_to_image = define_G(self.num_objs + rep_size, 3, ngf, n_downsample_global, n_blocks_global,
                                        'instance')
        self.noise_override = None        # test hook: (1, mask_noise_dim) row injected instead of randn

    def scene_graph_to_vectors(self, objs, triples, attributes):
        s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
        edges = torch.stack([s, o], dim=1)
        obj_vecs = self.obj_embeddings(objs)
        pred_vecs = self.pred_embeddings(p)
        if self.use_attributes:
            obj_vecs = torch.cat([obj_vecs, attributes], dim=1)
        obj_vecs, pred_vecs = self.gconv(obj_vecs, pred_vecs, edges)
        if self.gconv_net is not None:
            obj_vecs, pred_vecs = self.gconv_net(obj_vecs, pred_vecs, edges)
        return obj_vecs, pred_vecs

    def create_components_vecs(self, imgs, boxes, obj_to_img, objs, obj_vecs, features=None):
        O = objs.size(0)
        noise = self.noise_override if self.noise_override is not None else \
            torch.randn((1, self.mask_noise_dim), dtype=obj_vecs.dtype, device=obj_vecs.device)
        mask_vecs = torch.cat([obj_vecs, noise.repeat(O, 1)], dim=1)
        if features is None:
            crops = crop_bbox_batch(imgs, boxes, obj_to_img, self.object_size)
            obj_repr = self.repr_net(self.image_encoder(crops))
        else:                                               # inference only (model.py:158-163)
            obj_repr = self.repr_net(mask_vecs)
            for ind, feature in enumerate(features):
                if feature is not None:
                    obj_repr[ind, :] = feature
        one_hot = torch.zeros(O, self.num_objs, dtype=obj_repr.dtype).scatter_(1, objs.view(-1, 1), 1.0)
        layout_vecs = torch.cat([one_hot, obj_repr], dim=1)
        wrong = self.fake_pool.query(objs, obj_repr)
        return obj_vecs, mask_vecs, layout_vecs, torch.cat([one_hot, wrong], dim=1)

    def forward(self, gt_imgs, objs, triples, obj_to_img, boxes_gt=None, masks_gt=None, attributes=None,
                test_mode=False, use_gt_box=False, features=None):
        O = objs.size(0)
        obj_vecs, _ = self.scene_graph_to_vectors(objs, triples, attributes)
        box_vecs, mask_vecs, layout_vecs, wrong_vecs = self.create_components_vecs(
            gt_imgs, boxes_gt, obj_to_img, objs, obj_vecs, features)
        boxes_pred = self.box_net(box_vecs)
        masks_pred = self.mask_net(mask_vecs.view(O, -1, 1, 1)).squeeze(1).sigmoid()
        H, W = self.image_size
        if test_mode:                                       # model.py:111-117
            boxes = boxes_gt if use_gt_box else boxes_pred
            masks = masks_gt if masks_gt is not None else masks_pred
            pred_layout = masks_to_layout(layout_vecs, boxes, masks, obj_to_img, H, W, test_mode=True)
            return self.layout_to_image(pred_layout), boxes_pred, masks_pred, None, pred_layout, None
        gt_layout = masks_to_layout(layout_vecs, boxes_gt, masks_gt, obj_to_img, H, W)
        pred_layout = masks_to_layout(layout_vecs, boxes_gt, masks_pred, obj_to_img, H, W)
        wrong_layout = masks_to_layout(wrong_vecs, boxes_gt, masks_gt, obj_to_img, H, W)
        imgs_pred = self.layout_to_image(gt_layout)
        return imgs_pred, boxes_pred, masks_pred, gt_layout, pred_layout, wrong_layout


# ------------------------------------------------------------------------------------------
# trainer.py / train.py:190-215
# ------------------------------------------------------------------------------------------

class Trainer:
    """trainer.py:15-134,205-340 without logging/checkpoint glue.  ``criterionVGG`` (trainer.py:57) is built when
    ``--vgg_features_weight > 0``; its weights are the caller's business (see Vgg19 above)."""

    def __init__(self, args, vocab, model_extra=None):
        self.args, self.vocab = args, vocab
        self.num_obj = len(vocab['object_to_idx'])
        self.gan_g_loss, self.gan_d_loss = get_gan_losses(args.gan_loss_type)
        mk = dict(vocab=vocab, image_size=args.image_size, embedding_dim=args.embedding_dim,
                  gconv_dim=args.gconv_dim, gconv_hidden_dim=args.gconv_hidden_dim,
                  gconv_num_layers=args.gconv_num_layers, mlp_normalization=args.mlp_normalization,
                  appearance_normalization=args.appearance_normalization, activation=args.activation,
                  mask_size=args.mask_size, n_downsample_global=args.n_downsample_global, box_dim=args.box_dim,
                  use_attributes=args.use_attributes, box_noise_dim=args.box_noise_dim,
                  mask_noise_dim=args.mask_noise_dim, pool_size=args.pool_size, rep_size=args.rep_size)
        mk.update(model_extra or {})
        self.model = Model(**mk)
        self.criterionVGG = VGGLoss() if args.vgg_features_weight > 0 else None
        self.criterionGAN = GANLoss(use_lsgan=not args.no_lsgan)
        adam = lambda m, lr: torch.optim.Adam(m.parameters(), lr=lr, betas=(args.beta1, 0.999))
        self.optimizer = adam(self.model, args.learning_rate)
        self.netD = define_D(self.num_obj + args.rep_size + args.output_nc, args.ndf, args.n_layers_D,
                             args.norm_D, args.no_lsgan, args.num_D)
        self.optimizer_d_img = adam(self.netD, args.learning_rate)
        self.obj_discriminator = AcCropDiscriminator(vocab=vocab, arch=args.d_obj_arch,
                                                     normalization=args.d_normalization,
                                                     activation=args.d_activation, padding=args.d_padding,
                                                     object_size=args.crop_size)
        self.optimizer_d_obj = adam(self.obj_discriminator, args.learning_rate)
        self.mask_discriminator = define_mask_D(1, args.ndf_mask, args.n_layers_D_mask, args.norm_D_mask,
                                                args.no_lsgan, args.num_D_mask, self.num_obj)
        self.optimizer_d_mask = adam(self.mask_discriminator, args.mask_learning_rate)

    def _one_hot(self, objs, like):
        return torch.zeros(objs.size(0), self.num_obj, dtype=like.dtype).scatter_(1, objs.view(-1, 1), 1.0)

    def train_generator(self, imgs, imgs_pred, masks, masks_pred, layout, objs, boxes, boxes_pred, obj_to_img,
                        use_gt):
        a = self.args
        L = self.generator_losses = LossManager()
        if use_gt:
            if a.l1_pixel_loss_weight > 0:
                L.add_loss(F.l1_loss(imgs_pred, imgs), 'L1_pixel_loss', a.l1_pixel_loss_weight)
            L.add_loss(F.mse_loss(boxes_pred, boxes), 'bbox_pred', a.bbox_pred_loss_weight)
        if self.criterionVGG is not None:                                        # trainer.py:218-221
            L.add_loss(self.criterionVGG(imgs_pred, imgs), 'g_vgg', a.vgg_features_weight)
        scores_fake, ac_loss, _ = self.obj_discriminator(imgs_pred, objs, boxes, obj_to_img)
        L.add_loss(ac_loss, 'ac_loss', a.ac_loss_weight)
        L.add_loss(self.gan_g_loss(scores_fake), 'g_gan_obj_loss', a.d_obj_weight)
        one_hot = self._one_hot(objs, masks_pred)
        mfake = self.mask_discriminator(masks_pred.unsqueeze(1), one_hot)
        L.add_loss(self.criterionGAN(mfake, True), 'g_gan_mask_obj_loss', a.d_mask_weight)
        if a.d_mask_features_weight > 0:
            mreal = self.mask_discriminator(masks.to(masks_pred.dtype).unsqueeze(1), one_hot)
            L.add_loss(features_loss(mfake, mreal), 'g_mask_features_loss', a.d_mask_features_weight)
        pred_real = self.netD(torch.cat((layout, imgs), dim=1))
        pred_fake = self.netD(torch.cat((layout.detach(), imgs_pred), dim=1))
        L.add_loss(self.criterionGAN(pred_fake, True), 'g_gan_img_loss', a.d_img_weight)
        if a.d_img_features_weight > 0:
            L.add_loss(features_loss(pred_fake, pred_real), 'g_gan_features_loss_img', a.d_img_features_weight)
        L.all_losses['total_loss'] = L.total_loss.item()
        self.optimizer.zero_grad()
        L.total_loss.backward()
        self.optimizer.step()

    def train_obj_discriminator(self, imgs, imgs_pred, objs, boxes, boxes_pred, obj_to_img):
        L = self.d_obj_losses = LossManager()
        sf, ac_fake, _ = self.obj_discriminator(imgs_pred, objs, boxes_pred, obj_to_img)
        sr, ac_real, _ = self.obj_discriminator(imgs, objs, boxes, obj_to_img)
        L.add_loss(self.gan_d_loss(sr, sf), 'd_obj_gan_loss', 0.5)
        L.add_loss(ac_real, 'd_ac_loss_real')
        L.add_loss(ac_fake, 'd_ac_loss_fake')
        self.optimizer_d_obj.zero_grad()
        L.total_loss.backward()
        self.optimizer_d_obj.step()

    def train_mask_discriminator(self, masks, masks_pred, objs):
        L = self.d_mask_losses = LossManager()
        one_hot = self._one_hot(objs, masks_pred)
        sf = self.mask_discriminator(masks_pred.unsqueeze(1), one_hot)
        sr = self.mask_discriminator(masks.to(masks_pred.dtype).unsqueeze(1), one_hot)
        L.add_loss(self.criterionGAN(sf, False), 'fake_loss', 0.5)
        L.add_loss(self.criterionGAN(sr, True), 'real_loss', 0.5)
        self.optimizer_d_mask.zero_grad()
        L.total_loss.backward()
        self.optimizer_d_mask.step()

    def train_image_discriminator(self, imgs, imgs_pred, layout, layout_wrong):
        L = self.d_img_losses = LossManager()
        L.add_loss(self.criterionGAN(self.netD(torch.cat((layout, imgs_pred), 1)), False), 'fake_image_loss', 0.25)
        L.add_loss(self.criterionGAN(self.netD(torch.cat((layout_wrong, imgs), 1)), False), 'wrong_texture_loss',
                   0.25)
        L.add_loss(self.criterionGAN(self.netD(torch.cat((layout, imgs), 1)), True), 'd_img_gan_real_loss', 0.5)
        self.optimizer_d_img.zero_grad()
        L.total_loss.backward()
        self.optimizer_d_img.step()

    def step(self, batch, use_gt=True):
        """train.py:190-215: one full G+D iteration.  Returns the model outputs."""
        imgs, objs, boxes, masks, triples, obj_to_img, _, attributes = batch
        if not use_gt:
            attributes = torch.zeros_like(attributes)
        out = self.model(imgs, objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=masks, attributes=attributes)
        imgs_pred, boxes_pred, masks_pred, layout, layout_pred, layout_wrong = out
        self.train_generator(imgs, imgs_pred, masks, masks_pred, layout, objs, boxes, boxes_pred, obj_to_img, use_gt)
        self.train_mask_discriminator(masks, masks_pred.detach(), objs)
        self.train_obj_discriminator(imgs, imgs_pred.detach(), objs, boxes, boxes.detach(), obj_to_img)
        self.train_image_discriminator(imgs, imgs_pred.detach(), layout.detach(), layout_wrong.detach())
        return out
