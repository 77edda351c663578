"""Discriminators (surface of /root/reference/scene_generation/discriminators.py).

* AcDiscriminator / AcCropDiscriminator (:10-51): bilinear crop + small CNN + GAP + FC, aux classifier.
* MultiscaleDiscriminator / NLayerDiscriminator (:172-245): k4 PatchGANs returning every feature map.  The
  image discriminator is always fed cat((layout, image), 1) (trainer.py:246,250,328); ``forward`` therefore also
  accepts the pair and folds the concat into the first conv's gather (two base pointers, nothing copied).
* MultiscaleMaskDiscriminator / NLayerMaskDiscriminator (:87-169): k3 PatchGAN with the one-hot class map
  concatenated before the second-last conv -- fed to that conv as a second gather source.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .bilinear import crop_bbox_batch
from .layers import (GlobalAvgPool, build_cnn, get_norm_layer, Conv2d, LeakyReLU, Sigmoid, Linear, BatchNorm2d,
                     AvgPool3s2, FusedSequential)


class AcDiscriminator(nn.Module):
    def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', pooling='avg'):
        super().__init__()
        self.vocab = vocab
        cnn, D = build_cnn(arch=arch, normalization=normalization, activation=activation, pooling=pooling,
                           padding=padding)
        self.cnn = FusedSequential(cnn, GlobalAvgPool(), Linear(D, 1024))
        num_objects = len(vocab['object_to_idx'])
        self.real_classifier = Linear(1024, 1)
        self.obj_classifier = Linear(1024, num_objects)

    def forward(self, x, y):
        if x.dim() == 3:
            x = x[:, None]
        vecs = self.cnn(x)
        real_scores = self.real_classifier(vecs)
        obj_scores = self.obj_classifier(vecs)
        ac_loss = ops.cross_entropy(obj_scores, y)
        return real_scores, ac_loss


class AcCropDiscriminator(nn.Module):
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


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif classname.find('BatchNorm2d') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def define_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1):
    netD = MultiscaleDiscriminator(input_nc, ndf, n_layers_D, get_norm_layer(norm_type=norm), use_sigmoid, num_D)
    netD.apply(weights_init)
    return netD


def define_mask_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1, num_objects=None):
    netD = MultiscaleMaskDiscriminator(input_nc, ndf, n_layers_D, get_norm_layer(norm_type=norm), use_sigmoid, num_D,
                                       num_objects)
    netD.apply(weights_init)
    return netD


def _patchgan_blocks(input_nc, ndf, n_layers, norm_layer, use_sigmoid, kw, extra_in=0):
    padw = int(np.ceil((kw - 1.0) / 2))
    sequence = [[Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), LeakyReLU(0.2, True)]]
    nf = ndf
    for n in range(1, n_layers):
        nf_prev, nf = nf, min(nf * 2, 512)
        sequence += [[Conv2d(nf_prev, nf, kernel_size=kw, stride=2, padding=padw), norm_layer(nf), LeakyReLU(0.2, True)]]
    nf_prev, nf = nf, min(nf * 2, 512)
    sequence += [[Conv2d(nf_prev + extra_in, nf, kernel_size=kw, stride=1, padding=padw), norm_layer(nf),
                  LeakyReLU(0.2, True)]]
    sequence += [[Conv2d(nf, 1, kernel_size=kw, stride=1, padding=padw)]]
    if use_sigmoid:
        sequence += [[Sigmoid()]]
    return [FusedSequential(*s) for s in sequence]


class NLayerDiscriminator(nn.Module):
    """PatchGAN (discriminators.py:206-245)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=BatchNorm2d, use_sigmoid=False):
        super().__init__()
        self.n_layers = n_layers
        for n, blk in enumerate(_patchgan_blocks(input_nc, ndf, n_layers, norm_layer, use_sigmoid, 4)):
            setattr(self, 'model' + str(n), blk)

    def forward(self, input):
        res = [input]
        for n in range(self.n_layers + 2):
            res.append(getattr(self, 'model' + str(n))(res[-1]))
        return res[1:]


class NLayerMaskDiscriminator(nn.Module):
    """discriminators.py:128-169."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=BatchNorm2d, use_sigmoid=False, num_objects=None):
        super().__init__()
        self.n_layers = n_layers
        for n, blk in enumerate(_patchgan_blocks(input_nc, ndf, n_layers, norm_layer, use_sigmoid, 3, num_objects)):
            setattr(self, 'model' + str(n), blk)

    def forward(self, input):
        res = [input]
        for n in range(self.n_layers + 2):
            res.append(getattr(self, 'model' + str(n))(res[-1]))
        return res[1:]


def _first_block(block, a, b):
    """conv(cat(a, b)) + LeakyReLU with the concat folded into the gather."""
    conv = block[0]
    act, slope = ops.ACT_NONE, 0.0
    rest = list(block)[1:]
    if rest and isinstance(rest[0], LeakyReLU):
        act, slope, rest = rest[0].code, rest[0].slope, rest[1:]
    h = conv(a, x2=b, act=act, slope=slope)
    for m in rest:
        h = m(h)
    return h


class MultiscaleDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=BatchNorm2d, use_sigmoid=False, num_D=3):
        super().__init__()
        self.num_D = num_D
        self.n_layers = n_layers
        for i in range(num_D):
            netD = NLayerDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid)
            for j in range(n_layers + 2):
                setattr(self, 'scale' + str(i) + '_layer' + str(j), getattr(netD, 'model' + str(j)))
        self.downsample = AvgPool3s2()

    def singleD_forward(self, model, input, input2=None):
        result = []
        h = _first_block(model[0], input, input2) if input2 is not None else model[0](input)
        result.append(h)
        for i in range(1, len(model)):
            h = model[i](h)
            result.append(h)
        return result

    def forward(self, input, input2=None):
        """``input`` = the 207-channel tensor, or (layout, image) as two tensors (concat folded)."""
        num_D = self.num_D
        result = []
        a, b = input, input2
        for i in range(num_D):
            model = [getattr(self, 'scale' + str(num_D - 1 - i) + '_layer' + str(j)) for j in range(self.n_layers + 2)]
            result.append(self.singleD_forward(model, a, b))
            if i != (num_D - 1):
                f = ops.hint(a, 'factored') if ops.FACTORED_LAYOUT else None
                if f is not None and b is not None:
                    # pooling is linear: pool the planes of the factored form.  The first conv then never reads the dense
                    # pooled layout, so it is not computed: a storage-less placeholder of the right shape carries the hint
                    planes = self.downsample(f.Z)
                    a_lo = a.new_empty((1,)).expand((a.size(0), a.size(1)) + tuple(planes.shape[2:]))
                    ops.set_hints(a_lo, factored=(f if a.requires_grad else f.detached()).with_planes(planes))
                else:
                    a_lo = ops.carry_hints(a, self.downsample(ops.ensure_dense(a)))   # pooling keeps zero channels zero
                a = a_lo
                b = self.downsample(b) if b is not None else None
        return result


class MultiscaleMaskDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=BatchNorm2d, use_sigmoid=False, num_D=3,
                 num_objects=None):
        super().__init__()
        self.num_D = num_D
        self.n_layers = n_layers
        for i in range(num_D):
            netD = NLayerMaskDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid, num_objects)
            for j in range(n_layers + 2):
                setattr(self, 'scale' + str(i) + '_layer' + str(j), getattr(netD, 'model' + str(j)))
        self.downsample = AvgPool3s2()

    def singleD_forward(self, model, input, cond):
        result = [input]
        for i in range(len(model) - 2):
            result.append(model[i](result[-1]))
        a, b, c, d = result[-1].shape
        # one-hot class map (discriminators.py:107-110): fed to the conv as a [N, classes] second gather source that
        # the kernel broadcasts over the (c, d) grid -- the expand()+cat() is never materialised
        result.append(_first_block(model[len(model) - 2], result[-1], cond.view(a, -1)))
        result.append(model[len(model) - 1](result[-1]))
        return result[1:]

    def forward(self, input, cond):
        num_D = self.num_D
        result = []
        input_downsampled = input
        for i in range(num_D):
            model = [getattr(self, 'scale' + str(num_D - 1 - i) + '_layer' + str(j)) for j in range(self.n_layers + 2)]
            result.append(self.singleD_forward(model, input_downsampled, cond))
            if i != (num_D - 1):
                input_downsampled = self.downsample(input_downsampled)
        return result
