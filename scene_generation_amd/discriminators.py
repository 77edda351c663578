"""Discriminators (surface of /root/reference/scene_generation/discriminators.py).

* AcDiscriminator / AcCropDiscriminator (:10-51): bilinear crop + small CNN + GAP + FC, aux classifier.
* MultiscaleDiscriminator / NLayerDiscriminator (:172-245): k4 PatchGANs returning every feature map.  The
  image discriminator is always fed cat((layout, image), 1) (trainer.py:246,250,328); ``forward`` therefore also
  accepts the pair and folds the concat into the first conv's gather (two base pointers, nothing copied).
* MultiscaleMaskDiscriminator / NLayerMaskDiscriminator (:87-169): k3 PatchGAN with the one-hot class map
  concatenated before the second-last conv -- fed to that conv as a second gather source.

Both PatchGAN families are one stage table (``_patchgan_plan``) turned into FusedSequential stages; the multi-scale wrappers
adopt the stages of their per-scale networks under the reference's ``scale<i>_layer<j>`` names (the state_dict contract,
SURVEY 8b) and share one pyramid walker.
"""
import torch.nn as nn

from . import ops, streams
from .bilinear import crop_bbox_batch
from .generators import weights_init           # the same pix2pixHD initialiser as discriminators.py:57-63
from .layers import (GlobalAvgPool, build_cnn, get_norm_layer, Conv2d, LeakyReLU, Sigmoid, Linear, BatchNorm2d,
                     AvgPool3s2, FusedSequential)

MAX_WIDTH = 512          # channel cap of the PatchGAN stages (discriminators.py:143,221)


# ---------------------------------------------------------------------------------------------
# object discriminator
# ---------------------------------------------------------------------------------------------

class AcDiscriminator(nn.Module):
    """crop CNN -> 1024-d feature -> (real/fake score, class logits); returns (scores, auxiliary-classifier loss)"""

    def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', pooling='avg'):
        super().__init__()
        self.vocab = vocab
        trunk, width = build_cnn(arch=arch, normalization=normalization, activation=activation, pooling=pooling,
                                 padding=padding)
        self.cnn = FusedSequential(trunk, GlobalAvgPool(), Linear(width, 1024))
        self.real_classifier = Linear(1024, 1)
        self.obj_classifier = Linear(1024, len(vocab['object_to_idx']))

    def forward(self, x, y):
        feats = self.cnn(x.unsqueeze(1) if x.dim() == 3 else x)
        return self.real_classifier(feats), ops.cross_entropy(self.obj_classifier(feats), y)


class AcCropDiscriminator(nn.Module):
    """AcDiscriminator on the bilinear crops of the objects' boxes; also hands the crops back (TensorBoard panels)"""

    def __init__(self, vocab, arch, normalization='none', activation='relu', object_size=64, padding='same',
                 pooling='avg'):
        super().__init__()
        self.vocab = vocab
        self.discriminator = AcDiscriminator(vocab, arch, normalization, activation, padding, pooling)
        self.object_size = object_size

    def forward(self, imgs, objs, boxes, obj_to_img):
        crops = crop_bbox_batch(imgs, boxes, obj_to_img, self.object_size)
        return self.discriminator(crops, objs) + (crops,)


# ---------------------------------------------------------------------------------------------
# PatchGAN stages
# ---------------------------------------------------------------------------------------------

def _patchgan_plan(input_nc, ndf, n_layers, extra_in=0):
    """(cin, cout, stride, normalised) of the n_layers + 2 conv stages: n_layers stride-2 stages whose width doubles up to
    MAX_WIDTH (the first one without normalisation), one stride-1 stage that also takes ``extra_in`` conditioning channels,
    and the 1-channel score head."""
    widths = [ndf]
    while len(widths) < n_layers:
        widths.append(min(2 * widths[-1], MAX_WIDTH))
    top = min(2 * widths[-1], MAX_WIDTH)
    plan = [(input_nc, widths[0], 2, False)]
    plan += [(a, b, 2, True) for a, b in zip(widths, widths[1:])]
    plan.append((widths[-1] + extra_in, top, 1, True))
    plan.append((top, 1, 1, None))
    return plan


class _PatchGAN(nn.Module):
    """``model0 .. model<n_layers+1>`` (+ a sigmoid stage when asked for); forward returns the output of every stage"""
    KERNEL = 4

    def __init__(self, input_nc, ndf, n_layers, norm_layer, use_sigmoid, extra_in=0):
        super().__init__()
        self.n_layers = n_layers
        k, pad = self.KERNEL, self.KERNEL // 2               # ceil((k - 1) / 2)
        stages = []
        for cin, cout, stride, normalised in _patchgan_plan(input_nc, ndf, n_layers, extra_in):
            mods = [Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad)]
            if normalised is not None:
                mods += ([norm_layer(cout)] if normalised else []) + [LeakyReLU(0.2, True)]
            stages.append(mods)
        if use_sigmoid:
            stages.append([Sigmoid()])
        for n, mods in enumerate(stages):
            setattr(self, 'model%d' % n, FusedSequential(*mods))

    def stages(self):
        return [getattr(self, 'model%d' % n) for n in range(self.n_layers + 2)]

    def forward(self, input):
        feats, h = [], input
        for stage in self.stages():
            h = stage(h)
            feats.append(h)
        return feats


class NLayerDiscriminator(_PatchGAN):
    """4x4 PatchGAN (discriminators.py:206-245)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=BatchNorm2d, use_sigmoid=False):
        super().__init__(input_nc, ndf, n_layers, norm_layer, use_sigmoid)


class NLayerMaskDiscriminator(_PatchGAN):
    """3x3 PatchGAN whose second-last conv also sees ``num_objects`` conditioning channels (discriminators.py:128-169)."""
    KERNEL = 3

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=BatchNorm2d, use_sigmoid=False, num_objects=None):
        super().__init__(input_nc, ndf, n_layers, norm_layer, use_sigmoid, extra_in=num_objects or 0)


def _run_two_source(stage, a, b):
    """``stage`` on the channel concat of (a, b) with the concat folded into its conv's gather: ``b`` is a second NCHW tensor
    or an [N, C2] row that the kernel broadcasts over the grid; a LeakyReLU right behind the conv joins its epilogue."""
    mods = list(stage)
    fused = len(mods) > 1 and isinstance(mods[1], LeakyReLU)
    h = mods[0](a, x2=b, act=mods[1].code if fused else ops.ACT_NONE, slope=mods[1].slope if fused else 0.0)
    for m in mods[2 if fused else 1:]:
        h = m(h)
    return h


class _ScalePyramid(nn.Module):
    """num_D PatchGANs, the i-th on the input average-pooled i times; stages registered as ``scale<i>_layer<j>``; the results
    are listed finest scale first, which is the network with the HIGHEST index (discriminators.py:192-202)."""

    def _adopt(self, num_D, n_layers, make_net):
        self.num_D, self.n_layers = num_D, n_layers
        for i in range(num_D):
            for j, stage in enumerate(make_net().stages()):
                setattr(self, 'scale%d_layer%d' % (i, j), stage)
        self.downsample = AvgPool3s2()

    def _scales(self):
        for level in range(self.num_D):
            i = self.num_D - 1 - level
            yield level + 1 < self.num_D, [getattr(self, 'scale%d_layer%d' % (i, j)) for j in range(self.n_layers + 2)]


def _chain(stages, h, feats):
    for stage in stages:
        h = stage(h)
        feats.append(h)
    return feats


class MultiscaleDiscriminator(_ScalePyramid):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=BatchNorm2d, use_sigmoid=False, num_D=3):
        super().__init__()
        self._adopt(num_D, n_layers, lambda: NLayerDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid))

    def singleD_forward(self, model, input, input2=None):
        first = model[0](input) if input2 is None else _run_two_source(model[0], input, input2)
        return _chain(model[1:], first, [first])

    def _pool_label(self, a, has_image):
        """the label map one pyramid level down"""
        f = ops.hint(a, 'factored') if ops.FACTORED_LAYOUT else None
        if f is None or not has_image:
            return ops.carry_hints(a, self.downsample(ops.ensure_dense(a)))     # pooling keeps all-zero channels all-zero
        # pooling is linear: pool the planes of the factored form.  The first conv then never reads the dense pooled
        # layout, so it is not computed: a storage-less placeholder of the right shape carries the hint
        planes = self.downsample(f.Z)
        ghost = a.new_empty((1,)).expand((a.size(0), a.size(1)) + tuple(planes.shape[2:]))
        return ops.set_hints(ghost, factored=(f if a.requires_grad else f.detached()).with_planes(planes))

    def forward(self, input, input2=None):
        """``input`` = the 207-channel tensor, or (layout, image) as two tensors (concat folded).  The scales only share
        the (pooled) input: the pyramid is built first, then every scale runs on its own stream (streams.fork)."""
        levels, a, b = [], input, input2
        for more, stages in self._scales():
            levels.append((stages, a, b))
            if more:
                a, b = self._pool_label(a, b is not None), (None if b is None else self.downsample(b))
        out = [None] * len(levels)
        with streams.fork(input.device, 'imgD') as f:
            for i, (stages, a, b) in enumerate(levels):
                with f.branch(i, reads=(a, b)):
                    out[i] = self.singleD_forward(stages, a, b)
                    f.produced(out[i])
        return out


class MultiscaleMaskDiscriminator(_ScalePyramid):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=BatchNorm2d, use_sigmoid=False, num_D=3,
                 num_objects=None):
        super().__init__()
        self._adopt(num_D, n_layers,
                    lambda: NLayerMaskDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid, num_objects))

    def singleD_forward(self, model, input, cond):
        feats = _chain(model[:-2], input, [])
        h = feats[-1] if feats else input
        # one-hot class map (discriminators.py:107-110): handed to the conv as an [N, classes] second gather source that the
        # kernel broadcasts over the grid -- the expand() + cat() is never materialised
        feats.append(_run_two_source(model[-2], h, cond.reshape(h.size(0), -1)))
        feats.append(model[-1](feats[-1]))
        return feats

    def forward(self, input, cond):
        levels, h = [], input
        for more, stages in self._scales():
            levels.append((stages, h))
            if more:
                h = self.downsample(h)
        out = [None] * len(levels)
        with streams.fork(input.device, 'maskD') as f:
            for i, (stages, h) in enumerate(levels):
                with f.branch(i, reads=(h, cond)):
                    out[i] = self.singleD_forward(stages, h, cond)
                    f.produced(out[i])
        return out


def define_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1):
    return MultiscaleDiscriminator(input_nc, ndf, n_layers_D, get_norm_layer(norm_type=norm), use_sigmoid,
                                   num_D).apply(weights_init)


def define_mask_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1, num_objects=None):
    return MultiscaleMaskDiscriminator(input_nc, ndf, n_layers_D, get_norm_layer(norm_type=norm), use_sigmoid, num_D,
                                       num_objects).apply(weights_init)
