"""GAN losses (surface of /root/reference/scene_generation/losses.py) as deterministic HIP reductions that
return 0-dim device tensors -- no ``.item()`` on the step path."""
import torch
import torch.nn as nn

from . import ops
from .utils import weighted_sum


def get_gan_losses(gan_type):
    """Returns (g_loss(scores_fake), d_loss(scores_real, scores_fake)) (losses.py:8-23)."""
    if gan_type == 'gan':
        return gan_g_loss, gan_d_loss
    elif gan_type == 'wgan':
        return wgan_g_loss, wgan_d_loss
    elif gan_type == 'lsgan':
        return lsgan_g_loss, lsgan_d_loss
    raise ValueError('Unrecognized GAN type "%s"' % gan_type)


def _pair(kind, scores_real, scores_fake):
    """l(real | 1) + l(fake | 0): the two terms and their sum as ONE launch (ops.multi_loss; host tensors: plain terms)"""
    if scores_real.is_cuda:
        return ops.multi_loss(kind, [scores_real.reshape(-1), scores_fake.reshape(-1)], None, [1.0, 0.0], [1.0, 1.0])
    one = {ops.LOSS_BCE_CONST: ops.bce_logits_const, ops.LOSS_MSE_SIGMOID_CONST: ops.mse_sigmoid_const}[kind]
    return weighted_sum([one(scores_real.reshape(-1), 1.0), one(scores_fake.reshape(-1), 0.0)], [1.0, 1.0])


def bce_loss(input, target):
    """Numerically stable BCE-with-logits against a CONSTANT target (losses.py:26-44; every call site uses
    full_like targets, losses.py:47-56)."""
    return ops.bce_logits_const(input, float(target))


def gan_g_loss(scores_fake):
    return ops.bce_logits_const(scores_fake.reshape(-1), 1.0)


def gan_d_loss(scores_real, scores_fake):
    assert scores_real.size() == scores_fake.size()
    return _pair(ops.LOSS_BCE_CONST, scores_real, scores_fake)


def wgan_g_loss(scores_fake):
    """losses.py:93-101"""
    return weighted_sum([ops.mean(scores_fake)], [-1.0])


def wgan_d_loss(scores_real, scores_fake):
    """losses.py:104-112"""
    return weighted_sum([ops.mean(scores_fake), ops.mean(scores_real)], [1.0, -1.0])


def lsgan_g_loss(scores_fake):
    """losses.py:115-119: MSE of sigmoid(scores) against 1"""
    return ops.mse_sigmoid_const(scores_fake.reshape(-1), 1.0)


def lsgan_d_loss(scores_real, scores_fake):
    """losses.py:122-132"""
    assert scores_real.size() == scores_fake.size()
    return _pair(ops.LOSS_MSE_SIGMOID_CONST, scores_real, scores_fake)


class GANLoss(nn.Module):
    """MSE (use_lsgan, the default) or BCE-on-probabilities against a constant label, summed over discriminator scales
    (losses.py:135-175)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=None):
        super().__init__()
        self.use_lsgan = bool(use_lsgan)
        self.real_label = target_real_label
        self.fake_label = target_fake_label

    def _one(self, pred, t):
        return ops.mse_const(pred, t) if self.use_lsgan else ops.bce_prob_const(pred, t)

    def __call__(self, input, target_is_real):
        t = self.real_label if target_is_real else self.fake_label
        if isinstance(input[0], list):
            if len(input) > 1 and input[0][-1].is_cuda:      # the per-scale terms and their sum as one launch
                kind = ops.LOSS_MSE_CONST if self.use_lsgan else ops.LOSS_BCE_PROB_CONST
                return ops.multi_loss(kind, [input_i[-1] for input_i in input], None, [t] * len(input))
            terms = [self._one(input_i[-1], t) for input_i in input]
            return terms[0] if len(terms) == 1 else weighted_sum(terms, [1.0] * len(terms))
        return self._one(input[-1], t)


# torchvision's VGG configuration 'E' (vgg19.features): 16 conv3x3(pad 1)+ReLU and 5 max-pools; the reference keeps
# features[0:30] in five slices that end at relu1_1, relu2_1, relu3_1, relu4_1, relu5_1 (losses.py:183-198)
_VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')
_VGG19_SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))


def vgg19_feature_layers():
    """[(index in torchvision's vgg19.features, kind, cin, cout)] for features[0:30]"""
    out, cin, idx = [], 3, 0
    for v in _VGG19_CFG:
        if v == 'M':
            out.append((idx, 'pool', cin, cin))
            idx += 1
        else:
            out.append((idx, 'conv', cin, v))
            out.append((idx + 1, 'relu', v, v))
            cin = v
            idx += 2
    return [l for l in out if l[0] < 30]


class Vgg19(nn.Module):
    """losses.py:179-209: torchvision vgg19.features[0:30] cut into slice1..slice5 (module names = the torchvision
    indices, so a torchvision ``features.<i>.weight`` state_dict maps to ``slice<k>.<i>.weight``).  Every conv is
    ``conv3x3(pad 1) + ReLU`` in ONE HIP launch (Winograd F(2x2,3x3) where it applies); parameters are frozen, so
    backward is data gradients only.

    torchvision's ImageNet weights cannot be downloaded here: without ``weights`` the filters are drawn He-normal
    (torchvision's own VGG init, fan_out) from a private generator -- enough for the arithmetic, shapes and cost of the
    reference's default training step, not for its perceptual meaning."""

    def __init__(self, requires_grad=False, weights=None, seed=19):
        super().__init__()
        from .layers import Conv2d, ReLU, MaxPool2d, FusedSequential
        layers = vgg19_feature_layers()
        for k, (lo, hi) in enumerate(_VGG19_SLICES):
            seq = FusedSequential()
            for idx, kind, cin, cout in layers:
                if lo <= idx < hi:
                    seq.add_module(str(idx), Conv2d(cin, cout, kernel_size=3, padding=1) if kind == 'conv' else
                                   (ReLU(True) if kind == 'relu' else MaxPool2d(2, 2)))
            setattr(self, 'slice%d' % (k + 1), seq)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, Conv2d):
                    std = (2.0 / (m.out_channels * 9)) ** 0.5
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
                    m.bias.zero_()
        if weights is not None:
            self.load_torchvision_state_dict(weights)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False
        # static shapes, frozen parameters: one hipGraph for the forward and one for the data gradient (graphs.py).  The five
        # feature maps are handed out as COPIES: VGGLoss calls the extractor twice per step (x, then y) and under no_grad /
        # with a detached x both calls replay the same graph, whose static output buffers the second call would overwrite
        from .graphs import GraphedSegment
        self._graphed = GraphedSegment(self._features, params=list(self.parameters()), name='Vgg19', clone_outputs=True,
                                       modules=[self])

    def load_torchvision_state_dict(self, sd):
        """``sd``: torchvision vgg19 state_dict (keys ``features.<i>.weight|bias``) or a path to one"""
        if isinstance(sd, str):
            sd = torch.load(sd, map_location='cpu')
        own = dict(self.named_parameters())
        with torch.no_grad():
            for name, p in own.items():
                idx = name.split('.', 1)[1]                         # 'slice2.5.weight' -> '5.weight'
                src = sd.get('features.' + idx, sd.get(idx))
                if src is None:
                    raise KeyError('VGG19 weights: no entry for features.%s' % idx)
                p.copy_(src.to(p.device, p.dtype))

    def _features(self, X):
        h_relu1 = self.slice1(X)
        h_relu2 = self.slice2(h_relu1)
        h_relu3 = self.slice3(h_relu2)
        h_relu4 = self.slice4(h_relu3)
        h_relu5 = self.slice5(h_relu4)
        return [h_relu1, h_relu2, h_relu3, h_relu4, h_relu5]

    def forward(self, X):
        return self._graphed(X)


class VGGLoss(nn.Module):
    """losses.py:212-224: sum_i w_i * L1(vgg(x)_i, vgg(y)_i.detach()), w = 1/32, 1/16, 1/8, 1/4, 1."""

    def __init__(self, weights=None):
        super().__init__()
        self.vgg = Vgg19(weights=weights)
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        x_vgg = self.vgg(x)
        with torch.no_grad():            # y is the ground-truth image and the network is frozen: nothing to record
            y_vgg = self.vgg(y)
        if x_vgg[0].is_cuda:
            return ops.l1_multi(x_vgg, y_vgg, self.weights)         # five terms + their sum: one launch
        return weighted_sum([ops.l1(a, b) for a, b in zip(x_vgg, y_vgg)], self.weights)
