"""GAN losses (surface of /root/reference/scene_generation/losses.py) as deterministic HIP reductions that
return 0-dim device tensors -- no ``.item()`` on the step path."""
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


def bce_loss(input, target):
    """Numerically stable BCE-with-logits against a CONSTANT target (losses.py:26-44; every call site uses
    full_like targets, losses.py:47-56)."""
    return ops.bce_logits_const(input, float(target))


def gan_g_loss(scores_fake):
    return ops.bce_logits_const(scores_fake.reshape(-1), 1.0)


def gan_d_loss(scores_real, scores_fake):
    assert scores_real.size() == scores_fake.size()
    return ops.bce_logits_const(scores_real.reshape(-1), 1.0) + ops.bce_logits_const(scores_fake.reshape(-1), 0.0)


def wgan_g_loss(scores_fake):
    raise NotImplementedError("gan_loss_type 'wgan' is not on the default training path (args.py:95)")


wgan_d_loss = lsgan_g_loss = lsgan_d_loss = wgan_g_loss


class GANLoss(nn.Module):
    """LSGAN objective: MSE against a constant label, summed over discriminator scales (losses.py:135-175)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=None):
        super().__init__()
        if not use_lsgan:
            raise NotImplementedError('--no_lsgan 1 (BCE on sigmoid outputs) is not on the default training path')
        self.real_label = target_real_label
        self.fake_label = target_fake_label

    def __call__(self, input, target_is_real):
        t = self.real_label if target_is_real else self.fake_label
        if isinstance(input[0], list):
            terms = [ops.mse_const(input_i[-1], t) for input_i in input]
            return terms[0] if len(terms) == 1 else weighted_sum(terms, [1.0] * len(terms))
        return ops.mse_const(input[-1], t)


class VGGLoss(nn.Module):
    """losses.py:179-224 needs torchvision's pretrained VGG19 (not obtainable offline): SURVEY 8f rank 2."""

    def __init__(self):
        super().__init__()
        raise NotImplementedError('VGGLoss needs pretrained VGG19 weights; run with --vgg_features_weight 0')
