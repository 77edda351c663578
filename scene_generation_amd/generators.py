"""Generators (surface of /root/reference/scene_generation/generators.py): mask_net :16-28,
AppearanceEncoder :31-48, define_G :51-57, GlobalGenerator :62-91 -- built from HIP-backed layers, same
module indices => same state_dict keys."""
import torch.nn as nn

from .layers import (GlobalAvgPool, build_cnn, ResnetBlock, get_norm_layer, Interpolate, Conv2d, ConvTranspose2d,
                     BatchNorm2d, ReLU, Tanh, Linear, ReflectionPad2d, FusedSequential)


def weights_init(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif classname.find('BatchNorm2d') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def mask_net(dim, mask_size):
    layers, cur_size = [], 1
    while cur_size < mask_size:
        layers += [Interpolate(scale_factor=2, mode='nearest'), Conv2d(dim, dim, kernel_size=3, padding=1),
                   BatchNorm2d(dim), ReLU()]
        cur_size *= 2
    if cur_size != mask_size:
        raise ValueError('Mask size must be a power of 2')
    layers.append(Conv2d(dim, 1, kernel_size=1))
    return FusedSequential(*layers)


class AppearanceEncoder(nn.Module):
    def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', vecs_size=1024,
                 pooling='avg'):
        super().__init__()
        self.vocab = vocab
        cnn, channels = build_cnn(arch=arch, normalization=normalization, activation=activation, pooling=pooling,
                                  padding=padding)
        self.cnn = FusedSequential(cnn, GlobalAvgPool(), Linear(channels, vecs_size))

    def forward(self, crops):
        return self.cnn(crops)


def define_G(input_nc, output_nc, ngf, n_downsample_global=3, n_blocks_global=9, norm='instance'):
    netG = GlobalGenerator(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global, get_norm_layer(norm))
    netG.apply(weights_init)
    return netG


class GlobalGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=BatchNorm2d,
                 padding_type='reflect'):
        assert n_blocks >= 0
        super().__init__()
        activation = ReLU(True)
        model = [ReflectionPad2d(3), Conv2d(input_nc, ngf, kernel_size=7, padding=0), norm_layer(ngf), activation]
        for i in range(n_downsampling):
            mult = 2 ** i
            model += [Conv2d(ngf * mult, ngf * mult * 2, kernel_size=3, stride=2, padding=1),
                      norm_layer(ngf * mult * 2), activation]
        mult = 2 ** n_downsampling
        for i in range(n_blocks):
            model += [ResnetBlock(ngf * mult, padding_type=padding_type, activation=activation,
                                  norm_layer=norm_layer)]
        for i in range(n_downsampling):
            mult = 2 ** (n_downsampling - i)
            model += [ConvTranspose2d(ngf * mult, int(ngf * mult / 2), kernel_size=3, stride=2, padding=1,
                                      output_padding=1), norm_layer(int(ngf * mult / 2)), activation]
        model += [ReflectionPad2d(3), Conv2d(ngf, output_nc, kernel_size=7, padding=0), Tanh()]
        self.model = FusedSequential(*model)

    def forward(self, input):
        return self.model(input)
