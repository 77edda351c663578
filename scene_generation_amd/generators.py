"""Generators (surface of /root/reference/scene_generation/generators.py): mask_net :16-28,
AppearanceEncoder :31-48, define_G :51-57, GlobalGenerator :62-91 -- built from HIP-backed layers, same
module indices => same state_dict keys."""
import math

import torch.nn as nn

from .layers import (GlobalAvgPool, build_cnn, ResnetBlock, get_norm_layer, Interpolate, Conv2d, ConvTranspose2d,
                     BatchNorm2d, ReLU, Tanh, Linear, ReflectionPad2d, FusedSequential)


def weights_init(m):
    """pix2pixHD-style init (generators.py:7-13): every module whose class name contains 'Conv' gets N(0, 0.02) weights,
    BatchNorm2d gets N(1, 0.02) scale and zero shift.  Same traversal order => same RNG consumption as the reference."""
    name = type(m).__name__
    if 'Conv' in name:
        nn.init.normal_(m.weight.data, mean=0.0, std=0.02)
    elif 'BatchNorm2d' in name:
        nn.init.normal_(m.weight.data, mean=1.0, std=0.02)
        m.bias.data.zero_()


def mask_net(dim, mask_size):
    """1x1 -> mask_size x mask_size by doubling (generators.py:16-28): [up x2, conv3x3, BN, ReLU] per octave, 1x1 head."""
    octaves = int(round(math.log2(mask_size))) if mask_size >= 1 else -1
    if octaves < 0 or 2 ** octaves != mask_size:
        raise ValueError('Mask size must be a power of 2')
    stages = []
    for _ in range(octaves):
        stages.extend((Interpolate(scale_factor=2, mode='nearest'), Conv2d(dim, dim, kernel_size=3, padding=1),
                       BatchNorm2d(dim), ReLU()))
    return FusedSequential(*stages, Conv2d(dim, 1, kernel_size=1))


class AppearanceEncoder(nn.Module):
    """crop -> conv stack -> global average pool -> linear (generators.py:31-48)"""

    def __init__(self, vocab, arch, normalization='none', activation='relu', padding='same', vecs_size=1024,
                 pooling='avg'):
        super().__init__()
        self.vocab = vocab
        trunk, width = build_cnn(arch=arch, normalization=normalization, activation=activation, pooling=pooling,
                                 padding=padding)
        self.cnn = FusedSequential(trunk, GlobalAvgPool(), Linear(width, vecs_size))

    def forward(self, crops):
        return self.cnn(crops)


def define_G(input_nc, output_nc, ngf, n_downsample_global=3, n_blocks_global=9, norm='instance'):
    # (the reference also asserts CUDA and moves the module there, generators.py:54-55; placement is the Trainer's job here)
    return GlobalGenerator(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global,
                           get_norm_layer(norm)).apply(weights_init)


class _RangeRunner(object):
    """``h -> seq(h, start=a, end=b)`` as an OBJECT holding the sequential: ``copy.deepcopy`` of a GlobalGenerator (EMA copy,
    model snapshot) then rebinds the graphed segments to the COPY's layers through the memo (a closure / lambda is atomic to
    deepcopy and would keep running the original's layers -- ADVICE r3)."""

    def __init__(self, seq, a, b):
        self.seq, self.a, self.b = seq, a, b

    def __call__(self, h):
        return self.seq(h, start=self.a, end=self.b)


class GlobalGenerator(nn.Module):
    """pix2pixHD global generator (generators.py:62-91): 7x7 stem, stride-2 encoder, residual trunk, transposed-conv
    decoder, 7x7 tanh head.  One flat Sequential => the reference's ``model.<index>`` state_dict keys."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=BatchNorm2d,
                 padding_type='reflect'):
        if n_blocks < 0:
            raise AssertionError('n_blocks must be non-negative')
        super().__init__()
        relu = ReLU(True)                                   # one shared module instance, like the reference
        widths = [ngf << level for level in range(n_downsampling + 1)]
        seq = [ReflectionPad2d(3), Conv2d(input_nc, widths[0], kernel_size=7, padding=0), norm_layer(widths[0]), relu]
        for narrow, wide in zip(widths[:-1], widths[1:]):
            seq += [Conv2d(narrow, wide, kernel_size=3, stride=2, padding=1), norm_layer(wide), relu]
        seq += [ResnetBlock(widths[-1], padding_type=padding_type, activation=relu, norm_layer=norm_layer)
                for _ in range(n_blocks)]
        for wide, narrow in zip(widths[:0:-1], widths[-2::-1]):
            seq += [ConvTranspose2d(wide, narrow, kernel_size=3, stride=2, padding=1, output_padding=1),
                    norm_layer(narrow), relu]
        seq += [ReflectionPad2d(3), Conv2d(widths[0], output_nc, kernel_size=7, padding=0), Tanh()]
        self.model = FusedSequential(*seq)
        # everything behind the first convolution has static shapes (N x ngf x H x W in, the image out) whatever the scene
        # graphs look like: replayed as hipGraphs (graphs.py); the stem sees the per-batch object lists (factored layout
        # conv) and stays eager.  The remainder is cut into up to three segments at residual-block boundaries, each with
        # its own forward / backward graph: the backward then hands its parameter gradients to the optimiser -- and, under
        # data parallelism, to the bucketed all-reduce -- segment by segment (last third of the network first) instead of
        # all at once behind one monolithic replay.  Only the image is handed out as a copy: callers keep it across
        # iterations (train.py:203,219)
        from .graphs import GraphedSegment
        mods = list(self.model)
        blocks = [i for i, m in enumerate(mods) if isinstance(m, ResnetBlock)]
        cuts = [2]
        if len(blocks) >= 3:
            cuts += [blocks[len(blocks) // 3], blocks[(2 * len(blocks)) // 3]]
        cuts.append(len(mods))
        self._tail_cuts = cuts
        self._tail = [GraphedSegment(self._runner(a, b), params=[p for m in mods[a:b] for p in m.parameters()],
                                     clone_outputs=(b == len(mods)), name='GlobalGenerator[%d:%d]' % (a, b), modules=mods[a:b])
                      for a, b in zip(cuts[:-1], cuts[1:])]

    def _runner(self, a, b):
        return _RangeRunner(self.model, a, b)

    def forward(self, input):
        h = self.model(input, end=2)          # ReflectionPad2d(3) + Conv7x7 over the layout
        for seg in self._tail:
            h = seg(h)
        return h
