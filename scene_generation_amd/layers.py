"""Layer library of the hot path, same surface as /root/reference/scene_generation/layers.py
(build_mlp :215-231, build_cnn :128-212, ResnetBlock :234-273, get_norm_layer :292-301,
get_activation :34-47, GlobalAvgPool :82-85, Interpolate :304-314).

The parameter containers subclass torch.nn's (same names/shapes/init => identical state_dict keys),
but every forward is a HIP launch through scene_generation_amd.ops.  ``FusedSequential`` keeps the
reference's module indices (so ``mask_net.1.weight`` etc. load unchanged) while fusing at run time:
  ReflectionPad2d + Conv2d        -> one implicit-GEMM launch (pad folded into the gather)
  Interpolate(x2) + Conv2d        -> one launch (nearest upsample folded into the gather)
  Conv2d/Linear + activation      -> activation in the GEMM epilogue
  Instance/BatchNorm + activation -> one normalisation launch
"""
import functools

import torch
import torch.nn as nn

from . import ops


# ---------------------------------------------------------------------------------------------
# leaf layers
# ---------------------------------------------------------------------------------------------

def _pair_to_int(v, what):
    if isinstance(v, (tuple, list)):
        assert all(x == v[0] for x in v), '%s must be square' % what
        return int(v[0])
    return int(v)


class Conv2d(nn.Conv2d):
    def forward(self, x, x2=None, reflect_pad=0, upsample=1, act=ops.ACT_NONE, slope=0.0):
        assert self.groups == 1 and _pair_to_int(self.dilation, 'dilation') == 1
        pad = _pair_to_int(self.padding, 'padding')
        reflect = reflect_pad > 0
        assert not (reflect and pad > 0)
        return ops.conv2d(x, self.weight, self.bias, stride=_pair_to_int(self.stride, 'stride'),
                          pad=reflect_pad if reflect else pad, reflect=reflect, upsample=upsample, act=act,
                          slope=slope, x2=x2)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x):
        return ops.conv_transpose2d(x, self.weight, self.bias, stride=_pair_to_int(self.stride, 'stride'),
                                    pad=_pair_to_int(self.padding, 'padding'),
                                    out_pad=_pair_to_int(self.output_padding, 'output_padding'))


class Linear(nn.Linear):
    def forward(self, x, act=ops.ACT_NONE, slope=0.0):
        return ops.linear(x, self.weight, self.bias, act=act, slope=slope)


class Embedding(nn.Embedding):
    def forward(self, idx):
        return ops.embedding(self.weight, idx)


class InstanceNorm2d(nn.InstanceNorm2d):
    def forward(self, x, skip=None, act=ops.ACT_NONE, slope=0.0):
        assert not self.affine and not self.track_running_stats, 'only the affine=False form is used (layers.py:296)'
        return ops.instance_norm(x, skip=skip, eps=self.eps, act=act, slope=slope)


class _BatchNormMixin:
    def _bn(self, x, act, slope):
        training = self.training or not self.track_running_stats
        mom = 0.1 if self.momentum is None else self.momentum
        return ops.batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var,
                              self.num_batches_tracked if (self.training and self.track_running_stats) else None,
                              training, mom, self.eps, act, slope)


class BatchNorm2d(nn.BatchNorm2d, _BatchNormMixin):
    def forward(self, x, act=ops.ACT_NONE, slope=0.0):
        return self._bn(x, act, slope)


class BatchNorm1d(nn.BatchNorm1d, _BatchNormMixin):
    def forward(self, x, act=ops.ACT_NONE, slope=0.0):
        return self._bn(x, act, slope)


class _Act(nn.Module):
    code, slope = ops.ACT_NONE, 0.0

    def forward(self, x):
        return ops.activation(x, self.code, self.slope)


class ReLU(_Act):
    code = ops.ACT_RELU

    def __init__(self, inplace=False):
        super().__init__()


class LeakyReLU(_Act):
    code = ops.ACT_LEAKY

    def __init__(self, negative_slope=0.01, inplace=False):
        super().__init__()
        self.slope = float(negative_slope)

    def extra_repr(self):
        return 'negative_slope=%g' % self.slope


class Tanh(_Act):
    code = ops.ACT_TANH


class Sigmoid(_Act):
    code = ops.ACT_SIGMOID


class ReflectionPad2d(nn.Module):
    def __init__(self, padding):
        super().__init__()
        self.padding = int(padding)

    def forward(self, x):
        return ops.reflect_pad(x, self.padding)


class Interpolate(nn.Module):
    """layers.py:304-314; the hot path only uses scale_factor=2, mode='nearest' (generators.py:20)."""

    def __init__(self, size=None, scale_factor=None, mode='nearest', align_corners=None):
        super().__init__()
        self.size, self.scale_factor, self.mode, self.align_corners = size, scale_factor, mode, align_corners
        self.hot = size is None and scale_factor == 2 and mode == 'nearest'

    def forward(self, x):
        if self.hot:
            return ops.upsample2(x)
        # other modes only occur off the training path (scripts/inception_score.py:29 resizes to 299x299 bilinear for the
        # evaluation network): plain torch on the device
        return torch.nn.functional.interpolate(x, size=self.size, scale_factor=self.scale_factor, mode=self.mode,
                                               align_corners=self.align_corners)


class GlobalAvgPool(nn.Module):
    def forward(self, x):
        return ops.global_avg_pool(x)


class AvgPool3s2(nn.Module):
    """nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False) (discriminators.py:100,186)."""

    def forward(self, x):
        return ops.avgpool3s2(x)


class MaxPool2d(nn.Module):
    """nn.MaxPool2d(kernel_size=k, stride=k) (VGG19 of VGGLoss; build_cnn 'P<k>' with pooling='max', layers.py:183-184).
    k = 2 runs the vectorised kernel, any other window the general one."""

    def __init__(self, kernel_size=2, stride=None):
        super().__init__()
        self.kernel_size = _pair_to_int(kernel_size, 'kernel_size')
        stride = self.kernel_size if stride is None else _pair_to_int(stride, 'stride')
        if stride != self.kernel_size or self.kernel_size < 1:
            raise NotImplementedError('MaxPool2d: only stride == kernel_size has a HIP kernel (the only form build_cnn builds)')

    def forward(self, x):
        return ops.maxpool2(x) if self.kernel_size == 2 else ops.pool2d(x, self.kernel_size, avg=False)

    def extra_repr(self):
        return 'kernel_size=%d, stride=%d' % (self.kernel_size, self.kernel_size)


class AvgPool2d(nn.Module):
    """nn.AvgPool2d(kernel_size=k, stride=k) (build_cnn 'P<k>' with pooling='avg', layers.py:185-186)."""

    def __init__(self, kernel_size=2, stride=None):
        super().__init__()
        self.kernel_size = _pair_to_int(kernel_size, 'kernel_size')
        stride = self.kernel_size if stride is None else _pair_to_int(stride, 'stride')
        if stride != self.kernel_size or self.kernel_size < 1:
            raise NotImplementedError('AvgPool2d: only stride == kernel_size has a HIP kernel (the only form build_cnn builds)')

    def forward(self, x):
        return ops.pool2d(x, self.kernel_size, avg=True)

    def extra_repr(self):
        return 'kernel_size=%d, stride=%d' % (self.kernel_size, self.kernel_size)


class ReplicationPad2d(nn.Module):
    def __init__(self, padding):
        super().__init__()
        self.padding = int(padding)

    def forward(self, x):
        return ops.replicate_pad(x, self.padding)


class Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


# ---------------------------------------------------------------------------------------------
# run-time fusing container
# ---------------------------------------------------------------------------------------------

class FusedSequential(nn.Sequential):
    def forward(self, x, skip=None, start=0, end=None):
        """``start`` / ``end``: run only modules [start, end) (the generator splits itself into a dynamic-shape stem and a
        static-shape remainder that is replayed as a hipGraph)"""
        mods = list(self)[start:end]
        n = len(mods)
        i = 0
        while i < n:
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < n else None
            if isinstance(m, ReflectionPad2d) and isinstance(nxt, Conv2d):
                nrm = mods[i + 2] if i + 2 < n else None
                # (the fused operator IS plain InstanceNorm: an affine / running-statistics norm or a dilated conv -- both
                #  rejected by the unfused modules' own forwards -- must reach those checks, not be computed as something else)
                if (isinstance(nrm, InstanceNorm2d) and nxt.groups == 1
                        and not getattr(nrm, 'affine', False) and not getattr(nrm, 'track_running_stats', False)
                        and _pair_to_int(getattr(nxt, 'dilation', 1), 'dilation') == 1
                        and ops.conv_instnorm_fusable(x, nxt.weight, int(m.padding),
                                                      _pair_to_int(nxt.stride, 'stride'), _pair_to_int(nxt.padding, 'padding'))):
                    # pad + conv + InstanceNorm (+ act) (+ the block's residual when this is its last norm): one fused operator
                    act, slope, used = _peek_act(mods, i + 3, norm=True)
                    sk = None
                    if i + 3 + used == n and skip is not None:
                        sk, skip = skip, None
                    x = ops.conv2d_instnorm(x, nxt.weight, nxt.bias, skip=sk, eps=nrm.eps, act=act, slope=slope)
                    i += 3 + used
                    continue
                act, slope, used = _peek_act(mods, i + 2)
                x = nxt(x, reflect_pad=m.padding, act=act, slope=slope)
                i += 2 + used
            elif isinstance(m, Interpolate) and m.hot and isinstance(nxt, Conv2d):
                act, slope, used = _peek_act(mods, i + 2)
                x = nxt(x, upsample=2, act=act, slope=slope)
                i += 2 + used
            elif isinstance(m, (Conv2d, Linear)):
                act, slope, used = _peek_act(mods, i + 1)
                x = m(x, act=act, slope=slope)
                i += 1 + used
            elif isinstance(m, (InstanceNorm2d, BatchNorm2d, BatchNorm1d)):
                act, slope, used = _peek_act(mods, i + 1, norm=True)
                if isinstance(m, InstanceNorm2d) and i + 1 + used == n and skip is not None:
                    x = m(x, skip=skip, act=act, slope=slope)
                    skip = None
                else:
                    x = m(x, act=act, slope=slope)
                i += 1 + used
            else:
                x = m(x)
                i += 1
        if skip is not None:
            x = x + skip
        return x


def _peek_act(mods, j, norm=False):
    """activation module directly following -> (code, slope, 1) else (none, 0, 0).  tanh/sigmoid only fuse into GEMM
    epilogues, not into normalisation launches."""
    if j < len(mods) and isinstance(mods[j], _Act):
        a = mods[j]
        if norm and a.code not in (ops.ACT_RELU, ops.ACT_LEAKY):
            return ops.ACT_NONE, 0.0, 0
        return a.code, a.slope, 1
    return ops.ACT_NONE, 0.0, 0


# ---------------------------------------------------------------------------------------------
# builders (reference surface)
# ---------------------------------------------------------------------------------------------

def get_normalization_2d(channels, normalization):
    if normalization == 'instance':
        return InstanceNorm2d(channels)
    elif normalization == 'batch':
        return BatchNorm2d(channels)
    elif normalization == 'none':
        return None
    raise ValueError('Unrecognized normalization type "%s"' % normalization)


def get_activation(name):
    """layers.py:34-47 quirk kept: whatever the name, a LeakyReLU is built; only '-slope' is honoured."""
    kwargs = {}
    if name.lower().startswith('leakyrelu') and '-' in name:
        kwargs['negative_slope'] = float(name.split('-')[1])
    return LeakyReLU(**kwargs)


class Dropout(nn.Module):
    """nn.Dropout (layers.py:230): the Bernoulli mask comes from torch's device generator, the multiply is a HIP launch; eval
    mode is the identity.  (No reference pin is possible for the random mask; the arithmetic is tested against x * mask / (1-p).)"""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        if not 0.0 <= p <= 1.0:
            raise ValueError('dropout probability has to be between 0 and 1, but got %r' % (p,))
        self.p = float(p)

    def forward(self, x):
        return ops.dropout(x, self.p, self.training)

    def extra_repr(self):
        return 'p=%g' % self.p


class ResidualBlock(nn.Module):
    """build_cnn's 'R' layer (layers.py:84-118): x + [norm, act, Conv(K, same), norm, act, Conv(K, same)](x), state_dict keys
    ``net.<i>`` as in the reference.  Two properties of the reference forward are kept:
      * it evaluates the branch TWICE and adds the second result (layers.py:114-115), so BatchNorm running statistics and
        ``num_batches_tracked`` advance twice per training forward -- done here too (only when it is observable: training mode
        and a BatchNorm inside),
      * with padding='valid' its shortcut slice ``x[:, :, 0:-0, 0:-0]`` is empty and the add fails: rejected at construction."""

    def __init__(self, channels, normalization='batch', activation='relu', padding='same', kernel_size=3, init='default'):
        super().__init__()
        K, P = kernel_size, _get_padding(kernel_size, padding)
        if P == 0:
            raise ValueError('ResidualBlock(padding="valid") cannot run in the reference either: its shortcut slice is empty '
                             '(layers.py:111-113)')
        self.padding = P
        mods = [get_normalization_2d(channels, normalization), get_activation(activation),
                Conv2d(channels, channels, kernel_size=K, padding=P),
                get_normalization_2d(channels, normalization), get_activation(activation),
                Conv2d(channels, channels, kernel_size=K, padding=P)]
        self.net = FusedSequential(*[m for m in mods if m is not None])

    def forward(self, x):
        if self.training and any(isinstance(m, nn.modules.batchnorm._BatchNorm) for m in self.net):
            with torch.no_grad():
                self.net(x)                      # the reference's first, discarded evaluation: only its buffer updates remain
        return ops.add(x, self.net(x))


def _get_padding(K, mode):
    if mode == 'valid':
        return 0
    if mode == 'same':
        assert K % 2 == 1, 'Invalid kernel size %d for "same" padding' % K
        return (K - 1) // 2
    raise ValueError('Invalid padding "%s"' % mode)


def build_cnn(arch, normalization='batch', activation='relu', padding='same', pooling='max', init='default'):
    """Architecture-string CNN builder (layers.py:128-212), every layer kind of the reference: IX (input channels), CK-X[-S]
    (conv), R (residual block), UX (nearest upsample), PX (max / avg pooling, window = stride = X), FC-X-Y."""
    if isinstance(arch, str):
        arch = arch.split(',')
    cur_C = 3
    if len(arch) > 0 and arch[0][0] == 'I':
        cur_C = int(arch[0][1:])
        arch = arch[1:]
    first_conv = True
    flat = False
    layers = []
    for i, s in enumerate(arch):
        if s[0] == 'C':
            if not first_conv:
                layers.append(get_normalization_2d(cur_C, normalization))
                layers.append(get_activation(activation))
            first_conv = False
            vals = [int(v) for v in s[1:].split('-')]
            K, next_C = vals[0], vals[1]
            stride = vals[2] if len(vals) == 3 else 1
            layers.append(Conv2d(cur_C, next_C, kernel_size=K, padding=_get_padding(K, padding), stride=stride))
            cur_C = next_C
        elif s[0] == 'R':                                   # layers.py:172-177: residual block, no norm in front of the very first conv
            layers.append(ResidualBlock(cur_C, normalization='none' if first_conv else normalization, activation=activation,
                                        padding=padding, init=init))
            first_conv = False
        elif s[0] == 'U':
            layers.append(Interpolate(scale_factor=int(s[1:]), mode='nearest'))
        elif s[0] == 'P':                                   # layers.py:181-189: k x k pooling, stride k
            factor = int(s[1:])
            if pooling == 'max':
                layers.append(MaxPool2d(kernel_size=factor, stride=factor))
            elif pooling == 'avg':
                layers.append(AvgPool2d(kernel_size=factor, stride=factor))
            else:
                # (the reference leaves ``pool`` unbound here and dies with UnboundLocalError, layers.py:183-187)
                raise ValueError('Invalid pooling "%s"' % pooling)
        elif s[:2] == 'FC':                                 # layers.py:190-199: flatten + Linear (+ activation unless last)
            _, Din, Dout = s.split('-')
            if not flat:
                layers.append(Flatten())
            flat = True
            layers.append(Linear(int(Din), int(Dout)))
            if i + 1 < len(arch):
                layers.append(get_activation(activation))
            cur_C = int(Dout)
        else:
            raise ValueError('Invalid layer "%s"' % s)
    layers = [l for l in layers if l is not None]
    return FusedSequential(*layers), cur_C


def build_mlp(dim_list, activation='relu', batch_norm='none', dropout=0, final_nonlinearity=True):
    layers = []
    for i in range(len(dim_list) - 1):
        layers.append(Linear(dim_list[i], dim_list[i + 1]))
        final_layer = (i == len(dim_list) - 2)
        if not final_layer or final_nonlinearity:
            if batch_norm == 'batch':
                layers.append(BatchNorm1d(dim_list[i + 1]))
            if activation == 'relu':
                layers.append(ReLU())
            elif activation == 'leakyrelu':
                layers.append(LeakyReLU())
        if dropout > 0:
            layers.append(Dropout(p=dropout))
    return FusedSequential(*layers)


class ResnetBlock(nn.Module):
    """x + [pad(1), Conv3x3, norm, act, (Dropout), pad(1), Conv3x3, norm](x)  (layers.py:234-273), padding_type 'reflect'
    (the generator's, generators.py:79: the pad is folded into the conv's gather), 'replicate' (ReplicationPad2d + conv) or
    'zero' (Conv2d(padding=1), no pad modules -- the module indices and hence the state_dict keys follow the reference in all
    three).  Reflect form: four launches -- two pad-folded implicit GEMMs, IN+ReLU, IN+residual-add."""

    def __init__(self, dim, padding_type, norm_layer, activation=None, use_dropout=False):
        super().__init__()
        activation = ReLU(True) if activation is None else activation

        def conv():
            if padding_type == 'reflect':
                return [ReflectionPad2d(1), Conv2d(dim, dim, kernel_size=3, padding=0)]
            if padding_type == 'replicate':
                return [ReplicationPad2d(1), Conv2d(dim, dim, kernel_size=3, padding=0)]
            if padding_type == 'zero':
                return [Conv2d(dim, dim, kernel_size=3, padding=1)]
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)      # layers.py:250
        first = conv() + [norm_layer(dim), activation]
        if use_dropout:                                     # layers.py:256-257 (pix2pixHD option, off in generators.py:79)
            first.append(Dropout(0.5))
        self.conv_block = FusedSequential(*first, *conv(), norm_layer(dim))

    def forward(self, x):
        return self.conv_block(x, skip=x)


def get_norm_layer(norm_type='instance'):
    if norm_type == 'batch':
        return functools.partial(BatchNorm2d, affine=True)
    elif norm_type == 'instance':
        return functools.partial(InstanceNorm2d, affine=False)
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)
