"""Shared by the CPU (oracle) and GPU (HIP) tests of the FULL-WIDTH reference step goldens (tests/golden/step_full_*.npz,
captured by tools/make_golden.py::golden_step_full from the reference Trainer, trainer.py:205-325 / train.py:190-215)."""
import random

import numpy as np
import torch

from scene_generation_amd.synthetic import fill_deterministic, make_batch, make_vocab
from scene_generation_amd.args import parser

OUT_NAMES = ['imgs_pred', 'boxes_pred', 'masks_pred', 'layout', 'layout_pred', 'layout_wrong']


def slices(t, n=64):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return torch.cat([f[:n], f[-n:], f[::step][:n]])


def run_step_full(g, make_trainer, to_device=lambda b: b, before_step=None):
    """Replays the two golden iterations on ``make_trainer(args, vocab)`` and returns the measured deviations, one dict per
    iteration: {'loss': max relative loss error, 'out_abs': max slice error (relative to the slice's largest entry),
    'out_stat': max relative |sum| error of an output, 'param_stat': max relative |sum| error of a post-step parameter or
    buffer, 'worst_*': where}"""
    argv = [str(a) for a in g['argv'].tolist()]
    args = parser.parse_args(argv)
    N, lo, hi, size = int(g['N']), int(g['min_objs']), int(g['max_objs']), int(g['size'])
    tr = make_trainer(args, make_vocab())
    for m in (tr.model, tr.netD, tr.obj_discriminator, tr.mask_discriminator):
        fill_deterministic(m)
    random.seed(4321)
    devs = []
    for it in range(2):
        dev = {'loss': 0.0, 'out_abs': 0.0, 'out_stat': 0.0, 'param_stat': 0.0, 'worst_param': '', 'worst_loss': '',
               'worst_out': ''}
        devs.append(dev)
        batch = make_batch(N=N, min_objs=lo, max_objs=hi, size=size, seed=200 + it)
        pre = 'it%d_' % it
        noise = torch.from_numpy(g[pre + 'noise'])
        if before_step is not None:
            before_step(tr, batch, noise)
        tr.model.noise_override = noise.to(next(tr.model.parameters()).device)
        out = tr.step(to_device(batch), use_gt=(it == 0))
        for n, t in zip(OUT_NAMES, out):
            t = t.detach().float().cpu()
            ref = g[pre + n + '_stats']
            got = float(t.double().abs().sum())
            dev['out_stat'] = max(dev['out_stat'], abs(got - ref[1]) / max(1.0, ref[1]))
            ref_s = torch.from_numpy(g[pre + n + '_slices']).double()
            e = float((slices(t).double() - ref_s).abs().max()) / max(1.0, float(ref_s.abs().max()))
            if e > dev['out_abs']:
                dev['out_abs'], dev['worst_out'] = e, n
        for lname, L in [('g', tr.generator_losses), ('dmask', tr.d_mask_losses), ('dobj', tr.d_obj_losses),
                         ('dimg', tr.d_img_losses)]:
            for k, v in L.items():
                ref = float(g[pre + 'loss_' + lname + '_' + k])
                e = abs(float(v) - ref) / max(1.0, abs(ref))
                if e > dev['loss']:
                    dev['loss'], dev['worst_loss'] = e, '%s%s_%s' % (pre, lname, k)
        for mname, m in [('model', tr.model), ('netD', tr.netD), ('objD', tr.obj_discriminator),
                         ('maskD', tr.mask_discriminator)]:
            sd = m.state_dict()
            keys = g[pre + 'keys_' + mname].tolist()
            st = g[pre + 'stats_' + mname]
            assert sorted(sd.keys()) == keys, 'state_dict keys of %s differ from the reference' % mname
            for k, (s, a) in zip(keys, st):
                got = float(sd[k].double().abs().sum())
                e = abs(got - a) / max(1.0, a)
                if e > dev['param_stat']:
                    dev['param_stat'], dev['worst_param'] = e, '%s%s.%s' % (pre, mname, k)
    return devs
