import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scene_generation_amd import ops, _hip
from scene_generation_amd.args import parser
from scene_generation_amd.synthetic import make_batch, make_vocab, batch_to, fill_deterministic
from scene_generation_amd.trainer import Trainer
argv = ['--image_size', '32,32', '--batch_size', '3', '--vgg_features_weight', '0', '--output_dir', '/tmp/o',
        '--n_downsample_global', '2', '--gconv_hidden_dim', '64', '--gconv_num_layers', '3', '--mask_size', '8',
        '--ndf', '8', '--ndf_mask', '8', '--crop_size', '16', '--d_obj_arch', 'C4-8-2,C4-16-2', '--pool_size', '2']
vocab, bk = make_vocab(12, 4, 35), dict(N=3, min_objs=2, max_objs=4, size=32, mask_size=8, num_objs=12, num_preds=4)
args = parser.parse_args(argv)
res = {}
for opt in (2, 1):
    _hip.set_option('fixedtap', opt)
    torch.manual_seed(0)
    tr = Trainer(args, vocab, device='cuda')
    for m in (tr.model, tr.netD, tr.obj_discriminator, tr.mask_discriminator):
        fill_deterministic(m)
    grads = {}
    for it in range(2):
        batch = batch_to(make_batch(seed=it, **bk), 'cuda')
        random.seed(5 + it)
        tr.model.noise_override = torch.zeros(1, args.mask_noise_dim, device='cuda')
        out = tr.step(batch, use_gt=(it == 0))
        for oname in ('optimizer', 'optimizer_d_img', 'optimizer_d_obj', 'optimizer_d_mask'):
            o = getattr(tr, oname)
            grads[(it, oname)] = o.fp.grad.clone()
        grads[(it, 'imgs_pred')] = out[0].detach().clone()
    res[opt] = (grads, tr)
g2, g1 = res[2][0], res[1][0]
tr = res[1][1]
for k in g2:
    if not torch.equal(g2[k], g1[k]):
        d = (g2[k] - g1[k]).abs()
        print('DIFF', k, float(d.max()), float(g2[k].abs().max()))
        if k[1] == 'optimizer':
            fp = tr.optimizer.fp
            names = [n for n, _ in tr.model.named_parameters()]
            for i, (o, n) in enumerate(zip(fp.offsets, fp.numels)):
                dd = float(d[o:o + n].max())
                if dd > 0:
                    print('    param %d %s: max diff %.3e (|g| max %.3e)' % (i, names[i] if i < len(names) else '?', dd, float(g2[k][o:o + n].abs().max())))
_hip.set_option('fixedtap', 1)
