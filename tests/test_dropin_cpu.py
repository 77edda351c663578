"""The drop-in switch (scene_generation_amd.install_as) against the reference's own entry script, CPU only.

Container-only: needs the read-only reference checkout (it never travels to the GPU box, where this file skips).  A
subprocess aliases the package as ``scene_generation``, executes the import block of the reference's train.py
(train.py:9-13) with stand-ins for the third-party packages this image lacks (torchvision, pycocotools, skimage,
tensorboardX), then builds the checkpoint dict and the Trainer the way train.py:119-188 does.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

SCRIPT = r'''
import ast, os, sys, types, json
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(ref)r)
import PIL.Image                      # data/utils.py touches PIL.Image after a bare ``import PIL``
for n in ['torchvision', 'torchvision.transforms', 'torchvision.models', 'torchvision.utils', 'torchvision.datasets',
          'torchvision.models.inception', 'pycocotools', 'pycocotools.mask', 'skimage', 'skimage.transform', 'tensorboardX']:
    sys.modules[n] = types.ModuleType(n)
tv = sys.modules['torchvision']
tv.transforms, tv.models, tv.utils, tv.datasets = [sys.modules['torchvision.' + k] for k in
                                                   ('transforms', 'models', 'utils', 'datasets')]
sys.modules['torchvision.models.inception'].inception_v3 = lambda *a, **k: None
sys.modules['skimage.transform'].resize = lambda *a, **k: None
for n in ('Normalize', 'Compose', 'ToTensor', 'Resize'):
    setattr(tv.transforms, n, type(n, (), {'__init__': lambda s, *a, **k: None}))

import scene_generation_amd
scene_generation_amd.install_as('scene_generation')

# ---- train.py:9-15 executed verbatim from the reference file (the import statements only) ----
src = open(os.path.join(%(ref)r, 'train.py')).read()
tree = ast.parse(src)
imports = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
ns = {}
exec(compile(ast.Module(body=imports, type_ignores=[]), 'train.py', 'exec'), ns)
assert ns['Trainer'].__module__ == 'scene_generation_amd.trainer', ns['Trainer'].__module__
assert ns['get_args'].__module__ == 'scene_generation_amd.args'
assert ns['CocoSceneGraphDataset'].__module__ == 'scene_generation.data.coco'
assert os.path.realpath(sys.modules['scene_generation.data.coco'].__file__).startswith(os.path.realpath(%(ref)r))
assert os.path.realpath(sys.modules['scene_generation.metrics'].__file__).startswith(os.path.realpath(%(ref)r))
import torch
iou = ns['jaccard'](torch.tensor([[0., 0., 1., 1.]]), torch.tensor([[0., 0., .5, 1.]]))
assert abs(float(iou[0]) - 0.5) < 1e-6
# the evaluation script's only import from the package (scripts/inception_score.py:12,30)
from scene_generation.layers import Interpolate
up = Interpolate(size=(9, 9), mode='bilinear')
assert tuple(up(torch.zeros(1, 3, 4, 4)).shape) == (1, 3, 9, 9)

# ---- train.py:119-163 (get_checkpoint) + :166-170 (Trainer construction), executed from the reference source ----
fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'get_checkpoint'][0]
exec(compile(ast.Module(body=[fn], type_ignores=[]), 'train.py', 'exec'), ns)
from scene_generation_amd.synthetic import make_vocab
args = ns['get_args'](['--output_dir', %(out)r, '--image_size', '32,32', '--n_downsample_global', '2', '--ndf', '8',
                       '--ndf_mask', '8', '--gconv_hidden_dim', '64'])
os.makedirs(args.output_dir, exist_ok=True)
vocab = make_vocab()
t, epoch, checkpoint = ns['get_checkpoint'](args, vocab)
trainer = ns['Trainer'](args, vocab, checkpoint)
for attr in ('model', 'netD', 'obj_discriminator', 'mask_discriminator', 'optimizer', 'optimizer_d_obj',
             'optimizer_d_mask', 'optimizer_d_img', 'num_obj', 'writer', 'criterionVGG', 'criterionGAN'):
    assert getattr(trainer, attr) is not None, attr
assert trainer.criterionVGG is not None          # the reference's default flags (vgg_features_weight 10) construct
assert set(checkpoint['model_kwargs']) >= {'vocab', 'image_size', 'rep_size'} and checkpoint['d_img_kwargs']['input_nc'] == 207
# save with the reference's schema, reload through restore_checkpoint (latest and *_best_state)
p0 = trainer.model.box_net[0].weight.detach().clone()
path = trainer.save_checkpoint(checkpoint, 7, args, 1, (0.1, 1.5, 0.2, None), (0.2, 1.7, 0.3, None))
ck = torch.load(path, map_location='cpu', weights_only=False)
for k in ('model_state', 'optim_state', 'd_obj_state', 'd_obj_optim_state', 'd_mask_state', 'd_mask_optim_state',
          'd_img_state', 'd_img_optim_state', 'model_best_state', 'optim_best_state', 'd_obj_best_state',
          'd_mask_best_state', 'd_img_best_state', 'd_img_optim_best_state', 'best_t', 'counters', 'val_inception',
          'train_inception', 'checkpoint_ts', 'model_kwargs', 'd_obj_kwargs', 'd_mask_kwargs', 'd_img_kwargs', 'vocab', 'args'):
    assert k in ck, k
assert ck['counters'] == {'t': 7, 'epoch': 1} and ck['best_t'] == [7]
with torch.no_grad():
    trainer.model.box_net[0].weight.add_(1.0)
trainer.restore_checkpoint(ck, best=True)
assert torch.equal(trainer.model.box_net[0].weight, p0)
_tl = torch.load                   # train.py:127 predates torch 2.6's weights_only=True default (the file holds dicts)
torch.load = lambda *a, **k: _tl(*a, **dict(k, weights_only=False))
args2 = ns['get_args'](['--output_dir', %(out)r, '--restore_from_checkpoint', '1', '--image_size', '32,32'])
t2, epoch2, ck2 = ns['get_checkpoint'](args2, vocab)          # train.py:121-125 reads the file just written
assert t2 == 7 and epoch2 == 1
tr2 = ns['Trainer'](args2, vocab, ck2)
tr2.restore_checkpoint(ck2)
assert torch.equal(tr2.model.box_net[0].weight, p0)
trainer.generator_losses = None
print('DROPIN_OK')
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'scene_generation')), reason='reference checkout not present')
def test_install_as_runs_reference_train_py_imports_and_trainer_construction(tmp_path):
    code = SCRIPT % dict(root=ROOT, ref=REF, out=str(tmp_path / 'out'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS='4'))
    assert r.returncode == 0 and 'DROPIN_OK' in r.stdout, textwrap.shorten(r.stdout + r.stderr, 4000)


def test_install_as_without_host_checkout_only_overlays_hot_path(tmp_path):
    """On a box without the reference (the GPU box): the alias resolves the hot-path modules and leaves
    ``scene_generation.data`` absent instead of breaking them."""
    code = textwrap.dedent('''
        import sys
        sys.path = [p for p in sys.path if 'reference' not in p]
        sys.path.insert(0, %r)
        import scene_generation_amd
        scene_generation_amd.install_as('scene_generation', host_package_dir='')
        from scene_generation.model import Model
        from scene_generation.layout import masks_to_layout, boxes_to_layout
        from scene_generation.graph import GraphTripleConv, GraphTripleConvNet
        from scene_generation.losses import get_gan_losses, GANLoss, VGGLoss
        assert Model.__module__ == 'scene_generation_amd.model'
        for t in ('gan', 'wgan', 'lsgan'):
            get_gan_losses(t)
        print('OVERLAY_OK')
    ''' % ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and 'OVERLAY_OK' in r.stdout, r.stdout + r.stderr
