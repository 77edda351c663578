"""G+D training harness (surface of /root/reference/scene_generation/trainer.py:15-340 and the step driver
train.py:190-215) on the MI355X-native modules.

Same attributes and step functions as the reference Trainer; differences in HOW the step is issued:
  * four FusedAdam optimisers over flat parameter/gradient buffers (one launch per step / zero_grad)
  * while the generator trains, discriminator parameters are frozen (requires_grad False) so their weight
    gradients -- which the reference computes at trainer.py:262 and then discards at :277,298,323 -- are skipped
  * the image discriminator receives (layout, image) as two tensors; the concat is folded into its first conv
  * losses stay on the device (LossManager is lazy); nothing in a step forces a host sync except VectorPool's
    class-id copy
  * optional data parallelism: per-optimiser GradReducer (RCCL all-reduce of the flat gradient buffers)
TensorBoard / image logging of the reference (trainer.py:342-397) is glue outside the hot path: ``write_losses``
prints; checkpoints keep the reference schema (trainer.py:136-203, train.py:132-162).
"""
import contextlib
import os

import torch

from . import ops
from .discriminators import AcCropDiscriminator, define_mask_D, define_D
from .losses import get_gan_losses, GANLoss
from .model import Model
from .optim import FusedAdam
from .parallel import GradReducer, broadcast_params
from .utils import LossManager, weighted_sum


@contextlib.contextmanager
def _frozen(*modules):
    params = [p for m in modules if m is not None for p in m.parameters() if p.requires_grad]
    for p in params:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in params:
            p.requires_grad_(True)


class Trainer:
    def __init__(self, args, vocab, checkpoint=None, device='cuda', distributed=False, model_extra=None):
        self.vocab = vocab
        self.args = args
        self.device = device
        self.distributed = distributed
        self.num_obj = len(vocab['object_to_idx'])
        self.writer = None
        checkpoint = checkpoint if checkpoint is not None else {'model_kwargs': {}, 'd_obj_kwargs': {},
                                                               'd_mask_kwargs': {}, 'd_img_kwargs': {}}
        self.gan_g_loss, self.gan_d_loss = get_gan_losses(args.gan_loss_type)
        self._model_extra = model_extra or {}
        self.init_generator(args, checkpoint)
        self.init_image_discriminator(args, checkpoint)
        self.init_obj_discriminator(args, checkpoint)
        self.init_mask_discriminator(args, checkpoint)
        self.reducers = []
        if distributed:
            for opt in (self.optimizer, self.optimizer_d_img, self.optimizer_d_obj, self.optimizer_d_mask):
                if opt is not None:
                    broadcast_params(opt.fp)
                    r = GradReducer(opt.fp)
                    opt.pre_step_hooks.append(r.wait)
                    self.reducers.append(r)

    def _adam(self, module, lr):
        return FusedAdam(module.parameters(), lr=lr, betas=(self.args.beta1, 0.999))

    def init_generator(self, args, checkpoint):
        if args.restore_from_checkpoint:
            model_kwargs = checkpoint['model_kwargs']
        else:
            model_kwargs = {
                'vocab': self.vocab, 'image_size': args.image_size, 'embedding_dim': args.embedding_dim,
                'gconv_dim': args.gconv_dim, 'gconv_hidden_dim': args.gconv_hidden_dim,
                'gconv_num_layers': args.gconv_num_layers, 'mlp_normalization': args.mlp_normalization,
                'appearance_normalization': args.appearance_normalization, 'activation': args.activation,
                'mask_size': args.mask_size, 'n_downsample_global': args.n_downsample_global,
                'box_dim': args.box_dim, 'use_attributes': args.use_attributes, 'box_noise_dim': args.box_noise_dim,
                'mask_noise_dim': args.mask_noise_dim, 'pool_size': args.pool_size, 'rep_size': args.rep_size,
            }
            model_kwargs.update(self._model_extra)
            checkpoint['model_kwargs'] = model_kwargs
        self.model = Model(**model_kwargs).to(self.device)
        if args.vgg_features_weight > 0:
            raise NotImplementedError('VGG feature loss needs pretrained VGG19 weights (not available offline); '
                                      'run with --vgg_features_weight 0')
        self.criterionVGG = None
        self.criterionGAN = GANLoss(use_lsgan=not args.no_lsgan)
        self.optimizer = self._adam(self.model, args.learning_rate)

    def init_obj_discriminator(self, args, checkpoint):
        self.obj_discriminator, self.optimizer_d_obj = None, None
        if args.d_obj_weight > 0:
            if args.restore_from_checkpoint:
                d_obj_kwargs = checkpoint['d_obj_kwargs']
            else:
                d_obj_kwargs = {'vocab': self.vocab, 'arch': args.d_obj_arch, 'normalization': args.d_normalization,
                                'activation': args.d_activation, 'padding': args.d_padding,
                                'object_size': args.crop_size}
                checkpoint['d_obj_kwargs'] = d_obj_kwargs
            self.obj_discriminator = AcCropDiscriminator(**d_obj_kwargs).to(self.device)
            self.obj_discriminator.train()
            self.optimizer_d_obj = self._adam(self.obj_discriminator, args.learning_rate)

    def init_mask_discriminator(self, args, checkpoint):
        self.mask_discriminator, self.optimizer_d_mask = None, None
        if args.d_mask_weight > 0:
            if args.restore_from_checkpoint:
                d_mask_kwargs = checkpoint['d_mask_kwargs']
            else:
                d_mask_kwargs = {'input_nc': 1, 'ndf': args.ndf_mask, 'n_layers_D': args.n_layers_D_mask,
                                 'norm': args.norm_D_mask, 'use_sigmoid': args.no_lsgan, 'num_D': args.num_D_mask,
                                 'num_objects': self.num_obj}
                checkpoint['d_mask_kwargs'] = d_mask_kwargs
            self.mask_discriminator = define_mask_D(**d_mask_kwargs).to(self.device)
            self.mask_discriminator.train()
            self.optimizer_d_mask = self._adam(self.mask_discriminator, args.mask_learning_rate)

    def init_image_discriminator(self, args, checkpoint):
        if args.d_img_weight == 0:
            self.netD, self.optimizer_d_img = None, None
            return
        if args.restore_from_checkpoint:
            d_img_kwargs = checkpoint['d_img_kwargs']
        else:
            d_img_kwargs = {'input_nc': self.num_obj + args.rep_size + args.output_nc, 'ndf': args.ndf,
                            'n_layers_D': args.n_layers_D, 'norm': args.norm_D, 'use_sigmoid': args.no_lsgan,
                            'num_D': args.num_D}
            checkpoint['d_img_kwargs'] = d_img_kwargs
        self.netD = define_D(**d_img_kwargs).to(self.device)
        self.netD.train()
        self.optimizer_d_img = self._adam(self.netD, args.learning_rate)

    # ---- checkpoints (reference schema) ----
    def restore_checkpoint(self, checkpoint):
        self.model.load_state_dict(checkpoint['model_state'])
        self.optimizer.load_state_dict(checkpoint['optim_state'])
        if self.obj_discriminator is not None:
            self.obj_discriminator.load_state_dict(checkpoint['d_obj_state'])
            self.optimizer_d_obj.load_state_dict(checkpoint['d_obj_optim_state'])
        if self.mask_discriminator is not None:
            self.mask_discriminator.load_state_dict(checkpoint['d_mask_state'])
            self.optimizer_d_mask.load_state_dict(checkpoint['d_mask_optim_state'])
        if self.netD is not None:
            self.netD.load_state_dict(checkpoint['d_img_state'])
            self.optimizer_d_img.load_state_dict(checkpoint['d_img_optim_state'])

    def save_checkpoint(self, checkpoint, t, args, epoch, train_results=None, val_results=None):
        if self.obj_discriminator is not None:
            checkpoint['d_obj_state'] = self.obj_discriminator.state_dict()
            checkpoint['d_obj_optim_state'] = self.optimizer_d_obj.state_dict()
        if self.mask_discriminator is not None:
            checkpoint['d_mask_state'] = self.mask_discriminator.state_dict()
            checkpoint['d_mask_optim_state'] = self.optimizer_d_mask.state_dict()
        if self.netD is not None:
            checkpoint['d_img_state'] = self.netD.state_dict()
            checkpoint['d_img_optim_state'] = self.optimizer_d_img.state_dict()
        checkpoint['model_state'] = self.model.state_dict()
        checkpoint['optim_state'] = self.optimizer.state_dict()
        checkpoint.setdefault('counters', {})['t'] = t
        checkpoint['counters']['epoch'] = epoch
        path = os.path.join(args.output_dir, '%s_with_model.pt' % args.checkpoint_name)
        os.makedirs(args.output_dir, exist_ok=True)
        torch.save(checkpoint, path)
        return path

    # ---- step functions ----
    def train_generator(self, imgs, imgs_pred, masks, masks_pred, layout, objs, boxes, boxes_pred, obj_to_img,
                        use_gt):
        """trainer.py:205-263.  The mask / image discriminator forwards recorded here are the same computation the
        discriminator steps repeat on detached inputs (same weights: no discriminator is updated in between,
        train.py:205-215; InstanceNorm has no running statistics), so they are kept in ``self._shared`` and the D steps
        back-propagate through them instead of re-running them.  Their parameters are excluded from THIS backward
        (the reference computes and then discards those gradients, trainer.py:262 vs :298,323).  The object
        discriminator has BatchNorm running statistics that every forward updates, so it is not shared."""
        args = self.args
        self.generator_losses = L = LossManager()
        self._shared = shared = {}
        share = getattr(self, 'share_d_forward', True)
        d_shared = [p for m in (self.mask_discriminator, self.netD) if m is not None for p in m.parameters()]
        with _frozen(self.obj_discriminator), ops.skip_param_grads(d_shared):
            if use_gt:
                if args.l1_pixel_loss_weight > 0:
                    L.add_loss(ops.l1(imgs_pred, imgs), 'L1_pixel_loss', args.l1_pixel_loss_weight)
                L.add_loss(ops.mse(boxes_pred, boxes), 'bbox_pred', args.bbox_pred_loss_weight)

            scores_fake, ac_loss, g_fake_crops = self.obj_discriminator(imgs_pred, objs, boxes, obj_to_img)
            L.add_loss(ac_loss, 'ac_loss', args.ac_loss_weight)
            L.add_loss(self.gan_g_loss(scores_fake), 'g_gan_obj_loss', args.d_obj_weight)

            if self.mask_discriminator is not None:
                one_hot_obj = ops.one_hot(objs, self.num_obj)
                scores_fake = self.mask_discriminator(masks_pred.unsqueeze(1), one_hot_obj)
                if share:
                    shared['mask_fake'] = scores_fake
                L.add_loss(self.criterionGAN(scores_fake, True), 'g_gan_mask_obj_loss', args.d_mask_weight)
                if args.d_mask_features_weight > 0:
                    with (contextlib.nullcontext() if share else torch.no_grad()):
                        scores_real = self.mask_discriminator(masks.float().unsqueeze(1), one_hot_obj)
                    if share:
                        shared['mask_real'] = scores_real
                    L.add_loss(self.calculate_features_loss(scores_fake, scores_real), 'g_mask_features_loss',
                               args.d_mask_features_weight)      # real features enter detached (trainer.py:339)

            if self.netD is not None:
                lay = ops.detach_keep(layout)       # no gradient reaches the layout through D (trainer.py:246-248)
                img_pred_fake = self.netD(lay, imgs_pred)
                if share:
                    shared['img_fake'] = img_pred_fake
                L.add_loss(self.criterionGAN(img_pred_fake, True), 'g_gan_img_loss', args.d_img_weight)
                if args.d_img_features_weight > 0:
                    with (contextlib.nullcontext() if share else torch.no_grad()):
                        pred_real = self.netD(lay, imgs)        # "train textures" pass
                    if share:
                        shared['img_real'] = pred_real
                    L.add_loss(self.calculate_features_loss(img_pred_fake, pred_real), 'g_gan_features_loss_img',
                               args.d_img_features_weight)

            L.set_value('total_loss', L.total_loss)
            self.optimizer.zero_grad()
            L.total_loss.backward(retain_graph=bool(shared))
        self.optimizer.step()

    def train_obj_discriminator(self, imgs, imgs_pred, objs, boxes, boxes_pred, obj_to_img):
        if self.obj_discriminator is not None:
            self.d_obj_losses = L = LossManager()
            scores_fake, ac_loss_fake, self.d_fake_crops = self.obj_discriminator(imgs_pred, objs, boxes_pred,
                                                                                  obj_to_img)
            scores_real, ac_loss_real, self.d_real_crops = self.obj_discriminator(imgs, objs, boxes, obj_to_img)
            L.add_loss(self.gan_d_loss(scores_real, scores_fake), 'd_obj_gan_loss', 0.5)
            L.add_loss(ac_loss_real, 'd_ac_loss_real')
            L.add_loss(ac_loss_fake, 'd_ac_loss_fake')
            self.optimizer_d_obj.zero_grad()
            L.total_loss.backward()
            self.optimizer_d_obj.step()

    def train_mask_discriminator(self, masks, masks_pred, objs):
        if self.mask_discriminator is not None:
            self.d_mask_losses = L = LossManager()
            shared = getattr(self, '_shared', {})
            one_hot_obj = ops.one_hot(objs, self.num_obj)
            scores_fake = shared.pop('mask_fake', None)
            if scores_fake is None:
                scores_fake = self.mask_discriminator(masks_pred.unsqueeze(1), one_hot_obj)
            scores_real = shared.pop('mask_real', None)
            if scores_real is None:
                scores_real = self.mask_discriminator(masks.float().unsqueeze(1), one_hot_obj)
            L.add_loss(self.criterionGAN(scores_fake, False), 'fake_loss', 0.5)
            L.add_loss(self.criterionGAN(scores_real, True), 'real_loss', 0.5)
            self.optimizer_d_mask.zero_grad()
            # inputs=: a shared forward still hangs off the generator's graph; only the discriminator leaves are wanted
            torch.autograd.backward(L.total_loss, inputs=list(self.mask_discriminator.parameters()))
            self.optimizer_d_mask.step()

    def train_image_discriminator(self, imgs, imgs_pred, layout, layout_wrong):
        if self.netD is not None:
            self.d_img_losses = L = LossManager()
            shared = getattr(self, '_shared', {})
            alpha = (1 / 2) * (.5)
            pred_fake = shared.pop('img_fake', None)
            if pred_fake is None:
                pred_fake = self.discriminate(layout, imgs_pred)
            pred_real = shared.pop('img_real', None)
            if pred_real is None:
                pred_real = self.discriminate(layout, imgs)
            L.add_loss(self.criterionGAN(pred_fake, False), 'fake_image_loss', alpha)
            L.add_loss(self.criterionGAN(self.discriminate(layout_wrong, imgs), False), 'wrong_texture_loss', alpha)
            L.add_loss(self.criterionGAN(pred_real, True), 'd_img_gan_real_loss', 0.5)
            self.optimizer_d_img.zero_grad()
            torch.autograd.backward(L.total_loss, inputs=list(self.netD.parameters()))
            self.optimizer_d_img.step()
            self._shared = {}

    def discriminate(self, input_label, test_image):
        return self.netD(input_label, test_image)        # cat((label, image), 1) folded into the first conv

    def calculate_features_loss(self, pred_fake, pred_real):
        """trainer.py:331-340."""
        nums_d = len(pred_fake)
        feat_weights = 4.0 / len(pred_fake[0])
        D_weights = 1.0 / nums_d
        terms = [ops.l1(pred_fake[i][j], pred_real[i][j].detach())
                 for i in range(nums_d) for j in range(len(pred_fake[i]) - 1)]
        return weighted_sum(terms, [D_weights * feat_weights] * len(terms))

    def step(self, batch, use_gt=True):
        """One full G+D iteration = train.py:190-215.  ``batch`` = the 8-tuple of coco_collate_fn on the device."""
        imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img, attributes = batch
        if not use_gt:
            attributes = torch.zeros_like(attributes)
        model_out = self.model(imgs, objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=masks,
                               attributes=attributes)
        imgs_pred, boxes_pred, masks_pred, layout, layout_pred, layout_wrong = model_out
        self.train_generator(imgs, imgs_pred, masks, masks_pred, layout, objs, boxes, boxes_pred, obj_to_img, use_gt)
        self.train_mask_discriminator(masks, masks_pred.detach(), objs)
        self.train_obj_discriminator(imgs, imgs_pred.detach(), objs, boxes, boxes.detach(), obj_to_img)
        self.train_image_discriminator(imgs, imgs_pred.detach(), ops.detach_keep(layout), ops.detach_keep(layout_wrong))
        return model_out

    def write_losses(self, checkpoint, t):
        print('t = %d / %d' % (t, self.args.num_iterations))
        for tag, L in (('G', getattr(self, 'generator_losses', None)), ('D_obj', getattr(self, 'd_obj_losses', None)),
                       ('D_mask', getattr(self, 'd_mask_losses', None)), ('D_img', getattr(self, 'd_img_losses', None))):
            if L is None:
                continue
            for name, val in L.items():
                print(' %s [%s]: %.4f' % (tag, name, val))
                if checkpoint is not None:
                    key = 'losses' if tag == 'G' else 'd_losses'
                    checkpoint.setdefault(key, {}).setdefault(name, []).append(val)
