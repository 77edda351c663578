"""G+D training harness (surface of /root/reference/scene_generation/trainer.py:15-340 and the step driver
train.py:190-215) on the MI355X-native modules.

Same attributes and step functions as the reference Trainer; differences in HOW the step is issued:
  * four FusedAdam optimisers over flat parameter/gradient buffers (one launch per step / zero_grad)
  * while the generator trains, discriminator parameters are frozen (requires_grad False) so their weight
    gradients -- which the reference computes at trainer.py:262 and then discards at :277,298,323 -- are skipped
  * the image discriminator receives (layout, image) as two tensors; the concat is folded into its first conv
  * losses stay on the device (LossManager is lazy); nothing in a step forces a host sync except VectorPool's
    class-id copy
  * the image discriminator's real and wrong-texture passes (trainer.py:250,304-308) run as ONE 2N batch over the stacked
    factored layouts (_real_and_wrong_pass): the launches of the discriminators are occupancy-starved at N = 32
  * optional data parallelism: per-optimiser GradReducer (RCCL all-reduce of the flat gradient buffers)
  * optional host-side flow control (``max_lead_steps`` / SG_LEAD_STEPS, default off): see ``_lead_point``
TensorBoard / image logging of the reference (trainer.py:342-397) is glue outside the hot path: ``write_losses``
prints; checkpoints keep the reference schema (trainer.py:136-203, train.py:132-162).
"""
import contextlib
import os

import torch

from . import ops
from .discriminators import AcCropDiscriminator, define_mask_D, define_D
from .losses import get_gan_losses, GANLoss, VGGLoss
from .model import Model
from . import optim, streams
from .optim import FusedAdam
from .parallel import GradReducer, broadcast_params, broadcast_int, control_group
from .utils import LossManager, respect_cpu_quota, weighted_sum


class _NullWriter(object):
    """stand-in for tensorboardX.SummaryWriter (trainer.py:21) when tensorboardX is not installed: logging is glue"""

    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass


def _make_writer(args):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(args.output_dir)
    except Exception:
        return _NullWriter()


def _make_grid(imgs, pad=2):
    """torchvision.utils.make_grid(imgs, normalize=True, scale_each=True) for the TensorBoard panels (host side)"""
    imgs = imgs.detach().float().cpu()
    if imgs.size(1) == 1:
        imgs = imgs.expand(-1, 3, -1, -1)
    lo = imgs.amin(dim=(1, 2, 3), keepdim=True)
    hi = imgs.amax(dim=(1, 2, 3), keepdim=True)
    imgs = (imgs - lo) / (hi - lo).clamp(min=1e-5)
    n, c, h, w = imgs.shape
    cols = min(8, n)
    rows = (n + cols - 1) // cols
    grid = torch.zeros(c, rows * (h + pad) + pad, cols * (w + pad) + pad)
    for i in range(n):
        r, q = divmod(i, cols)
        grid[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + q * (w + pad): pad + q * (w + pad) + w] = imgs[i]
    return grid


def _has_batchnorm(module):
    return module is not None and any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in module.modules())


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
    def __init__(self, args, vocab, checkpoint=None, device=None, distributed=False, model_extra=None):
        self.vocab = vocab
        self.args = args
        respect_cpu_quota()          # host threads <= the container's CPU quota: an OpenMP burst must not freeze the launch thread
        # the reference moves everything to 'cuda' (trainer.py:54,77,103,130).  Without a device the object can still be
        # CONSTRUCTED (state_dict surgery, checkpoint conversion); any forward raises: there is no CPU compute path
        self.device = device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu')
        self.distributed = distributed
        self.num_obj = len(vocab['object_to_idx'])
        self.writer = _make_writer(args)                                  # trainer.py:21
        self.colors = torch.randint(0, 256, [self.num_obj, 3]).float()    # trainer.py:22 (consumes the same RNG draws)
        checkpoint = checkpoint if checkpoint is not None else {'model_kwargs': {}, 'd_obj_kwargs': {},
                                                               'd_mask_kwargs': {}, 'd_img_kwargs': {}}
        self.gan_g_loss, self.gan_d_loss = get_gan_losses(args.gan_loss_type)
        self._model_extra = model_extra or {}
        self.init_generator(args, checkpoint)
        self.init_image_discriminator(args, checkpoint)
        self.init_obj_discriminator(args, checkpoint)
        self.init_mask_discriminator(args, checkpoint)
        # Sharing the mask / image discriminator forwards between the generator step and the discriminator steps is only
        # exact for discriminators WITHOUT BatchNorm: a BatchNorm forward updates running statistics, and the reference
        # runs it once more per discriminator step (trainer.py:281-325)
        self._shareable = {'mask': not _has_batchnorm(self.mask_discriminator), 'img': not _has_batchnorm(self.netD)}
        self.share_d_forward = True
        # the real and wrong-texture passes of the image discriminator as one 2N batch (_real_and_wrong_pass; A/B: SG_BATCH_REAL_WRONG=0)
        self.batch_real_wrong = os.environ.get('SG_BATCH_REAL_WRONG', '1') != '0'
        # how many iterations the launching thread may run ahead of the GPU (0 = the runtime's own limit: ~2.3 iterations)
        self.max_lead_steps = int(os.environ.get('SG_LEAD_STEPS', '0'))
        self._lead_events = {}
        self.reducers = []
        if distributed:
            control_group()                      # collective: create the host-side agreement group on every rank now
            for opt in (self.optimizer, self.optimizer_d_img, self.optimizer_d_obj, self.optimizer_d_mask):
                if opt is not None:
                    broadcast_params(opt.fp)
                    # every generator parameter is used once per step, so its gradient is final when it is delivered and
                    # the bucket all-reduces overlap the backward; discriminator parameters collect several
                    # contributions per step (fake / real / wrong passes): their one bucket (<= 24 MB each) leaves when
                    # their backward has ended (flush() in _finish_d_step) and runs under the steps that follow
                    r = GradReducer(opt.fp, optimizer=opt, overlap=opt is self.optimizer)
                    opt.use_spill = not r.overlap      # (spilled contributions are folded in after the buckets have left)
                    opt.grad_listeners.append(r.param_ready)
                    opt.late_listeners.append(r.late_contribution)
                    opt.zero_grad_hooks.append(r.begin_step)
                    # the 1 / world of the mean is applied by the Adam kernel while it reads the gradient (no scaling pass)
                    opt.pre_step_hooks.append(r.wait_deferred_scale)
                    self.reducers.append(r)

    def _lead_point(self, k):
        """Host-side flow control (opt-in), called at fixed places of an iteration: wait until the GPU has passed THIS place of
        the iteration ``max_lead_steps`` before the current one, then mark the place in the stream.

        Left alone the launching thread runs ~2.3 iterations ahead (it needs 11-13 ms to issue what the GPU executes in 32 ms)
        until a launch blocks inside the runtime for 15-20 ms -- a per-queue resource of ~1 300 eager launches is handed back a
        quarter at a time.  That costs nothing (the GPU always has > 1.5 iterations queued; 32.02 vs 32.05 ms/step with the
        bound at one iteration, profiles/r06_host_stall.md), so the bound is OFF by default; it exists as an instrument: with it
        no launch ever blocks, every wait is one ``Event.synchronize`` at a known place (interpreter lock released), which is
        what separated the runtime's back-pressure from the stall round 6 was hunting (a CPU-quota freeze of the whole process
        caused by an OpenMP burst in the input staging: scene_generation_amd/pipeline.py).  Host-only: nothing is added to the
        stream except an event record."""
        n = self.max_lead_steps
        if n <= 0 or torch.device(self.device).type != 'cuda' or torch.cuda.is_current_stream_capturing():
            return
        q = self._lead_events.setdefault(k, [])
        if len(q) >= n:
            q.pop(0).synchronize()
        ev = torch.cuda.Event()
        ev.record()
        q.append(ev)

    def _adam(self, module, lr, join_exclude=()):
        opt = FusedAdam(module.parameters(), lr=lr, betas=(self.args.beta1, 0.999), lazy_zero=optim.LAZY_ZERO)
        opt.join_exclude = tuple(join_exclude)
        return opt

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
        self.criterionVGG = None
        if args.vgg_features_weight > 0:             # trainer.py:57; ImageNet weights via --vgg_weights <state_dict path>
            vgg_weights = getattr(args, 'vgg_weights', None) or None
            if vgg_weights is None:
                import sys
                print('scene_generation_amd.Trainer: WARNING -- --vgg_features_weight %g without --vgg_weights: the perceptual '
                      'loss runs on a RANDOM-INIT VGG19 (torchvision ImageNet weights cannot be downloaded here).  Same '
                      'arithmetic and cost as the reference objective, NOT its perceptual meaning; pass --vgg_weights '
                      '<torchvision vgg19 state_dict> for real training or --vgg_features_weight 0 to drop the term.'
                      % args.vgg_features_weight, file=sys.stderr)
            self.criterionVGG = VGGLoss(weights=vgg_weights).to(self.device)
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
            self.optimizer_d_obj = self._adam(self.obj_discriminator, args.learning_rate, join_exclude=('front',))

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
            # (group 'mstep': kernels of the front's stream write this optimiser's gradients -- it waits for that stream like the
            #  generator's, whether the group is on at construction time or switched on later)
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
        self.optimizer_d_img = self._adam(self.netD, args.learning_rate, join_exclude=('front',))

    # ---- checkpoints (reference schema: trainer.py:136-203, train.py:119-163) ----
    def restore_checkpoint(self, checkpoint, best=False):
        """trainer.py:136-150.  ``best=True`` loads the ``*_best_state`` entries (the ones scripts/sample_images.py
        reads) instead of the latest ones."""
        k = (lambda name: name.replace('_state', '_best_state')) if best else (lambda name: name)
        self.model.load_state_dict(checkpoint[k('model_state')])
        self.optimizer.load_state_dict(checkpoint[k('optim_state')])
        if self.obj_discriminator is not None:
            self.obj_discriminator.load_state_dict(checkpoint[k('d_obj_state')])
            self.optimizer_d_obj.load_state_dict(checkpoint[k('d_obj_optim_state')])
        if self.mask_discriminator is not None:
            self.mask_discriminator.load_state_dict(checkpoint[k('d_mask_state')])
            self.optimizer_d_mask.load_state_dict(checkpoint[k('d_mask_optim_state')])
        if self.netD is not None:
            self.netD.load_state_dict(checkpoint[k('d_img_state')])
            self.optimizer_d_img.load_state_dict(checkpoint[k('d_img_optim_state')])

    def save_checkpoint(self, checkpoint, t, args, epoch, train_results=None, val_results=None):
        """trainer.py:152-203: same keys, same "best" bookkeeping (best = highest validation inception mean)."""
        index = int(t / args.print_every)
        val_inception_mean = None
        if train_results is not None:
            t_avg_iou, t_inception_mean, t_inception_std = train_results[:3]
            self.writer.add_scalar('checkpoint/train_iou', t_avg_iou, index)
            self.writer.add_scalar('checkpoint/train_inception_mean', t_inception_mean, index)
            self.writer.add_scalar('checkpoint/train_inception_std', t_inception_std, index)
            checkpoint.setdefault('checkpoint_ts', []).append(t)
            checkpoint.setdefault('train_inception', []).append(t_inception_mean)
        if val_results is not None:
            val_avg_iou, val_inception_mean, val_inception_std = val_results[:3]
            self.writer.add_scalar('checkpoint/val_iou', val_avg_iou, index)
            self.writer.add_scalar('checkpoint/val_inception_mean', val_inception_mean, index)
            self.writer.add_scalar('checkpoint/val_inception_std', val_inception_std, index)
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
        if val_inception_mean is not None:
            history = checkpoint.setdefault('val_inception', [])
            history.append(val_inception_mean)
            # quirk kept: the reference appends BEFORE comparing (trainer.py:168,188), so max(history) is
            # never below the new value and only the first checkpoint ever becomes "best"
            if len(checkpoint.setdefault('best_t', [])) == 0 or max(history) < val_inception_mean:
                checkpoint['best_t'].append(t)
                for name in ('d_obj', 'd_mask', 'd_img'):
                    checkpoint[name + '_best_state'] = checkpoint.get(name + '_state')
                    checkpoint[name + '_optim_best_state'] = checkpoint.get(name + '_optim_state')
                checkpoint['model_best_state'] = checkpoint['model_state']
                checkpoint['optim_best_state'] = checkpoint['optim_state']
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
        self._lead_point(0)
        self.generator_losses = L = LossManager()
        self._shared = shared = {}
        share = getattr(self, 'share_d_forward', True)
        share_mask, share_img = share and self._shareable['mask'], share and self._shareable['img']
        d_shared = [p for m in (self.mask_discriminator, self.netD) if m is not None for p in m.parameters()]
        # group 'mstep' (streams.py): the mask discriminator's part of this step -- two forwards over O 16x16 masks, its losses, and
        # in the backward its data gradients: small launches, fed by masks_pred only -- continues the object front's side stream
        # beside the object / image discriminator work on the current stream (joined by ``fk.join()`` below, before the total is
        # formed)
        with _frozen(self.obj_discriminator), ops.skip_param_grads(d_shared), \
                streams.fork(imgs_pred.device, 'front', enabled=streams.group_on('mstep')) as fk:
            if use_gt:
                if args.l1_pixel_loss_weight > 0:
                    L.add_loss(ops.l1(imgs_pred, imgs), 'L1_pixel_loss', args.l1_pixel_loss_weight)
                L.add_loss(ops.mse(boxes_pred, boxes), 'bbox_pred', args.bbox_pred_loss_weight)

            if self.criterionVGG is not None:            # trainer.py:218-221
                L.add_loss(self.criterionVGG(imgs_pred, imgs), 'g_vgg', args.vgg_features_weight)

            fo = streams.fork(imgs_pred.device, 'objD').__enter__()      # joined below, before the total is formed
            with fo.branch(1, reads=(imgs_pred, objs, boxes, obj_to_img)):
                scores_fake, ac_loss, g_fake_crops = self.obj_discriminator(imgs_pred, objs, boxes, obj_to_img)
                g_obj = self.gan_g_loss(scores_fake)
                fo.produced((scores_fake, ac_loss, g_fake_crops, g_obj))
            L.add_loss(ac_loss, 'ac_loss', args.ac_loss_weight)
            L.add_loss(g_obj, 'g_gan_obj_loss', args.d_obj_weight)

            if self.mask_discriminator is not None:
              with fk.branch(1, reads=(masks_pred, masks, objs)):
                one_hot_obj = ops.one_hot(objs, self.num_obj)
                scores_fake = self.mask_discriminator(masks_pred.unsqueeze(1), one_hot_obj)
                if share_mask:
                    shared['mask_fake'] = scores_fake
                g_mask = self.criterionGAN(scores_fake, True)
                L.add_loss(g_mask, 'g_gan_mask_obj_loss', args.d_mask_weight)
                fk.produced((scores_fake, g_mask))
                if args.d_mask_features_weight > 0:
                    with (contextlib.nullcontext() if share_mask else torch.no_grad()):
                        scores_real = self.mask_discriminator(masks.float().unsqueeze(1), one_hot_obj)
                    if share_mask:
                        shared['mask_real'] = scores_real
                    g_mfeat = self.calculate_features_loss(scores_fake, scores_real)
                    L.add_loss(g_mfeat, 'g_mask_features_loss',
                               args.d_mask_features_weight)      # real features enter detached (trainer.py:339)
                    fk.produced((scores_real, g_mfeat))

            if self.netD is not None:
                lay = layout.detach()               # no gradient reaches the layout through D (trainer.py:246-248)
                img_pred_fake = self.netD(lay, imgs_pred)
                if share_img:
                    shared['img_fake'] = img_pred_fake
                L.add_loss(self.criterionGAN(img_pred_fake, True), 'g_gan_img_loss', args.d_img_weight)
                if args.d_img_features_weight > 0:
                    pred_real = self._real_and_wrong_pass(lay, imgs, shared) if share_img else None
                    if pred_real is None:
                        with (contextlib.nullcontext() if share_img else torch.no_grad()):
                            pred_real = self.netD(lay, imgs)        # "train textures" pass
                        if share_img:
                            shared['img_real'] = pred_real
                    L.add_loss(self.calculate_features_loss(img_pred_fake, pred_real), 'g_gan_features_loss_img',
                               args.d_img_features_weight)

            fk.join()                   # the weighted sum below reads loss terms the side branches produced
            fo.join()
            L.set_value('total_loss', L.total_loss)
            self.optimizer.zero_grad()
            L.total_loss.backward(retain_graph=bool(shared))
        self._step_or_defer(self.optimizer)

    def _step_or_defer(self, opt):
        """``opt.step()`` -- or, data parallel inside Trainer.step: send the remaining buckets of its gradient to the all-reduce
        now (no wait) and take the Adam step at the end of the iteration.  Nothing between here and there reads the
        parameters ``opt`` owns (the generator's: every later sub-step consumes tensors computed before; a discriminator's: the
        generator step has already used it, the other discriminator steps never do), so the result is the same while the
        collective runs under the sub-steps that follow instead of stalling the stream.  The collectives are issued in
        program order, the same on every rank."""
        if getattr(self, '_defer_g_step', False):
            for r in self.reducers:
                if r.optimizer is opt:
                    r.flush()
            self._deferred_steps.append(opt)
        elif getattr(self, '_adam_side', False) and opt is self.optimizer:
            # Inside Trainer.step, one GPU: the generator's Adam step -- 0.6 ms of pure HBM streaming over 5 GB of parameters,
            # gradients and moments -- runs on a side stream under the discriminator sub-steps that follow (MFMA-bound GEMMs;
            # none of them reads a generator parameter or writes a generator gradient: see the docstring).  The stream waits for
            # the backward (main + the object front, FusedAdam.step's join_all); Trainer.step joins it before it returns.
            dev = opt.fp.flat.device
            side = streams.side_stream(dev, 'adam', 1)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                opt.step()
            self._side_pending.append(side)
        else:
            opt.step()

    def _real_and_wrong_pass(self, lay, imgs, shared):
        """(layout, real images) and (wrong-texture layout, real images) -- the "real" and "wrong" passes of the image
        discriminator (trainer.py:250,304-308) -- as ONE forward over 2N images: the two layouts share planes and objects
        (model.py:119-124), InstanceNorm is per sample, so the batch halves are exactly the two passes; every layer then runs one
        launch twice the size instead of two (these launches are occupancy-starved at N = 32) and the discriminator step gets
        the weight gradients of both passes from one GEMM per layer.  Needs the factored layouts (``wrong_twin`` hint set by
        Model.forward) and ``share_d_forward``; returns the real pass (list of feature lists) or None when not applicable."""
        if not getattr(self, 'batch_real_wrong', True) or not ops.FACTORED_LAYOUT:
            return None
        f_gt, f_wrong = ops.hint(lay, 'factored'), ops.hint(lay, 'wrong_twin')
        if f_gt is None or f_wrong is None or f_gt.Z is not f_wrong.Z:
            return None
        N = imgs.size(0)
        f2 = ops.FactoredLayout.stacked(f_gt, f_wrong)
        ghost = lay.new_empty((1,)).expand((2 * N,) + tuple(lay.shape[1:]))      # never read: the convs run on the factored form
        both = self.netD(ops.set_hints(ghost, factored=f2), torch.cat([imgs, imgs], 0))
        shared['img_real'] = [[t[:N] for t in scale] for scale in both]
        shared['img_wrong'] = [[t[N:] for t in scale] for scale in both]
        return shared['img_real']

    def train_obj_discriminator(self, imgs, imgs_pred, objs, boxes, boxes_pred, obj_to_img):
        if self.obj_discriminator is not None:
            # group 'objD' (opt-in), inside Trainer.step: the whole sub-step on the object discriminator's side stream, beside the
            # image discriminator's sub-step
            if getattr(self, '_side_ok', False) and streams.group_on('objD') and imgs.is_cuda:
                main = torch.cuda.current_stream(imgs.device)
                side = streams.side_stream(imgs.device, 'objD', 1)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._train_obj_discriminator(imgs, imgs_pred, objs, boxes, boxes_pred, obj_to_img)
                    for t in (self.d_fake_crops, self.d_real_crops, self.d_obj_losses.total_loss):
                        t.record_stream(main)
                self._side_pending.append(side)
            else:
                self._train_obj_discriminator(imgs, imgs_pred, objs, boxes, boxes_pred, obj_to_img)

    def _train_obj_discriminator(self, imgs, imgs_pred, objs, boxes, boxes_pred, obj_to_img):
        if True:
            self.d_obj_losses = L = LossManager()
            scores_fake, ac_loss_fake, self.d_fake_crops = self.obj_discriminator(imgs_pred, objs, boxes_pred,
                                                                                  obj_to_img)
            scores_real, ac_loss_real, self.d_real_crops = self.obj_discriminator(imgs, objs, boxes, obj_to_img)
            L.add_loss(self.gan_d_loss(scores_real, scores_fake), 'd_obj_gan_loss', 0.5)
            L.add_loss(ac_loss_real, 'd_ac_loss_real')
            L.add_loss(ac_loss_fake, 'd_ac_loss_fake')
            self.optimizer_d_obj.zero_grad()
            L.total_loss.backward()
            self._step_or_defer(self.optimizer_d_obj)

    def train_mask_discriminator(self, masks, masks_pred, objs):
        if self.mask_discriminator is not None:
            # inside Trainer.step, one GPU, group 'mstep': the whole sub-step (losses, backward through the shared forwards --
            # recorded on the front's stream, so autograd runs them there anyway --, Adam) on that stream, beside the object /
            # image discriminator sub-steps that follow on the current one; Trainer.step joins the stream before it returns
            if getattr(self, '_side_ok', False) and streams.group_on('mstep') and streams.group_on('front') and masks_pred.is_cuda:
                main = torch.cuda.current_stream(masks_pred.device)
                side = streams.side_stream(masks_pred.device, 'front', 1)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._train_mask_discriminator(masks, masks_pred, objs)
                    self.d_mask_losses.total_loss.record_stream(main)
                self._side_pending.append(side)
            else:
                self._train_mask_discriminator(masks, masks_pred, objs)

    def _train_mask_discriminator(self, masks, masks_pred, objs):
        if True:
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
            self._step_or_defer(self.optimizer_d_mask)

    def train_image_discriminator(self, imgs, imgs_pred, layout, layout_wrong):
        if self.netD is not None:
            self._lead_point(1)
            self.d_img_losses = L = LossManager()
            shared = getattr(self, '_shared', {})
            alpha = (1 / 2) * (.5)
            pred_fake = shared.pop('img_fake', None)
            if pred_fake is None:
                pred_fake = self.discriminate(layout, imgs_pred)
            pred_real = shared.pop('img_real', None)
            if pred_real is None:
                pred_real = self.discriminate(layout, imgs)
            pred_wrong = shared.pop('img_wrong', None)      # batched with the real pass in the generator step, when possible
            if pred_wrong is None:
                pred_wrong = self.discriminate(layout_wrong, imgs)
            L.add_loss(self.criterionGAN(pred_fake, False), 'fake_image_loss', alpha)
            L.add_loss(self.criterionGAN(pred_wrong, False), 'wrong_texture_loss', alpha)
            L.add_loss(self.criterionGAN(pred_real, True), 'd_img_gan_real_loss', 0.5)
            self.optimizer_d_img.zero_grad()
            torch.autograd.backward(L.total_loss, inputs=list(self.netD.parameters()))
            self._step_or_defer(self.optimizer_d_img)
            self._shared = {}

    def discriminate(self, input_label, test_image):
        return self.netD(input_label, test_image)        # cat((label, image), 1) folded into the first conv

    def calculate_features_loss(self, pred_fake, pred_real):
        """trainer.py:331-340."""
        nums_d = len(pred_fake)
        feat_weights = 4.0 / len(pred_fake[0])
        D_weights = 1.0 / nums_d
        pairs = [(pred_fake[i][j], pred_real[i][j].detach()) for i in range(nums_d) for j in range(len(pred_fake[i]) - 1)]
        if pairs and pairs[0][0].is_cuda:                 # every term and the weighted sum in ONE launch (ops.MultiLossFn)
            return ops.l1_multi([a for a, _ in pairs], [b for _, b in pairs], [D_weights * feat_weights] * len(pairs))
        terms = [ops.l1(a, b) for a, b in pairs]
        return weighted_sum(terms, [D_weights * feat_weights] * len(terms))

    def draw_use_gt(self, rng=None):
        """The use_gt coin of train.py:195.  It decides which parameters receive gradients (box_net only trains on
        use_gt steps) and hence which Adam slots advance: under data parallelism every rank must see the same value, so
        rank 0 draws and broadcasts."""
        import random as _random
        coin = (rng or _random).randint(0, 1)
        if self.distributed and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            coin = broadcast_int(coin, self.device)      # host-side (gloo control group): no GPU synchronisation
        return coin != 0

    def step(self, batch, use_gt=True):
        """One full G+D iteration = train.py:190-215.  ``batch`` = the 8-tuple of coco_collate_fn on the device."""
        imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img, attributes = batch
        if not use_gt:
            attributes = torch.zeros_like(attributes)
        # the dense layouts are outputs for logging only when the layout convs run on the factored form: defer their
        # kernels (Model.lazy_layouts) and run them at the end iff the caller wants the dense tensors back
        self.model.lazy_layouts = True
        try:
            model_out = self.model(imgs, objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=masks,
                                   attributes=attributes)
        finally:
            self.model.lazy_layouts = False
        imgs_pred, boxes_pred, masks_pred, layout, layout_pred, layout_wrong = model_out
        # Under data parallelism the four Adam steps move to the end of the iteration (_step_or_defer): the generator's
        # 765 MB gradient all-reduce and the discriminators' buckets then overlap the sub-steps that follow them instead of
        # stalling the stream.  (train.py's own loop calls the four functions itself: unchanged.)
        self._defer_g_step = bool(self.reducers) and getattr(self, 'overlap_g_reduce', True)
        self._deferred_steps = []
        self._adam_side = (streams.group_on('adam') and imgs.is_cuda and not self._defer_g_step and not self.reducers
                           and not torch.cuda.is_current_stream_capturing())
        # side streams whose work this step must join before it returns; sub-steps may only leave work there inside step()
        self._side_pending = []
        self._side_ok = imgs.is_cuda and not self.reducers and not torch.cuda.is_current_stream_capturing()
        try:
            self.train_generator(imgs, imgs_pred, masks, masks_pred, layout, objs, boxes, boxes_pred, obj_to_img, use_gt)
            self.train_mask_discriminator(masks, masks_pred.detach(), objs)
            self.train_obj_discriminator(imgs, imgs_pred.detach(), objs, boxes, boxes.detach(), obj_to_img)
            self.train_image_discriminator(imgs, imgs_pred.detach(), layout.detach(), layout_wrong.detach())
            for opt in self._deferred_steps:      # in the order the reduces were issued: G, mask D, object D, image D
                opt.step()
        finally:
            self._defer_g_step = False
            self._deferred_steps = []
            self._adam_side = self._side_ok = False
            for side in self._side_pending:          # the next reader of what they wrote is on the current stream
                torch.cuda.current_stream(imgs.device).wait_stream(side)
            self._side_pending = []
        if getattr(self, 'dense_layout_outputs', True):
            for lay in (layout, layout_pred, layout_wrong):
                ops.ensure_dense(lay)
        return model_out

    def write_losses(self, checkpoint, t):
        """trainer.py:342-369 (this is where the lazy LossManagers synchronise with the device)."""
        index = int(t / self.args.print_every)
        print('t = %d / %d' % (t, self.args.num_iterations))
        for tag, scope, key, L in (('G', 'g_loss', 'losses', getattr(self, 'generator_losses', None)),
                                   ('D_obj', 'd_obj_loss', 'd_losses', getattr(self, 'd_obj_losses', None)),
                                   ('D_mask', 'd_mask_loss', 'd_losses', getattr(self, 'd_mask_losses', None)),
                                   ('D_img', 'd_img_loss', 'd_losses', getattr(self, 'd_img_losses', None))):
            if L is None:
                continue
            for name, val in L.items():
                print(' %s [%s]: %.4f' % (tag, name, val))
                if checkpoint is not None:
                    hist = checkpoint.setdefault(key, {})
                    if name not in hist:                 # plain dict or the reference's defaultdict(list)
                        hist[name] = []
                    hist[name].append(val)
                self.writer.add_scalar('%s/%s' % (scope, name), val, index)
        if checkpoint is not None:
            checkpoint.setdefault('losses_ts', []).append(t)

    def write_images(self, t, imgs, imgs_pred, layout_one_hot, layout_pred_one_hot):
        """trainer.py:371-397: TensorBoard panels (host-side glue, outside the hot path)."""
        index = int(t / self.args.print_every)
        w = self.writer
        w.add_image('img/real', _make_grid(imgs), index)
        if imgs_pred is not None:
            w.add_image('img/pred', _make_grid(imgs_pred), index)
        if self.obj_discriminator is not None and getattr(self, 'd_real_crops', None) is not None:
            w.add_image('objs/d_real', _make_grid(self.d_real_crops), index)
            w.add_image('objs/g_fake', _make_grid(self.d_fake_crops), index)
        w.add_image('img/layout', _make_grid(self.one_hot_to_rgb(layout_one_hot)), index)
        w.add_image('img/layout_pred', _make_grid(self.one_hot_to_rgb(layout_pred_one_hot)), index)

    def one_hot_to_rgb(self, one_hot):
        """trainer.py:393-397"""
        base = one_hot
        while base._base is not None:            # a channel slice of a lazily built layout (train.py:201-202)
            base = base._base
        ops.ensure_dense(base)
        one_hot_3d = torch.einsum('abcd,be->aecd', [one_hot.detach().cpu(), self.colors])
        one_hot_3d *= (255.0 / one_hot_3d.max())
        return one_hot_3d
