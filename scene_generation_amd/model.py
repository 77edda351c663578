"""Scene graph -> image generator (surface of /root/reference/scene_generation/model.py:12-172, training branch).

Differences from the reference are confined to how the work is issued:
  * every operator is a HIP launch (scene_generation_amd.ops); no ``.item()``/``.nonzero()`` host syncs except
    the single small D2H copy of the class ids inside VectorPool.query (utils.py)
  * ``gt_layout``/``wrong_layout`` tell the layout kernel that the first ``num_objs`` channels are the constant
    one-hot block, so its backward only touches the ``rep_size`` appearance channels (model.py:165-168)
"""
import torch
import torch.nn as nn

from . import ops, streams
from .bilinear import crop_bbox_batch
from .generators import mask_net, AppearanceEncoder, define_G
from .graph import GraphTripleConv, GraphTripleConvNet
from .layers import build_mlp, Embedding, Linear
from .layout import masks_to_layout
from .utils import VectorPool, active_layout_channels, to_device_async


class Model(nn.Module):
    def __init__(self, vocab, image_size=(64, 64), embedding_dim=128, gconv_dim=128, gconv_hidden_dim=512,
                 gconv_pooling='avg', gconv_num_layers=5, mask_size=32, mlp_normalization='none',
                 appearance_normalization='', activation='', n_downsample_global=4, box_dim=128,
                 use_attributes=False, box_noise_dim=64, mask_noise_dim=64, pool_size=100, rep_size=32,
                 ngf=64, n_blocks_global=9):
        super().__init__()
        self.vocab = vocab
        self.image_size = image_size
        self.use_attributes = use_attributes
        self.box_noise_dim = box_noise_dim
        self.mask_noise_dim = mask_noise_dim
        self.object_size = 64
        self.fake_pool = VectorPool(pool_size)

        self.num_objs = len(vocab['object_to_idx'])
        self.num_preds = len(vocab['pred_idx_to_name'])
        self.obj_embeddings = Embedding(self.num_objs, embedding_dim)
        self.pred_embeddings = Embedding(self.num_preds, embedding_dim)

        attributes_dim = vocab['num_attributes'] if use_attributes else 0
        if gconv_num_layers == 0:
            self.gconv = Linear(embedding_dim, gconv_dim)
        elif gconv_num_layers > 0:
            self.gconv = GraphTripleConv(input_dim=embedding_dim, attributes_dim=attributes_dim, output_dim=gconv_dim,
                                         hidden_dim=gconv_hidden_dim, pooling=gconv_pooling,
                                         mlp_normalization=mlp_normalization)
        self.gconv_net = None
        if gconv_num_layers > 1:
            self.gconv_net = GraphTripleConvNet(input_dim=gconv_dim, hidden_dim=gconv_hidden_dim,
                                                pooling=gconv_pooling, num_layers=gconv_num_layers - 1,
                                                mlp_normalization=mlp_normalization)

        self.box_dim = box_dim
        self.box_net = build_mlp([self.box_dim, gconv_hidden_dim, 4], batch_norm=mlp_normalization)

        self.g_mask_dim = gconv_dim + mask_noise_dim
        self.mask_net = mask_net(self.g_mask_dim, mask_size)

        self.repr_input = self.g_mask_dim
        self.repr_net = build_mlp([self.repr_input, 64, rep_size], batch_norm=mlp_normalization)

        self.image_encoder = AppearanceEncoder(vocab=vocab, arch='C4-64-2,C4-128-2,C4-256-2',
                                               normalization=appearance_normalization, activation=activation,
                                               padding='valid', vecs_size=self.g_mask_dim)

        self.layout_to_image = define_G(self.num_objs + rep_size, 3, ngf, n_downsample_global, n_blocks_global,
                                        'instance')
        # hooks that are not part of the reference surface
        self.noise_override = None          # (1, mask_noise_dim) row used instead of torch.randn (parity tests)
        self.layout_objects_hint = 0        # objects/image the layout kernel provisions LDS for (0 = default 12)
        self.objs_host = None               # optional host copies of ``objs`` / ``obj_to_img`` (lists): skip the one
        self.obj_to_img_host = None         # D2H copy per forward that VectorPool and the sparse first conv need
        self.lazy_layouts = False           # defer the dense layout kernels until a dense read (see forward)
        self.rep_size = rep_size

    def forward(self, gt_imgs, objs, triples, obj_to_img, boxes_gt=None, masks_gt=None, attributes=None,
                test_mode=False, use_gt_box=False, features=None):
        ops.clear_hints()                        # layout hints of the previous iteration (and the tensors they pin)
        O = objs.size(0)
        objs_h, o2i_h = self.objs_host, self.obj_to_img_host
        if objs_h is None or o2i_h is None:      # one sync (the reference's pool does objs.tolist(), utils.py:104)
            objs_h, o2i_h = torch.stack((objs, obj_to_img)).tolist()
        N = gt_imgs.size(0) if gt_imgs is not None else max(o2i_h) + 1
        lazy = self.lazy_layouts and ops.FACTORED_LAYOUT
        # The OBJECT FRONT -- embeddings, graph convolutions, box_net, mask_net -- and the IMAGE PATH of the training branch --
        # crops, AppearanceEncoder, the layouts built from the ground-truth boxes and masks, the generator (model.py:98-124) --
        # share no tensor: boxes_pred / masks_pred feed losses and the (deferred) pred_layout only.  The front is ~100 launches of
        # a few microseconds forward and ~250 backward; on a side stream (streams.fork, group 'front') they run beside the
        # generator's GEMMs, forward here and backward wherever autograd runs the nodes recorded on that stream.  Only when the
        # dense pred_layout is deferred (Trainer.step) and no inference-time features are given; results are bit-identical.
        side = lazy and not test_mode and features is None
        with streams.fork(objs.device, 'front', enabled=side) as fk:
            with fk.branch(1, reads=(objs, triples, attributes)):
                obj_vecs, pred_vecs = self.scene_graph_to_vectors(objs, triples, attributes)
                box_vecs, mask_vecs = obj_vecs, self._mask_vecs(obj_vecs, O)
                boxes_pred = self.box_net(box_vecs)
                mask_scores = self.mask_net(mask_vecs.view(O, -1, 1, 1))
                masks_pred = ops.activation(mask_scores.squeeze(1), ops.ACT_SIGMOID)
                fk.produced((boxes_pred, masks_pred, mask_vecs))
            scene_layout_vecs, wrong_layout_vecs = self._appearance_vecs(gt_imgs, boxes_gt, obj_to_img, objs, mask_vecs, features,
                                                                         objs_host=objs_h)
            return self._layouts_and_image(gt_imgs, objs, obj_to_img, boxes_gt, masks_gt, boxes_pred, masks_pred,
                                           scene_layout_vecs, wrong_layout_vecs, objs_h, o2i_h, N, lazy, test_mode, use_gt_box)

    def _layouts_and_image(self, gt_imgs, objs, obj_to_img, boxes_gt, masks_gt, boxes_pred, masks_pred, scene_layout_vecs,
                           wrong_layout_vecs, objs_h, o2i_h, N, lazy, test_mode, use_gt_box):
        H, W = self.image_size
        kw = dict(num_images=N, validate=False, max_per_image=self.layout_objects_hint)
        if test_mode:                                      # model.py:111-117
            boxes = boxes_gt if use_gt_box else boxes_pred
            masks = masks_gt if masks_gt is not None else masks_pred
            pred_layout = masks_to_layout(scene_layout_vecs, boxes, masks, obj_to_img, H, W, test_mode=True,
                                          num_images=N, validate=False)
            ops.set_hints(pred_layout, sparse=tuple(
                to_device_async(torch.from_numpy(a), pred_layout.device)
                for a in active_layout_channels(objs_h, o2i_h, N, self.num_objs, self.rep_size)))
            return self.layout_to_image(pred_layout), boxes_pred, masks_pred, None, pred_layout, None
        seg = ops.segment_offsets(obj_to_img, N)
        if lazy:
            # nothing on the training step reads the dense layouts (the convs over them run on the factored form): their
            # kernels are deferred until a dense read (ops.ensure_dense; Trainer.step does it for the outputs it returns)
            def layout_of(vecs, masks):
                return ops.masks_to_layout_deferred(vecs, boxes_gt, masks, seg, N, H, W, False, self.layout_objects_hint,
                                                    differentiable=vecs.requires_grad)
        else:
            def layout_of(vecs, masks):
                return masks_to_layout(vecs, boxes_gt, masks, obj_to_img, H, W, test_mode=False,
                                       grad_from_channel=self.num_objs, **kw)
        gt_layout = layout_of(scene_layout_vecs, masks_gt)
        # pred_layout feeds no loss (train.py:203,219); it stays differentiable w.r.t. masks_pred like the reference's
        pred_layout = layout_of(scene_layout_vecs, masks_pred)
        wrong_layout = layout_of(wrong_layout_vecs, masks_gt)
        dev = gt_layout.device
        if not ops.FACTORED_LAYOUT:
            # per image only the one-hot planes of its own classes + the representation block are non-zero: the
            # generator's first conv (204 -> 64 channels, 7x7, full resolution) skips the rest
            sparse = tuple(to_device_async(torch.from_numpy(a), dev)
                           for a in active_layout_channels(objs_h, o2i_h, N, self.num_objs, self.rep_size))
            # same lists + the 3 image channels the image discriminator concatenates behind the layout
            sparse_img = tuple(to_device_async(torch.from_numpy(a), dev)
                               for a in active_layout_channels(objs_h, o2i_h, N, self.num_objs, self.rep_size, extra=3))
            for lay in (gt_layout, pred_layout, wrong_layout):
                ops.set_hints(lay, sparse=sparse, sparse_cat={3: sparse_img})
        if ops.FACTORED_LAYOUT:
            # factored form of the two layouts that feed convolutions: planes S_o + per-object vectors (ops.FactoredLayout)
            counts = [0] * N
            plane = []
            for i in o2i_h:
                plane.append(counts[i])
                counts[i] += 1
            pidx = to_device_async(torch.tensor(plane, dtype=torch.int64), dev)
            Z = ops.layout_planes(boxes_gt, masks_gt, seg, pidx, N, max(counts), H, W)
            f_gt = ops.FactoredLayout(Z, objs, scene_layout_vecs[:, self.num_objs:], self.num_objs, obj_to_img, pidx, counts, seg)
            f_wrong = ops.FactoredLayout(Z, objs, wrong_layout_vecs[:, self.num_objs:].detach(), self.num_objs, obj_to_img,
                                         pidx, counts, seg)
            f_wrong._lists = f_gt._lists          # same objects: share the list cache
            # 'wrong_twin': the image discriminator sees (layout, real image) and (wrong layout, real image) -- same planes, same
            # objects, other appearance vectors -- and can run the two passes as ONE batch (Trainer.train_generator)
            ops.set_hints(gt_layout, factored=f_gt, wrong_twin=f_wrong)
            ops.set_hints(wrong_layout, factored=f_wrong)
        imgs_pred = self.layout_to_image(gt_layout)
        return imgs_pred, boxes_pred, masks_pred, gt_layout, pred_layout, wrong_layout

    def scene_graph_to_vectors(self, objs, triples, attributes):
        s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
        edges = torch.stack([s, o], dim=1)                   # index plumbing (T x 2 int64)
        obj_vecs = self.obj_embeddings(objs)
        pred_vecs = self.pred_embeddings(p.contiguous())
        if self.use_attributes:
            obj_vecs = ops.concat_cols(obj_vecs, attributes)
        # the destination-major CSR of the triples (segmented pool, gather adjoint): built once for all graph-conv layers
        csr = ops.build_csr(edges, objs.numel()) if objs.is_cuda else None
        if isinstance(self.gconv, Linear):
            obj_vecs = self.gconv(obj_vecs)
        else:
            obj_vecs, pred_vecs = self.gconv(obj_vecs, pred_vecs, edges, csr=csr)
        if self.gconv_net is not None:
            obj_vecs, pred_vecs = self.gconv_net(obj_vecs, pred_vecs, edges, csr=csr)
        return obj_vecs, pred_vecs

    def create_components_vecs(self, imgs, boxes, obj_to_img, objs, obj_vecs, features, objs_host=None):
        """model.py:146-172 -- the reference's method, kept whole for callers of the reference surface; forward() runs its two
        halves separately (the first belongs to the object front, the second to the image path)."""
        mask_vecs = self._mask_vecs(obj_vecs, objs.size(0))
        layout_vecs, wrong_layout_vecs = self._appearance_vecs(imgs, boxes, obj_to_img, objs, mask_vecs, features,
                                                               objs_host=objs_host)
        return obj_vecs, mask_vecs, layout_vecs, wrong_layout_vecs

    def _mask_vecs(self, obj_vecs, O):
        if self.noise_override is not None:
            noise = self.noise_override.to(obj_vecs.device, obj_vecs.dtype).view(1, self.mask_noise_dim)
        else:                                                # ONE noise row per batch (model.py:149-151)
            noise = torch.randn((1, self.mask_noise_dim), dtype=obj_vecs.dtype, device=obj_vecs.device)
        return ops.concat_cols(obj_vecs, noise.expand(O, self.mask_noise_dim))

    def _appearance_vecs(self, imgs, boxes, obj_to_img, objs, mask_vecs, features, objs_host=None):
        if features is None:
            crops = crop_bbox_batch(imgs, boxes, obj_to_img, self.object_size)
            obj_repr = self.repr_net(self.image_encoder(crops))
        else:                                                # only at inference time (model.py:158-163)
            obj_repr = self.repr_net(mask_vecs)
            rows = [i for i, f in enumerate(features) if f is not None]
            if rows:
                obj_repr = obj_repr.clone()
                obj_repr[torch.tensor(rows, device=obj_repr.device)] = torch.stack(
                    [torch.as_tensor(features[i], dtype=obj_repr.dtype).to(obj_repr.device).view(-1) for i in rows])

        one_hot_obj = ops.one_hot(objs, self.num_objs)
        layout_vecs = ops.concat_cols(one_hot_obj, obj_repr)

        wrong_objs_rep = self.fake_pool.query(objs, obj_repr, objs_host=objs_host if objs_host is not None else self.objs_host)
        wrong_layout_vecs = ops.concat_cols(one_hot_obj, wrong_objs_rep)
        return layout_vecs, wrong_layout_vecs
