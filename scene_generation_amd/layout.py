"""Object vectors + boxes + masks -> dense layout (surface of /root/reference/scene_generation/layout.py).

One fused HIP kernel per call (sg_masks_to_layout_fwd): never materialises the (O, D, H, W) canvas, no
Python loop, no per-object ``.item()``.  Semantics = the reference as executed by torch >= 1.3:
grid_sample(bilinear, zeros padding, align_corners=False).
"""
import torch

from . import ops


def _num_images(obj_to_img, num_images, validate):
    if num_images is not None and not validate:
        return int(num_images)
    # API-compatible path: one device->host sync, like obj_to_img.max().item() at layout.py:143
    o2i = obj_to_img.detach().cpu()
    N = int(o2i.max()) + 1 if o2i.numel() else 0
    if num_images is not None:
        N = max(N, int(num_images))
    counts = torch.bincount(o2i, minlength=N)
    if (counts == 0).any() or bool((o2i[1:] < o2i[:-1]).any()):
        # the reference raises ValueError from list.index at layout.py:153-154
        raise ValueError('obj_to_img must be sorted and every image in [0, N) must own at least one object')
    return N


def masks_to_layout(vecs, boxes, masks, obj_to_img, H, W=None, pooling='sum', test_mode=False,
                    num_images=None, validate=True, grad_from_channel=0, max_per_image=0):
    """
    - vecs (O, D), boxes (O, 4) [x0, y0, x1, y1] in [0, 1], masks (O, M, M) int64 or float32,
      obj_to_img (O,) int64 sorted.  Returns (N, D, H, W).   (layout.py:64-93)
    Extra keyword arguments (not in the reference) remove host syncs: ``num_images`` + ``validate=False`` skip
    the max()/validation sync; ``grad_from_channel`` tells backward that vecs[:, :c] is constant (the one-hot
    block, model.py:165-168); ``max_per_image`` sizes the LDS tile.
    """
    if pooling not in ('sum', 'avg'):
        raise ValueError('Invalid pooling "%s"' % pooling)
    O, D = vecs.size()
    M = masks.size(1)
    assert masks.size() == (O, M, M)
    if W is None:
        W = H
    N = _num_images(obj_to_img, num_images, validate)
    seg = ops.segment_offsets(obj_to_img, N)
    if test_mode:      # layout.py:87-92,157-169: front-to-back compositing in ascending-mass order, on the device
        return ops.masks_to_layout_test(vecs, boxes, masks, seg, N, H, W, pooling == 'avg')
    out = ops.MasksToLayoutFn.apply(vecs, boxes, masks, seg, N, H, W, pooling == 'avg', int(grad_from_channel),
                                    int(max_per_image), obj_to_img)
    if grad_from_channel > 0:
        # backward only reads d out[:, grad_from_channel:]; consumers (the generator's first conv) may skip the rest
        ops.set_hints(out, grad_from=int(grad_from_channel))
    return out


def boxes_to_layout(vecs, boxes, obj_to_img, H, W=None, pooling='sum', **kw):
    """Intended semantics of layout.py:28-61 (the reference raises TypeError at :59, so parity is unpinned):
    masks_to_layout with an all-ones 8x8 mask (layout.py:50)."""
    ones = torch.ones(vecs.size(0), 8, 8, dtype=torch.float32, device=vecs.device)
    return masks_to_layout(vecs, boxes, ones, obj_to_img, H, W, pooling=pooling, **kw)
