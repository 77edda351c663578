"""Differentiable bilinear crops (surface of /root/reference/scene_generation/bilinear.py).

``crop_bbox_batch`` is one gather kernel indexed by ``bbox_to_feats`` (any order), replacing the per-image
nonzero() / expand / cat / grid_sample / inverse-permutation pipeline of bilinear.py:67-98.
"""
from . import ops


def crop_bbox_batch(feats, bbox, bbox_to_feats, HH, WW=None, backend='cudnn'):
    """feats (N, C, H, W), bbox (B, 4) in [0,1], bbox_to_feats (B,) int64 -> crops (B, C, HH, WW).

    ``backend``: every value gives the grid_sample crop.  The reference's non-'cudnn' branch (bilinear.py:42-56) loops over the
    images and calls ``crop_bbox(cur_feats, cur_bbox, HH, WW)`` WITHOUT forwarding ``backend``, i.e. with crop_bbox's default
    'cudnn': its result is bit-identical to the 'cudnn' branch (checked against the reference in tools/make_golden.py,
    golden ``crop_jj_batch``).  The ``bilinear_sample`` geometry is only reachable through ``crop_bbox(backend='jj')``."""
    if WW is None:
        WW = HH
    return ops.CropBBoxFn.apply(feats, bbox, bbox_to_feats, int(HH), int(WW))


def crop_bbox(feats, bbox, HH, WW=None, backend='cudnn'):
    """feats[i] cropped by bbox[i] (bilinear.py:101-130)."""
    import torch
    N = feats.size(0)
    assert bbox.size(0) == N and bbox.size(1) == 4
    if WW is None:
        WW = HH
    if backend == 'jj':
        # bilinear.py:127-128 (``bilinear_sample``: pixel coordinates X * W without the half-pixel shift, clamped taps); called by
        # nothing in the reference (crop_bbox_batch never forwards its backend, see above), built for the surface's completeness
        return ops.CropBBoxJJFn.apply(feats, bbox, int(HH), int(WW))
    if backend != 'cudnn':
        return None                         # bilinear.py:125-130 falls off the end for any other value
    N = feats.size(0)
    assert bbox.size(0) == N and bbox.size(1) == 4
    idx = torch.arange(N, dtype=torch.int64, device=feats.device)
    return crop_bbox_batch(feats, bbox, idx, HH, WW, backend)
