"""Differentiable bilinear crops (surface of /root/reference/scene_generation/bilinear.py).

``crop_bbox_batch`` is one gather kernel indexed by ``bbox_to_feats`` (any order), replacing the per-image
nonzero() / expand / cat / grid_sample / inverse-permutation pipeline of bilinear.py:67-98.
"""
from . import ops


def crop_bbox_batch(feats, bbox, bbox_to_feats, HH, WW=None, backend='cudnn'):
    """feats (N, C, H, W), bbox (B, 4) in [0,1], bbox_to_feats (B,) int64 -> crops (B, C, HH, WW)."""
    if backend != 'cudnn':
        raise NotImplementedError("only the grid_sample ('cudnn') backend is on the training path (bilinear.py:40-41)")
    if WW is None:
        WW = HH
    return ops.CropBBoxFn.apply(feats, bbox, bbox_to_feats, int(HH), int(WW))


def crop_bbox(feats, bbox, HH, WW=None, backend='cudnn'):
    """feats[i] cropped by bbox[i] (bilinear.py:101-130)."""
    import torch
    N = feats.size(0)
    assert bbox.size(0) == N and bbox.size(1) == 4
    idx = torch.arange(N, dtype=torch.int64, device=feats.device)
    return crop_bbox_batch(feats, bbox, idx, HH, WW, backend)
