"""Evaluation hooks next to the training path: the validation loop of /root/reference/train.py:80-116
(``check_model``: test-mode forward + box IoU), the IoU itself (scene_generation/metrics.py:19-35) and the appearance feature
bank of /root/reference/scripts/encode_features.py:103-146 (``repr_net(image_encoder(crops))`` grouped by class, then
k-means centres ordered along a 1-D t-SNE).  The network work runs on the HIP modules (test-mode compositing, crop, encoder,
MLP); the box arithmetic is a few element-wise lines on O x 4 numbers.  The Inception score stays out of scope (needs the
pretrained Inception network): ``check_model`` drives any object with the reference's ``clean() / __call__ / compute_score``
interface, or none."""
import numpy as np
import torch

from .bilinear import crop_bbox_batch


def intersection(bbox_pred, bbox_gt):
    """area of the overlap of boxes given as (x0, y0, x1, y1) rows"""
    wh = (torch.minimum(bbox_pred[:, 2:], bbox_gt[:, 2:]) - torch.maximum(bbox_pred[:, :2], bbox_gt[:, :2])).clamp(min=0)
    return wh[:, 0] * wh[:, 1]


def _area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def jaccard(bbox_pred, bbox_gt):
    """(sum of IoUs, #IoU > 0.5, #IoU > 0.3) -- metrics.py:27-35; the sum stays a device tensor"""
    inter = intersection(bbox_pred, bbox_gt)
    iou = inter / (_area(bbox_pred) + _area(bbox_gt) - inter)
    return iou.sum(), int((iou > 0.5).sum()), int((iou > 0.3).sum())


def check_model(args, loader, model, inception_score=None, use_gt=True, device=None):
    """train.py:80-116: run the model the way it ran during training but in test mode (``use_gt``: ground-truth boxes and
    masks, attributes kept; else predicted boxes and masks, attributes zeroed), accumulate the box IoU against the
    ground truth over ``args.num_val_samples`` images, feed the images to ``inception_score``.
    Returns (avg_iou, inception_mean, inception_std, fid) with fid None like the reference."""
    device = device or next(model.parameters()).device
    total_iou, total_boxes, seen = None, 0, 0
    if inception_score is not None:
        inception_score.clean()
    with torch.no_grad():
        for batch in loader:
            imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img, attributes = [t.to(device) for t in batch]
            if use_gt:
                out = model(imgs, objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=masks, attributes=attributes,
                            test_mode=True, use_gt_box=True)
            else:
                out = model(imgs, objs, triples, obj_to_img, boxes_gt=boxes, masks_gt=None,
                            attributes=torch.zeros_like(attributes), test_mode=True, use_gt_box=False)
            imgs_pred, boxes_pred = out[0], out[1]
            iou = jaccard(boxes_pred, boxes)[0]
            total_iou = iou if total_iou is None else total_iou + iou
            total_boxes += boxes_pred.size(0)
            if inception_score is not None:
                inception_score(imgs_pred)
            seen += imgs.size(0)
            if seen >= args.num_val_samples:
                break
    mean = std = None
    if inception_score is not None:
        mean, std = inception_score.compute_score(splits=5)
    avg_iou = float(total_iou) / total_boxes if total_boxes else float('nan')      # the loop's only host synchronisation
    return avg_iou, mean, std, None


def encode_features(model, loader, object_size=64, device=None, max_objects=None):
    """encode_features.py:112-134: ``repr_net(image_encoder(crop))`` of every object, grouped by class ->
    {class id: float array [count, rep_size]} (the file the sampling GUI's feature bank is built from).  The model is used in
    whatever mode it is in (the script's default is eval: BatchNorm running statistics)."""
    device = device or next(model.parameters()).device
    num_objs = len(model.vocab['object_to_idx'])
    chunks = {label: [] for label in range(num_objs)}
    rep, count = None, 0
    with torch.no_grad():
        for data in loader:
            imgs, objs, boxes, obj_to_img = data[0].to(device), data[1], data[2].to(device), data[5].to(device)
            feat = model.repr_net(model.image_encoder(crop_bbox_batch(imgs, boxes, obj_to_img, object_size))).float().cpu().numpy()
            rep = feat.shape[1]
            for row, label in zip(feat, objs.tolist()):
                chunks[label].append(row)
            count += len(feat)
            if max_objects is not None and count >= max_objects:
                break
    rep = rep if rep is not None else getattr(model, 'rep_size', 0)
    # float64 banks like the reference's features.npy: encode_features.py:121,133 appends float32 rows to np.zeros((0, rep)),
    # and numpy promotes to float64 (the KMeans centres downstream inherit the dtype)
    return {label: (np.stack(rows).astype(np.float64) if rows else np.zeros((0, rep))) for label, rows in chunks.items()}


def cluster_features(features, n_clusters, random_state=0):
    """encode_features.py:83-100: per class, k-means centres (k = min(count, n_clusters)) sorted along a 1-D t-SNE embedding"""
    from sklearn.cluster import KMeans
    from sklearn.manifold import TSNE
    centers = {}
    for label, feat in features.items():
        if not feat.shape[0]:
            continue
        k = min(feat.shape[0], n_clusters)
        km = KMeans(n_clusters=k, random_state=random_state).fit(feat)
        if k == 1:
            centers[label] = km.cluster_centers_
        else:
            order = np.argsort(TSNE(n_components=1, perplexity=min(30.0, k - 1.0) if k > 1 else 1.0)
                               .fit_transform(km.cluster_centers_).reshape(-1))
            centers[label] = km.cluster_centers_[order]
    return centers
