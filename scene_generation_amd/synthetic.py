"""Synthetic COCO-shaped scene-graph batches + closed-form deterministic weight fill.

Pure torch-CPU host code shared by tools/make_golden.py (this container), the parity tests and
bench.py (GPU box).  Nothing here touches the reference or the oracle.

The batch layout is the collate contract of the reference's input pipeline
(/root/reference/scene_generation/data/coco.py:501-547): node ids are global, contiguous per
image, images ascending; every image ends with an ``__image__`` node (class 0, box [0,0,1,1],
all-ones mask; coco.py:313-317); triples are (s, p, o) int64 with one ``__in_image__`` (p=0)
triple per real object (coco.py:409-413).
"""
import math
import zlib
from collections import namedtuple

import torch

VOCAB_C = 172   # object classes incl. __image__ (scripts/sample_images.py:23 hard-codes 172)
VOCAB_P = 7     # predicates (coco.py:18,206)
VOCAB_A = 35    # 10 size + 25 location attribute dims (coco.py:25-26)

Batch = namedtuple('Batch', 'imgs objs boxes masks triples obj_to_img triple_to_img attributes')

# name -> (N images, min objs, max objs, spatial triples per object, H)
CONFIGS = {
    'c1': dict(N=4, min_objs=4, max_objs=4, spatial_per_obj=1, size=64),
    'c2': dict(N=32, min_objs=3, max_objs=8, spatial_per_obj=1, size=128),
    'c4': dict(N=8, min_objs=3, max_objs=16, spatial_per_obj=1, size=256),
    'c5': dict(N=32, min_objs=32, max_objs=32, spatial_per_obj=2, size=128),
}


def make_vocab(num_objs=VOCAB_C, num_preds=VOCAB_P, num_attributes=VOCAB_A):
    """The three vocab keys the training path reads (model.py:30-31,36)."""
    names = ['__in_image__', 'left of', 'right of', 'above', 'below', 'inside', 'surrounding']
    return {
        'object_to_idx': {i: i for i in range(num_objs)},
        'pred_idx_to_name': names[:num_preds] if num_preds <= len(names) else
        names + ['p%d' % i for i in range(len(names), num_preds)],
        'num_attributes': num_attributes,
    }


def make_batch(N=32, min_objs=3, max_objs=8, spatial_per_obj=1, size=128, mask_size=32,
               num_objs=VOCAB_C, num_preds=VOCAB_P, num_attributes=VOCAB_A, seed=0,
               zero_attributes=False):
    """Build one collated batch on the CPU."""
    g = torch.Generator().manual_seed(seed)
    H = W = size
    imgs = torch.rand(N, 3, H, W, generator=g) * 2 - 1
    objs, boxes, masks, triples, obj_to_img, triple_to_img, attrs = [], [], [], [], [], [], []
    base = 0
    for n in range(N):
        k = int(torch.randint(min_objs, max_objs + 1, (1,), generator=g))
        cls = torch.randint(1, num_objs, (k,), generator=g)
        x0 = torch.rand(k, generator=g) * 0.5
        y0 = torch.rand(k, generator=g) * 0.5
        w = 0.15 + torch.rand(k, generator=g) * 0.35
        h = 0.15 + torch.rand(k, generator=g) * 0.35
        bx = torch.stack([x0, y0, (x0 + w).clamp(max=1.0), (y0 + h).clamp(max=1.0)], 1)
        mk = (torch.rand(k, mask_size, mask_size, generator=g) < 0.7).long()
        at = torch.zeros(k + 1, num_attributes)
        if num_attributes >= 35:
            sz = torch.randint(0, 10, (k,), generator=g)
            loc = torch.randint(0, 25, (k,), generator=g)
            at[torch.arange(k), sz] = 1
            at[torch.arange(k), 10 + loc] = 1
            at[k, 9] = 1
            at[k, 10 + 12] = 1
        elif num_attributes > 0:
            a = torch.randint(0, num_attributes, (k + 1,), generator=g)
            at[torch.arange(k + 1), a] = 1
        # spatial triples between real objects (coco.py:358-406)
        for i in range(k):
            for _ in range(spatial_per_obj):
                if k > 1:
                    other = int(torch.randint(0, k - 1, (1,), generator=g))
                    other = other + 1 if other >= i else other
                else:
                    other = i
                p = int(torch.randint(1, max(num_preds, 2), (1,), generator=g))
                if int(torch.randint(0, 2, (1,), generator=g)):
                    s, o = i, other
                else:
                    s, o = other, i
                triples.append([base + s, p, base + o])
                triple_to_img.append(n)
        for i in range(k):
            triples.append([base + i, 0, base + k])
            triple_to_img.append(n)
        objs.append(torch.cat([cls, torch.zeros(1, dtype=torch.long)]))
        boxes.append(torch.cat([bx, torch.tensor([[0., 0., 1., 1.]])]))
        masks.append(torch.cat([mk, torch.ones(1, mask_size, mask_size, dtype=torch.long)]))
        attrs.append(at)
        obj_to_img.extend([n] * (k + 1))
        base += k + 1
    attributes = torch.cat(attrs)
    if zero_attributes:
        attributes = torch.zeros_like(attributes)
    return Batch(imgs, torch.cat(objs), torch.cat(boxes), torch.cat(masks),
                 torch.tensor(triples, dtype=torch.long), torch.tensor(obj_to_img, dtype=torch.long),
                 torch.tensor(triple_to_img, dtype=torch.long), attributes)


def make_config_batch(name, seed=0, **over):
    cfg = dict(CONFIGS[name])
    cfg.update(over)
    return make_batch(seed=seed, **cfg)


def shard_batch(batch, rank, world):
    """Data-parallel partition: rank r owns images [r*N/p, (r+1)*N/p) and their
    nodes/triples, re-based to local ids.  Pure index arithmetic on the host."""
    N = batch.imgs.size(0)
    assert N % world == 0, 'global batch must divide by world size'
    per = N // world
    lo, hi = rank * per, (rank + 1) * per
    osel = (batch.obj_to_img >= lo) & (batch.obj_to_img < hi)
    tsel = (batch.triple_to_img >= lo) & (batch.triple_to_img < hi)
    oidx = osel.nonzero().view(-1)
    obase = int(oidx[0]) if oidx.numel() else 0
    tri = batch.triples[tsel].clone()
    tri[:, 0] -= obase
    tri[:, 2] -= obase
    return Batch(batch.imgs[lo:hi].contiguous(), batch.objs[osel], batch.boxes[osel], batch.masks[osel], tri,
                 batch.obj_to_img[osel] - lo, batch.triple_to_img[tsel] - lo, batch.attributes[osel])


def batch_to(batch, device):
    return Batch(*[t.to(device) for t in batch])


# ----------------------------------------------------------------------------------------------
# Closed-form deterministic parameter fill (no RNG): identical on every host, so goldens captured
# from the reference in the build container can be reproduced bit-for-bit on the GPU box without
# shipping weights.
# ----------------------------------------------------------------------------------------------

# optional memo {(numel, salt): pattern} of the large fill patterns: a test session that fills many full-width models sets this
# to a dict (tests/conftest.py) and pays the int64 hash of a 183 M-parameter generator once instead of once per Trainer
HASH_CACHE = None


def _hash_uniform(numel, salt):
    """u in [-0.5, 0.5): 32-bit multiplicative hash of the element index, int64-exact."""
    if HASH_CACHE is not None and numel >= 65536:
        hit = HASH_CACHE.get((numel, salt))
        if hit is None:
            hit = HASH_CACHE[(numel, salt)] = _hash_uniform_compute(numel, salt)
        return hit
    return _hash_uniform_compute(numel, salt)


def _hash_uniform_compute(numel, salt):
    idx = torch.arange(numel, dtype=torch.int64)
    x = (idx * 2654435761 + (salt + 1) * 40503 * 65537) & 0xFFFFFFFF
    x = (x ^ (x >> 15)) * 2246822519 & 0xFFFFFFFF
    x = (x ^ (x >> 13)) & 0xFFFFFFFF
    return (x.double() / 4294967296.0 - 0.5).float()


def fill_deterministic(module, gain=1.0):
    """Fill every parameter/buffer of ``module`` with a closed-form pattern keyed by its state_dict NAME
    (crc32), so two implementations with the same key names get identical values regardless of order.

    weights (dim>=2): U(-a, a), a = gain*sqrt(3/fan_in);  1-D ``weight`` (norm scale): 1 + 0.2u;
    ``bias``: 0.2u; running_mean 0.1u; running_var 1 + 0.2u(+0.1); num_batches_tracked 0.
    """
    sd = module.state_dict()
    with torch.no_grad():
        for i, (name, t) in enumerate(sd.items()):
            leaf = name.rsplit('.', 1)[-1]
            if leaf == 'num_batches_tracked':
                t.zero_()
                continue
            u = _hash_uniform(t.numel(), zlib.crc32(name.encode()) & 0xFFFF).view(t.shape)
            if leaf == 'running_mean':
                v = 0.2 * u
            elif leaf == 'running_var':
                v = 1.1 + 0.4 * u
            elif t.dim() >= 2:
                fan_in = t[0].numel() if leaf == 'weight' else t.size(-1)
                if 'embedding' in name:
                    v = 2.0 * u
                else:
                    v = u * 2 * gain * math.sqrt(3.0 / max(fan_in, 1))
            elif leaf == 'weight':
                v = 1.0 + 0.4 * u
            else:
                v = 0.4 * u
            t.copy_(v.to(t.dtype))
    return module
