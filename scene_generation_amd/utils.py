"""Host-side helpers of the training step (mirror of /root/reference/scene_generation/utils.py).

* int_tuple / float_tuple / str_tuple / bool_flag : utils.py:22-40 (flag parsers).
* LossManager : utils.py:43-59 API, but values stay lazy device scalars -- the reference calls
  ``.item()`` inside add_loss (one device->host sync per loss, ~16 per step); here ``items()``
  materialises them only when somebody prints.
* VectorPool : utils.py:62-90 semantics (per-class replay pool, Python ``random.randint`` draws in the
  same order) with the pool RESIDENT IN HBM: one small D2H copy of the class ids per query instead of
  2*O syncs, index planning on the host, two HIP gather/scatter launches on the device.
"""
import random

import torch


def int_tuple(s):
    return tuple(int(i) for i in s.split(','))


def float_tuple(s):
    return tuple(float(i) for i in s.split(','))


def str_tuple(s):
    return tuple(s.split(','))


def bool_flag(s):
    if s == '1':
        return True
    if s == '0':
        return False
    raise ValueError('Invalid value "%s" for bool flag (should be 0 or 1)' % s)


def cpu_quota():
    """CPUs this process may use per scheduler period: the cgroup's ``cpu.max`` quota (v2; ``cpu.cfs_quota_us`` for v1) in units of
    CPUs, or None without a quota.  On the MI355X boxes of this project the container gets 16 CPUs of a 256-CPU host; a burst of
    more runnable threads than that (torch's OpenMP pool defaults to one thread per physical core) spends the quota within a few
    milliseconds and the kernel then freezes the WHOLE process until the next 100 ms period -- the launching thread included
    (scene_generation_amd/pipeline.py)."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


_QUOTA_APPLIED = [False]


def respect_cpu_quota(verbose=True):
    """Lower torch's intra-op thread count to the cgroup's CPU quota when it exceeds it (once per process; SG_KEEP_TORCH_THREADS=1
    opts out).  Called by Trainer.__init__: any host-side torch operator large enough to fan out over the default pool (one
    thread per physical core) while the step runs would otherwise get the whole process -- launch thread included -- frozen by
    the quota for the rest of a 100 ms period (``cpu_quota``).  Returns the thread count in effect."""
    import os
    if _QUOTA_APPLIED[0] or os.environ.get('SG_KEEP_TORCH_THREADS', '0') == '1':
        return torch.get_num_threads()
    _QUOTA_APPLIED[0] = True
    q = cpu_quota()
    n = torch.get_num_threads()
    if q is not None and n > max(1, int(q)):
        torch.set_num_threads(max(1, int(q)))
        if verbose:
            import sys
            print('scene_generation_amd: torch intra-op threads %d -> %d (cgroup CPU quota %.1f CPUs; SG_KEEP_TORCH_THREADS=1 keeps '
                  'the default)' % (n, torch.get_num_threads(), q), file=sys.stderr)
    return torch.get_num_threads()


def to_device_async(t, device):
    """Small host -> device copy that does not stall the host: a copy from PAGEABLE memory makes the host wait until the
    stream has drained (measured: ~3.5 ms each, 8 per step), a copy from pinned memory is just enqueued."""
    if isinstance(device, str):
        device = torch.device(device)
    if device.type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)



def weighted_sum(tensors, weights):
    """sum_i weights[i] * tensors[i] for scalar tensors.  On the device: ONE launch (ops.WeightedSumFn; the reference's
    chain of python-level ``loss = loss + w * term`` costs ~6 tiny kernels per term, forward + backward).  Host tensors
    (CPU unit tests of the host logic) take the plain torch expression."""
    if tensors[0].is_cuda:
        from . import ops
        return ops.weighted_sum(tensors, weights)
    w = torch.tensor([float(x) for x in weights], dtype=torch.float32)
    return torch.dot(torch.stack([t.reshape(()).float() for t in tensors]), w)


class LossManager(object):
    """utils.py:43-59 of the reference: ``total_loss = sum_i weight_i * loss_i`` and a name -> float dict.  Lazy: no
    ``.item()`` until somebody reads a value, and the weighted sum is built once when ``total_loss`` is first read."""

    def __init__(self):
        self._terms, self._weights = [], []
        self._total = None
        self._lazy = {}

    def add_loss(self, loss, name, weight=1.0, use_loss=True):
        if not isinstance(loss, torch.Tensor):
            loss = torch.as_tensor(float(loss))
        if use_loss:
            self._terms.append(loss)
            self._weights.append(float(weight))
            self._total = None
        self._lazy[name] = (loss.detach(), float(weight))

    @property
    def total_loss(self):
        if self._total is None and self._terms:
            self._total = weighted_sum(self._terms, self._weights)
        return self._total

    @property
    def all_losses(self):
        """name -> python float (materialised on access; this is where the host sync happens)"""
        out = {}
        for k, v in self._lazy.items():
            if isinstance(v, tuple):
                out[k] = float(v[0].item()) * v[1]
            else:
                out[k] = v.item() if isinstance(v, torch.Tensor) else float(v)
        return out

    def set_value(self, name, value):
        self._lazy[name] = value.detach() if isinstance(value, torch.Tensor) else value

    def items(self):
        return self.all_losses.items()


def plan_pool_query(classes, pool_len, pool_size, rng=random):
    """Host-side index planning for VectorPool.query (utils.py:67-90), processing objects in order.

    classes  : list[int] class id per object
    pool_len : dict class -> current fill of that class's pool (MUTATED to the post-query fill)
    Returns (src_kind, src_idx, slot) lists, one entry per object:
      out[i]   = vectors[src_idx[i]]            if src_kind[i] == 0   (a row of THIS batch, j <= i)
               = pool[class_i][src_idx[i]]      if src_kind[i] == 1   (content from BEFORE this query)
      slot[i]  = pool slot of class_i that vectors[i] is stored to afterwards, or -1 if a later object of the same
                 batch overwrites that slot (later writes win, exactly as the sequential reference loop).
    """
    src_kind, src_idx, slot = [], [], []
    shadow = {}                                   # (class, slot) -> batch row currently stored there
    for i, c in enumerate(classes):
        n = pool_len.get(c, 0)
        if n == 0:
            src_kind.append(0)
            src_idx.append(i)
            slot.append(0)
            shadow[(c, 0)] = i
            pool_len[c] = 1
        elif n < pool_size:
            r = rng.randint(0, n - 1)
            shadow[(c, n)] = i                    # append first (utils.py:79), then read slot r
            pool_len[c] = n + 1
            slot.append(n)
            j = shadow.get((c, r))
            src_kind.append(1 if j is None else 0)
            src_idx.append(r if j is None else j)
        else:
            r = rng.randint(0, n - 1)
            j = shadow.get((c, r))
            src_kind.append(1 if j is None else 0)
            src_idx.append(r if j is None else j)
            shadow[(c, r)] = i
            slot.append(r)
    # several objects of one batch may target the same (class, slot): only the LAST write survives
    slot = [sl if shadow[(c, sl)] == i else -1 for i, (c, sl) in enumerate(zip(classes, slot))]
    return src_kind, src_idx, slot


class VectorPool:
    def __init__(self, pool_size):
        self.pool_size = pool_size
        self.pool_len = {}                        # class -> fill (host)
        self.pool = None                          # (num_classes_seen_capacity, pool_size, R) device tensor
        self.capacity = 0

    def _ensure(self, max_class, R, like):
        if self.pool is None or max_class >= self.capacity or self.pool.size(2) != R:
            cap = max(max_class + 1, self.capacity, 16)
            new = torch.zeros(cap, self.pool_size, R, dtype=like.dtype, device=like.device)
            if self.pool is not None and self.pool.size(2) == R:
                new[:self.capacity] = self.pool
            self.pool, self.capacity = new, cap

    def query(self, objs, vectors, objs_host=None):
        if self.pool_size == 0:
            return vectors
        from . import ops
        classes = objs_host if objs_host is not None else objs.tolist()
        vectors = vectors.detach()
        self._ensure(max(classes), vectors.size(1), vectors)
        kind, idx, slot = plan_pool_query(classes, self.pool_len, self.pool_size)
        plan = to_device_async(torch.tensor([classes, kind, idx, slot], dtype=torch.int32), vectors.device)
        return ops.vector_pool_exchange(self.pool, vectors, plan)

    # host view used by tests / checkpoints
    def vectors_of(self, cls):
        n = self.pool_len.get(cls, 0)
        return [] if n == 0 else list(self.pool[cls, :n].cpu())


def active_layout_channels(objs, obj_to_img, num_images, num_objs, dense, extra=0):
    """Per-image list of layout channels that can be non-zero.

    The layout vectors are ``[one_hot(obj class) | representation]`` (model.py:165-168 of the reference), so image n
    only has the one-hot planes of the classes of its own objects plus the ``dense`` trailing channels non-zero.
    Host-side index plumbing on the (tiny) object lists; returns ``(chan_list [N, L] int32, chan_cnt [N] int32)`` as
    numpy arrays, lists ascending and padded with the first entry.  ``extra`` more channels right after the layout's
    (a channel-concatenated image, trainer.py:232-244) are always active.
    """
    import numpy as np
    per_img = [set() for _ in range(num_images)]
    for c, i in zip(objs, obj_to_img):
        if not (0 <= c < num_objs):
            raise ValueError('object class %d outside [0, %d)' % (c, num_objs))
        per_img[i].add(int(c))
    dense = dense + extra
    L = max(len(p) for p in per_img) + dense if per_img else dense
    chan_list = np.zeros((num_images, L), dtype=np.int32)
    chan_cnt = np.zeros((num_images,), dtype=np.int32)
    tail = list(range(num_objs, num_objs + dense))
    for n, p in enumerate(per_img):
        ch = sorted(p) + tail
        chan_cnt[n] = len(ch)
        chan_list[n, :len(ch)] = ch
        chan_list[n, len(ch):] = ch[0] if ch else 0
    return chan_list, chan_cnt
