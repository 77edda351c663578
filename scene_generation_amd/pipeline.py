"""Input-pipeline contract: collated host batch -> device batch.

The reference moves every tensor of a collated batch with a synchronous ``tensor.cuda()`` from pageable memory
(train.py:190-193) and later re-derives host-side facts from the device copies (``obj_to_img.max().item()`` and one
``.item()`` per object in layout.py:143-149, ``objs.tolist()`` in utils.py:67-90).  Here the host batch -- the 8-tuple of
``coco_collate_fn`` (data/coco.py:501-547: imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img, attributes) -- is

  * validated against the collate contract on the host (node ids contiguous per image, images ascending: what
    ``_pool_samples`` silently relies on, layout.py:153-154),
  * summarised into the host lists the step needs (class ids for the VectorPool plan, obj_to_img for the factored layout
    planes, per-image segment offsets), so nothing is copied BACK from the device,
  * copied through PINNED staging buffers on a side stream, one batch ahead of the step that consumes it.

Works for any iterable of collated batches: a torch DataLoader with the reference's ``coco_collate_fn`` or the synthetic
generator (scene_generation_amd.synthetic.make_batch).
"""
from collections import namedtuple

import torch

from .synthetic import Batch

DeviceBatch = namedtuple('DeviceBatch', 'batch objs_host obj_to_img_host seg_offsets_host num_images')


def validate_collated(batch):
    """the collate contract (data/coco.py:517-534) the layout / pooling kernels rely on; raises ValueError like the
    reference's ``list.index`` does at layout.py:153-154"""
    imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img, attributes = batch
    N, O, T = imgs.size(0), objs.size(0), triples.size(0)
    if not (boxes.shape == (O, 4) and masks.dim() == 3 and masks.size(0) == O and obj_to_img.shape == (O,)
            and triples.shape == (T, 3) and triple_to_img.shape == (T,) and attributes.size(0) == O):
        raise ValueError('collated batch: inconsistent shapes')
    o2i = obj_to_img.tolist()
    if any(b < a for a, b in zip(o2i, o2i[1:])) or (o2i and (o2i[0] != 0 or o2i[-1] != N - 1)) or len(set(o2i)) != N:
        raise ValueError('obj_to_img must be sorted and every image in [0, N) must own at least one object')
    if T and (int(triples[:, [0, 2]].max()) >= O or int(triples.min()) < 0):
        raise ValueError('triples reference objects outside the batch')
    return o2i


def segment_offsets(o2i, N):
    off = [0] * (N + 1)
    for i in o2i:
        off[i + 1] += 1
    for n in range(N):
        off[n + 1] += off[n]
    return off


def threading_current():
    import threading
    return threading.current_thread()


class DeviceBatchPrefetcher(object):
    """Iterates DeviceBatch objects; the H2D copies of batch k+1 are in flight (pinned memory, side stream) while the
    caller trains on batch k.

    ``threaded`` (default on a GPU): validation, the host summaries and the copy into the pinned slots run on a background
    thread, ``depth`` batches ahead.  Measured on MI355X (tools/probe/host_buffers_probe.py): staging a 7.97 MB batch costs ~7 ms
    of host time, and the thread that issues the step's ~1 100 launches has only ~3 ms of slack per 32.5 ms step -- staged
    inline, the step went from 32.5 to 42 ms."""

    def __init__(self, batches, device, validate=True, depth=2, threaded=None):
        self.src = iter(batches)
        self.device = torch.device(device)
        self.validate = validate
        self.cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.depth = max(1, int(depth))
        self._queue = []
        # pinned staging buffers, re-used: ``t.pin_memory()`` page-locks a fresh allocation per tensor and batch; a slot holds one
        # buffer per tensor of the 8-tuple and the event behind its last H2D copies, depth + 2 slots rotate
        self._slots = [dict() for _ in range(self.depth + 2)]
        self._staged = 0
        self.threaded = self.cuda if threaded is None else bool(threaded)
        self._q = self._thread = None
        if self.threaded:
            import queue
            import threading
            self._q = queue.Queue(maxsize=self.depth)
            self._stop = threading.Event()
            import weakref
            self._thread = threading.Thread(target=DeviceBatchPrefetcher._worker_main, name='sg-prefetch', daemon=True,
                                            args=(weakref.ref(self), self._q, self._stop))
            self._thread.start()

    def _stage_one(self):
        """next host batch -> (DeviceBatch, event) with its copies queued on the copy stream; None at the end of the source"""
        try:
            hb = next(self.src)
        except StopIteration:
            return None
        hb = Batch(*hb)
        o2i = validate_collated(hb) if self.validate else hb.obj_to_img.tolist()
        N = hb.imgs.size(0)
        objs_host = hb.objs.tolist()
        seg = segment_offsets(o2i, N)
        if self.cuda:
            slot = self._slots[self._staged % len(self._slots)]
            self._staged += 1
            if slot.get('ev') is not None:
                slot['ev'].synchronize()           # the copies that last read this slot's buffers (depth + 2 batches ago)
            moved = []
            with torch.cuda.stream(self.stream):
                for i, t in enumerate(hb):
                    t = t.contiguous()
                    buf = slot.get(i)
                    if buf is None or buf.dtype != t.dtype or buf.numel() < t.numel():
                        buf = slot[i] = torch.empty(t.numel() * 5 // 4 + 16, dtype=t.dtype, pin_memory=True)
                    view = buf[:t.numel()].view(t.shape)
                    view.copy_(t)                  # host memcpy into page-locked memory
                    moved.append(view.to(self.device, non_blocking=True))
            dev = Batch(*moved)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            slot['ev'] = ev
        else:
            dev, ev = hb, None
        return DeviceBatch(dev, objs_host, o2i, seg, N), ev

    @staticmethod
    def _worker_main(wref, q, stop):
        """Staging thread.  It holds the prefetcher only through a weak reference and only while it stages a batch: a consumer that
        drops the iterator without close() lets it be collected (``__del__`` -> close()), and the worker -- which would otherwise
        sit in a blocking put on a full queue forever, keeping the pinned slots and the source iterator alive -- sees the stop
        flag or the dead reference within 0.1 s.  Exceptions travel to the consumer through the same bounded put."""
        import queue
        me = wref()
        if me is None:
            return
        if me.cuda and me.device.index is not None:
            torch.cuda.set_device(me.device)
        del me
        while not stop.is_set():
            me = wref()
            if me is None:
                return
            try:
                item = me._stage_one()
            except BaseException as e:            # re-raised in the consumer (validation errors must not vanish in a thread)
                item = e
            del me
            while True:
                if stop.is_set() or wref() is None:
                    return
                try:
                    q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    continue
            if item is None or isinstance(item, BaseException):
                return

    def _stage(self):
        item = self._stage_one()
        if item is None:
            return False
        self._queue.append(item)
        return True

    def __iter__(self):
        return self

    def __next__(self):
        if self.threaded:
            if self._thread is None:
                raise StopIteration
            item = self._q.get()
            if item is None or isinstance(item, BaseException):
                self._thread = None
                if item is None:
                    raise StopIteration
                raise item
            db, ev = item
        else:
            while len(self._queue) < self.depth and self._stage():
                pass
            if not self._queue:
                raise StopIteration
            db, ev = self._queue.pop(0)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)     # stream-side wait: the host does not block
            for t in db.batch:
                t.record_stream(torch.cuda.current_stream(self.device))
        return db

    def close(self):
        """stop the staging thread and release what it holds (pinned slots, the source iterator).  Call it -- or use the
        prefetcher as a context manager -- when a loop leaves early (``break`` at max iterations): the worker reads up to
        depth + 1 batches ahead of the consumer, and those batches are consumed from the source."""
        th = self._thread
        if self.threaded and th is not None:
            self._stop.set()
            try:
                while True:
                    self._q.get_nowait()
            except Exception:
                pass
            self._thread = None
            if th is not threading_current():
                th.join(timeout=2.0)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
