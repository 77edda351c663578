"""Input-pipeline contract: collated host batch -> device batch.

The reference moves every tensor of a collated batch with a synchronous ``tensor.cuda()`` from pageable memory
(train.py:190-193) and later re-derives host-side facts from the device copies (``obj_to_img.max().item()`` and one
``.item()`` per object in layout.py:143-149, ``objs.tolist()`` in utils.py:67-90).  Here the host batch -- the 8-tuple of
``coco_collate_fn`` (data/coco.py:501-547: imgs, objs, boxes, masks, triples, obj_to_img, triple_to_img, attributes) -- is

  * validated against the collate contract on the host (node ids contiguous per image, images ascending: what
    ``_pool_samples`` silently relies on, layout.py:153-154),
  * summarised into the host lists the step needs (class ids for the VectorPool plan, obj_to_img for the factored layout
    planes, per-image segment offsets), so nothing is copied BACK from the device,
  * PACKED into one page-locked staging buffer, ``depth`` batches ahead of the step that consumes it (background thread: pure
    host work, no runtime call), and
  * brought to the device by ONE kernel on the consumer's own stream that reads the page-locked buffer through its device mapping
    (``sg_stage_copy``); the eight tensors of the device batch are views of one allocation.

What rounds 4-5 got wrong (pinned per-tensor buffers filled with ``Tensor.copy_``, hipMemcpyAsync on a copy stream, event
hand-over): fed from host batches the step ran 33-75 ms instead of 32 -- box-dependent, 498-967 images/s for the same code, 830.8 of
924.3 on the driver's box.  Found in round 6 (tools/probe/host_stall_probe.py, profiles/r06_host_stall.md): the GPU boxes run
this process in a cgroup with a CPU QUOTA (``cpu.max`` = 16 CPUs per 100 ms period, on a 256-CPU host), ``Tensor.copy_`` of the
6 MB image tensor fans out over torch's OpenMP pool (128 threads, which then spin), the burst spends the quota, and the kernel
freezes EVERY thread of the process -- the one issuing the launches included -- until the next period: a 20-60 ms stall every
~100 ms (``nr_throttled`` in cpu.stat counts them), during which the GPU runs dry.  Not the copies, not the second stream, not the
interpreter lock, not the garbage collector (each ruled out by its own A/B).  The staging below therefore uses a plain
single-threaded ``memmove`` per tensor; the copy-kernel form of the transfer was kept from the hunt because it is the simpler
hand-over (one queue: no second stream, no cross-stream event, no ``record_stream`` bookkeeping, no DMA engine) and costs
0.17 ms of a 32 ms step (8 MB over PCIe at 47 GB/s on the launch stream).  Measured after the fix: 32.24-32.30 ms/step from host
batches against 32.05 resident, threaded or inline staging alike.

Works for any iterable of collated batches: a torch DataLoader with the reference's ``coco_collate_fn`` or the synthetic
generator (scene_generation_amd.synthetic.make_batch).
"""
import ctypes
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
    """Iterates DeviceBatch objects; batch k+1 .. k+depth are validated, summarised and packed into page-locked memory while the
    caller trains on batch k; ``__next__`` issues the one copy kernel of the batch it returns on the current stream.

    ``threaded`` (default on a GPU): validation, the host summaries and the packing run on a background thread, ``depth``
    batches ahead: staging a 7.97 MB batch costs ~7 ms of host time (tools/probe/host_buffers_probe.py), the thread that issues
    the step's launches should not pay it."""

    ALIGN = 256

    def __init__(self, batches, device, validate=True, depth=2, threaded=None):
        self.src = iter(batches)
        self.device = torch.device(device)
        self.validate = validate
        self.cuda = self.device.type == 'cuda'
        self.depth = max(1, int(depth))
        self._queue = []
        # page-locked staging buffers, re-used (``t.pin_memory()`` would page-lock a fresh allocation per tensor and batch): one
        # buffer per slot holds the whole packed batch; 'ev' = the event behind the copy kernel that last read it.  depth + 2
        # slots rotate: when slot j % S is packed again, the consumer has long issued the copy of batch j - S (see _stage_one)
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
        """next host batch -> (DeviceBatch of HOST tensors, packing) with the batch packed into a page-locked slot; None at the
        end of the source.  ``packing`` = (slot, total bytes, [(offset, bytes, dtype, shape)] per tensor); None on a CPU device."""
        try:
            hb = next(self.src)
        except StopIteration:
            return None
        hb = Batch(*hb)
        o2i = validate_collated(hb) if self.validate else hb.obj_to_img.tolist()
        N = hb.imgs.size(0)
        objs_host = hb.objs.tolist()
        seg = segment_offsets(o2i, N)
        packing = None
        if self.cuda:
            slot = self._slots[self._staged % len(self._slots)]
            self._staged += 1
            if slot.get('ev') is not None:
                slot['ev'].synchronize()           # the copy kernel that last read this slot (depth + 2 batches ago)
                slot['ev'] = None
            fields, total = [], 0
            for t in hb:
                total = (total + self.ALIGN - 1) // self.ALIGN * self.ALIGN
                n = t.numel() * t.element_size()
                fields.append((total, n, t.dtype, tuple(t.shape)))
                total += n
            total = (total + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            buf = slot.get('buf')
            if buf is None or buf.numel() < total:
                buf = slot['buf'] = torch.empty(total * 5 // 4 + 4096, dtype=torch.uint8, pin_memory=True)
            base = buf.data_ptr()
            for t, (off, n, dtype, shape) in zip(hb, fields):
                if n:
                    # plain single-threaded memcpy with the interpreter lock released -- NOT ``Tensor.copy_``, whose OpenMP
                    # burst gets the whole process frozen by the cgroup's CPU quota (module docstring)
                    t = t if t.is_contiguous() else t.contiguous()
                    ctypes.memmove(base + off, t.data_ptr(), n)
            packing = (slot, total, fields)
        return DeviceBatch(hb, objs_host, o2i, seg, N), packing

    def _to_device(self, db, packing):
        """the consumer's side: one allocation, one copy kernel on the current stream, eight views"""
        if packing is None:
            return db
        from . import ops
        slot, total, fields = packing
        with torch.cuda.device(self.device):
            dev = torch.empty(total, dtype=torch.uint8, device=self.device)
            ops.stage_copy(dev, slot['buf'], total)
            ev = torch.cuda.Event()
            ev.record()
        slot['ev'] = ev
        moved = []
        for off, n, dtype, shape in fields:
            moved.append(dev[off:off + n].view(dtype).view(shape))
        return DeviceBatch(Batch(*moved), db.objs_host, db.obj_to_img_host, db.seg_offsets_host, db.num_images)

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
            db, packing = item
        else:
            while len(self._queue) < self.depth and self._stage():
                pass
            if not self._queue:
                raise StopIteration
            db, packing = self._queue.pop(0)
        return self._to_device(db, packing)

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
