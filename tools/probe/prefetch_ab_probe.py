"""A/B of the staging thread of DeviceBatchPrefetcher inside the real step (round 6): bounded put vs blocking put vs inline staging."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scene_generation_amd.args import parser
from scene_generation_amd import pipeline
from scene_generation_amd.pipeline import DeviceBatchPrefetcher
from scene_generation_amd.synthetic import make_batch, make_vocab
from scene_generation_amd.trainer import Trainer

dev = 'cuda:0'
torch.cuda.set_device(0)
args = parser.parse_args(['--image_size', '128,128', '--batch_size', '32', '--vgg_features_weight', '0', '--output_dir', '/tmp/o'])
torch.manual_seed(1234)
tr = Trainer(args, make_vocab(), device=dev)
tr.model.layout_objects_hint = 9
tr.dense_layout_outputs = False
hb = [make_batch(N=32, min_objs=3, max_objs=8, size=128, seed=i) for i in range(2)]
staged = list(DeviceBatchPrefetcher(hb, dev))
random.seed(0)


def step(db):
    tr.model.objs_host, tr.model.obj_to_img_host = db.objs_host, db.obj_to_img_host
    tr.step(db.batch, use_gt=tr.draw_use_gt())


for i in range(6):
    step(staged[i % 2])
torch.cuda.synchronize()
K = 20


def run(label, make):
    it = iter(make())
    for _ in range(4):
        step(next(it))
    torch.cuda.synchronize()
    tn = ts = 0.0
    t0 = time.perf_counter()
    for i in range(K):
        a = time.perf_counter(); db = next(it); b = time.perf_counter(); step(db); c = time.perf_counter()
        tn += b - a; ts += c - b
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('%-34s %.2f ms/step wall, host %.2f (next %.2f + step %.2f)' % (label, 1e3 * (time.perf_counter() - t0) / K, 1e3 * th / K,
                                                                        1e3 * tn / K, 1e3 * ts / K), flush=True)
    if hasattr(it, 'close'):
        it.close()


src = lambda: [hb[i % 2] for i in range(K + 4)]
t0 = time.perf_counter()
for i in range(K):
    step(staged[i % 2])
torch.cuda.synchronize()
print('%-34s %.2f ms/step wall' % ('pre-staged', 1e3 * (time.perf_counter() - t0) / K), flush=True)
from scene_generation_amd.pipeline import DeviceBatch


class FreshClones(object):
    """the pre-staged device batches, but every step gets NEW device tensors (clones on the main stream): same data, new identities,
    no host copy, no second stream"""
    def __init__(self, n):
        self.i, self.n = 0, n
    def __iter__(self):
        return self
    def __next__(self):
        if self.i >= self.n:
            raise StopIteration
        db = staged[self.i % 2]
        self.i += 1
        return DeviceBatch(type(db.batch)(*[t.clone() for t in db.batch]), db.objs_host, db.obj_to_img_host, db.seg_offsets_host, db.num_images)


class SideStreamClones(FreshClones):
    """... cloned on a side stream one step ahead, handed over with an event + record_stream (the prefetcher's device side only)"""
    def __init__(self, n):
        FreshClones.__init__(self, n)
        self.s = torch.cuda.Stream()
        self.q = []
    def _stage(self):
        db = staged[self.i % 2]
        self.i += 1
        self.s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.s):
            b = type(db.batch)(*[t.clone() for t in db.batch])
        ev = torch.cuda.Event(); ev.record(self.s)
        self.q.append((DeviceBatch(b, db.objs_host, db.obj_to_img_host, db.seg_offsets_host, db.num_images), ev))
    def __next__(self):
        while len(self.q) < 2 and self.i < self.n:
            self._stage()
        if not self.q:
            raise StopIteration
        db, ev = self.q.pop(0)
        torch.cuda.current_stream().wait_event(ev)
        for t in db.batch:
            t.record_stream(torch.cuda.current_stream())
        return db


for rep in range(2):
    run('fresh clones, main stream', lambda: FreshClones(K + 4))
    run('fresh clones, side stream + event', lambda: SideStreamClones(K + 4))
    run('threaded (bounded put)', lambda: DeviceBatchPrefetcher(src(), dev))
    run('inline (threaded=False)', lambda: DeviceBatchPrefetcher(src(), dev, threaded=False))
    run('threaded, depth 4', lambda: DeviceBatchPrefetcher(src(), dev, depth=4))
    run('threaded, validate=False', lambda: DeviceBatchPrefetcher(src(), dev, validate=False))
    sw = sys.getswitchinterval()
    sys.setswitchinterval(0.0005)
    run('threaded, switchinterval 0.5 ms', lambda: DeviceBatchPrefetcher(src(), dev))
    sys.setswitchinterval(sw)
