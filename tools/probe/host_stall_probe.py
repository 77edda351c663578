"""Where does the launching thread WAIT once the step is GPU-bound?  (round 6, host path)

tools/probe/queue_depth_probe.py shows that the runtime lets a thread run >= 6000 launches ahead of the GPU without blocking, yet
the step's host time is 28.7 ms under back-pressure against 11-13 ms into an idle GPU.  This probe times every C-ABI call, every
graph replay, every autograd backward and every host->device staging call of K back-to-back steps (no synchronisation in
between) and lists the calls that blocked, for a resident batch and for batches fed through DeviceBatchPrefetcher.
"""
import collections
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scene_generation_amd import ops, graphs, utils
from scene_generation_amd.ops import _core
from scene_generation_amd.args import parser
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

REC = collections.defaultdict(lambda: [0, 0.0, 0.0])       # name -> [calls, total s, max s]
SLOW = []


def note(name, dt):
    r = REC[name]
    r[0] += 1
    r[1] += dt
    r[2] = max(r[2], dt)
    if dt > 0.3e-3:
        SLOW.append((name, dt))


def timed_wrapper(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            note(name, time.perf_counter() - t0)
    return w


_orig_call = _core._call


def _call(name, *a):
    t0 = time.perf_counter()
    try:
        return _orig_call(name, *a)
    finally:
        note('C:' + name, time.perf_counter() - t0)


for mod in list(sys.modules.values()):             # every module that imported _call by name
    if mod is not None and getattr(mod, '_call', None) is _orig_call:
        mod._call = _call
torch.cuda.CUDAGraph.replay = timed_wrapper('graph.replay', torch.cuda.CUDAGraph.replay)
torch.autograd.backward = timed_wrapper('autograd.backward', torch.autograd.backward)
torch.Tensor.backward = timed_wrapper('Tensor.backward', torch.Tensor.backward)
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, 'to_device_async', None) is utils.to_device_async:
        mod.to_device_async = timed_wrapper('to_device_async', utils.to_device_async)
torch.empty = timed_wrapper('torch.empty', torch.empty)
torch.empty_like = timed_wrapper('torch.empty_like', torch.empty_like)
torch.cat = timed_wrapper('torch.cat', torch.cat)
torch.zeros = timed_wrapper('torch.zeros', torch.zeros)


def step(db):
    tr.model.objs_host, tr.model.obj_to_img_host = db.objs_host, db.obj_to_img_host
    tr.step(db.batch, use_gt=tr.draw_use_gt())


for i in range(6):
    step(staged[i % 2])
torch.cuda.synchronize()
K = 20


def run(label, it):
    for _ in range(4):
        step(next(it))
    torch.cuda.synchronize()
    REC.clear()
    del SLOW[:]
    del GC_LOG[:]
    per = []
    t0 = time.perf_counter()
    for i in range(K):
        a = time.perf_counter()
        db = next(it)
        b = time.perf_counter()
        step(db)
        per.append((b - a, time.perf_counter() - b))
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tw = time.perf_counter() - t0
    print('==== %s: %.2f ms/step wall, %.2f ms/step host' % (label, 1e3 * tw / K, 1e3 * th / K))
    print('per step host ms (next + step): ' + ' '.join('%.1f+%.1f' % (1e3 * x, 1e3 * y) for x, y in per))
    print('garbage collections in the timed region: %s; tracked objects %d' % (
        ', '.join('gen%d x%d %.1f ms (max %.1f)' % (g, sum(1 for x in GC_LOG if x[0] == g), 1e3 * sum(x[1] for x in GC_LOG if x[0] == g),
                                                   1e3 * max([x[1] for x in GC_LOG if x[0] == g] or [0])) for g in (0, 1, 2)),
        len(gc.get_objects())))
    rows = sorted(REC.items(), key=lambda kv: -kv[1][1])[:18]
    for name, (n, tot, mx) in rows:
        print('  %-44s %6d calls  %8.2f ms/step  max %7.2f ms' % (name, n, 1e3 * tot / K, 1e3 * mx))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, dt in SLOW:
        agg[name][0] += 1
        agg[name][1] += dt
    print('  calls > 0.3 ms: ' + ', '.join('%s x%d %.1f ms' % (n, c, 1e3 * t) for n, (c, t) in
                                            sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]))
    if hasattr(it, 'close'):
        it.close()


class Resident(object):
    def __init__(self):
        self.i = 0

    def __next__(self):
        self.i += 1
        return staged[self.i % 2]


src = lambda: [hb[i % 2] for i in range(K + 4)]
VARIANT = os.environ.get('PROBE_VARIANT', '')


def make_pf(**kw):
    pf = DeviceBatchPrefetcher(src(), dev, **kw)
    if VARIANT == 'mainstream':                    # H2D copies on the launch stream itself (no second queue, no event wait)
        pf.stream = torch.cuda.current_stream()
    return iter(pf)


if VARIANT == 'norecord':                          # (timing probe only: the allocator may hand the blocks back too early)
    torch.Tensor.record_stream = lambda self, s: None
import gc
GC_LOG = []
_gc_t = [0.0]


def _gc_cb(phase, info):
    if phase == 'start':
        _gc_t[0] = time.perf_counter()
    else:
        GC_LOG.append((info['generation'], time.perf_counter() - _gc_t[0], info['collected']))


gc.callbacks.append(_gc_cb)
if os.environ.get('PROBE_GC') == 'off':
    gc.disable()
elif os.environ.get('PROBE_GC') == 'freeze':
    gc.collect()
    gc.freeze()
print('cpus: affinity %d, os.cpu_count %d, torch threads %d; cgroup cpu.max: %s' % (
    len(os.sched_getaffinity(0)), os.cpu_count(), torch.get_num_threads(),
    (open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a')))
if os.path.exists('/sys/fs/cgroup/cpu.stat'):
    print('cpu.stat before: ' + ' '.join(open('/sys/fs/cgroup/cpu.stat').read().split()))
if os.environ.get('PROBE_TORCH_THREADS'):
    torch.set_num_threads(int(os.environ['PROBE_TORCH_THREADS']))
MODES = os.environ.get('PROBE_MODES', 'resident,threaded,inline').split(',')
for rep in range(int(os.environ.get('PROBE_REPS', '2'))):
    if not VARIANT and 'resident' in MODES:
        run('resident', Resident())
    if 'threaded' in MODES:
        run('prefetcher (threaded) %s' % VARIANT, make_pf())
    if 'inline' in MODES:
        run('prefetcher (inline) %s' % VARIANT, make_pf(threaded=False))
if os.path.exists('/sys/fs/cgroup/cpu.stat'):
    print('cpu.stat after: ' + ' '.join(open('/sys/fs/cgroup/cpu.stat').read().split()))
