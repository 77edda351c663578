"""Per-call breakdown of one training step on the GPU box: every C-ABI call of the step (graphs off) is bracketed with a HIP
event pair on the launch stream and aggregated by (entry point, shape).  Conv calls print their sgConvDesc and the TFLOP/s of
the direct-form MACs, so the table says WHICH layer runs on which kernel at what rate.

  python tools/step_shapes.py [--steps 3] [--config c2|c4|c5] [--out gpurun_out/shapes.md]
"""
import argparse
import os
import random
import sys
from collections import OrderedDict

os.environ.setdefault('SG_GRAPHS', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from scene_generation_amd import ops
from scene_generation_amd._hip import sgConvDesc
from scene_generation_amd.args import parser
from scene_generation_amd.synthetic import make_config_batch, make_vocab, CONFIGS
from scene_generation_amd.pipeline import DeviceBatchPrefetcher
from scene_generation_amd.trainer import Trainer



def desc_of(arg):
    obj = getattr(arg, '_obj', None)
    return obj if isinstance(obj, sgConvDesc) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--config', default='c2')
    ap.add_argument('--out', default='gpurun_out/shapes.md')
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    S = cfg['size']
    dev = 'cuda:0'
    args = parser.parse_args(['--image_size', '%d,%d' % (S, S), '--batch_size', str(cfg['N']), '--vgg_features_weight', '0',
                              '--output_dir', '/tmp/o'])
    torch.manual_seed(1234)
    tr = Trainer(args, make_vocab(), device=dev)
    tr.model.layout_objects_hint = cfg['max_objs'] + 1
    tr.dense_layout_outputs = False
    host = [make_config_batch(a.config, seed=i) for i in range(2)]
    staged = list(DeviceBatchPrefetcher(host, dev))
    random.seed(0)

    def one_step(i):
        db = staged[i % 2]
        tr.model.objs_host, tr.model.obj_to_img_host = db.objs_host, db.obj_to_img_host
        tr.step(db.batch, use_gt=random.randint(0, 1) != 0)

    for i in range(3):
        one_step(i)
    torch.cuda.synchronize()

    recs = []
    orig = ops._call

    def timed_call(name, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *args)
        e1.record()
        d = desc_of(args[0]) if args else None
        if d is not None:
            key = (name, tuple(getattr(d, f) for f, _ in sgConvDesc._fields_))
        else:
            key = (name, tuple(x for x in args if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 24))[:6])
        recs.append((key, e0, e1))

    ops._call = timed_call
    for i in range(a.steps):
        one_step(3 + i)
    torch.cuda.synchronize()
    ops._call = orig

    agg = OrderedDict()
    for key, e0, e1 in recs:
        v = agg.setdefault(key, [0, 0.0])
        v[0] += 1
        v[1] += e0.elapsed_time(e1)
    rows = []
    for (name, shp), (n, ms) in agg.items():
        tf = ''
        if len(shp) == 15:
            N, C1, C2, H, W, Cout, KS, st, pad, refl, ups, OH, OW, opad, bc = shp
            if 'convT' in name:
                macs = float(N) * H * W * Cout * C1 * KS * KS
            else:
                macs = float(N) * OH * OW * Cout * (C1 + C2) * KS * KS
            tf = '%.1f' % (2 * macs * n / (ms * 1e-3) / 1e12) if ms > 0 else ''
            shp = 'N%d C%d+%d %dx%d -> %d k%d s%d p%d%s u%d out %dx%d' % (N, C1, C2, H, W, Cout, KS, st, pad, 'r' if refl else '',
                                                                    ups, OH, OW)
        rows.append((ms / a.steps, n / a.steps, name, str(shp), tf))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    with open(a.out, 'w') as f:
        f.write('# per-call breakdown, config %s, %d steps, graphs off; total %.2f ms/step in %d distinct (call, shape)\n\n' % (
            a.config, a.steps, total, len(rows)))
        f.write('| ms/step | calls/step | entry point | shape | TFLOP/s (direct-form MACs) |\n|---|---|---|---|---|\n')
        for ms, n, name, shp, tf in rows:
            if ms < 0.02:
                continue
            f.write('| %.3f | %.1f | %s | %s | %s |\n' % (ms, n, name, shp, tf))
    print('wrote', a.out, 'total %.2f ms/step' % total)


if __name__ == '__main__':
    main()
