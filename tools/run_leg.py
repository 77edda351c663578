"""Run K full G+D steps of one BASELINE config leg (c2 = configs[1], c4 = configs[3] per-GPU shape, c5 = configs[4]) the way
bench.py's legs do -- for profiling one leg alone:  rocprofv3 --kernel-trace -d out -- python tools/run_leg.py c5 8
Prints images/s of the timed steps (after 4 warm-up steps)."""
import os
import random
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scene_generation_amd.args import parser
from scene_generation_amd.pipeline import DeviceBatchPrefetcher
from scene_generation_amd.synthetic import make_config_batch, make_vocab, CONFIGS
from scene_generation_amd.trainer import Trainer

name = sys.argv[1] if len(sys.argv) > 1 else 'c5'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = CONFIGS[name]
dev = 'cuda:0'
torch.cuda.set_device(0)
args = parser.parse_args(['--image_size', '%d,%d' % (cfg['size'], cfg['size']), '--batch_size', str(cfg['N']),
                          '--vgg_features_weight', '0', '--output_dir', '/tmp/o'])
torch.manual_seed(1234)
tr = Trainer(args, make_vocab(), device=dev)
tr.model.layout_objects_hint = cfg['max_objs'] + 1
tr.dense_layout_outputs = False
st = list(DeviceBatchPrefetcher([make_config_batch(name, seed=2000 + i) for i in range(2)], dev))
random.seed(0)


def one(i):
    db = st[i % 2]
    tr.model.objs_host, tr.model.obj_to_img_host = db.objs_host, db.obj_to_img_host
    tr.step(db.batch, use_gt=tr.draw_use_gt())


for i in range(4):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    one(4 + i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('%s: %.1f images/s, %.2f ms/step (%d steps, O=%d T=%d)' % (name, cfg['N'] * steps / dt, 1e3 * dt / steps, steps,
                                                                  st[0].batch.objs.numel(), st[0].batch.triples.size(0)))
