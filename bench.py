#!/usr/bin/env python
"""bench.py -- images/sec of one full G+D training step (train.py:190-215 semantics) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): COCO-Stuff-shaped synthetic scene graphs, 128x128, <=8 objects/image (+ the
__image__ node), batch 32 PER GPU (weak scaling; configs[2] = 256 over 8 GPUs), reference default widths (183 M-param
generator, 2-scale PatchGAN, object / mask discriminators), fp32, VGG loss off (needs pretrained weights).
A step = Model.forward + train_generator + the three discriminator steps incl. four Adam updates and, for N>1, the
RCCL gradient all-reduces.  Inputs are resident in HBM before the timed region.  One JSON line on rank 0.
"""
import argparse
import json
import os
import random
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
HBM_PEAK_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=6)
    p.add_argument('--warmup', type=int, default=2)
    p.add_argument('--batch_per_gpu', type=int, default=32)
    p.add_argument('--image_size', type=int, default=128)
    p.add_argument('--cpu_baseline', default='auto', choices=['auto', 'off'])
    p.add_argument('--cpu_images', type=int, default=4)
    p.add_argument('--no_prof', action='store_true')
    p.add_argument('--no_share_d_forward', action='store_true',
                   help='re-run the mask/image discriminator forwards in the D steps like the reference does')
    return p.parse_args()


def cpu_baseline(image_size, n_images):
    """The oracle (oracle/sg_oracle.py, kind "port") timed on the host cores on a bounded sample of the same
    workload: the full G+D step at the same widths on ``n_images`` images (1 warm-up + 1 timed step)."""
    from oracle import sg_oracle as O
    from scene_generation_amd.args import parser
    from scene_generation_amd.synthetic import make_batch, make_vocab
    args = parser.parse_args(['--image_size', '%d,%d' % (image_size, image_size), '--batch_size', str(n_images),
                              '--vgg_features_weight', '0', '--output_dir', '/tmp/o'])
    torch.manual_seed(0)
    tr = O.Trainer(args, make_vocab())
    batch = make_batch(N=n_images, min_objs=3, max_objs=8, size=image_size, seed=0)
    random.seed(0)
    tr.step(batch, use_gt=True)
    t0 = time.perf_counter()
    tr.step(batch, use_gt=False)
    dt = time.perf_counter() - t0
    return {'value': n_images / dt, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'full G+D step (same widths, fp32, torch-CPU oracle), %d images of the %dx%d workload, '
                      '1 warm-up + 1 timed step, %.1f s' % (n_images, image_size, image_size, dt)}


def main():
    a = parse()
    # the host driver only supports dmabuf IPC: RCCL / cross-process tensor sharing fails without it (set before HIP starts)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit('launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node %d bench.py '
                             '--gpus %d ...' % (a.gpus, a.gpus))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback in the product path)'
    if os.environ.get('SG_SHARE_GPU') == '1':        # debugging aid: every rank on GPU 0 (needs SG_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = 'cuda:%d' % local
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('SG_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)

    from scene_generation_amd import ops
    from scene_generation_amd.args import parser
    from scene_generation_amd.synthetic import make_batch, make_vocab, batch_to
    from scene_generation_amd.trainer import Trainer

    S, B = a.image_size, a.batch_per_gpu
    args = parser.parse_args(['--image_size', '%d,%d' % (S, S), '--batch_size', str(B * world),
                              '--vgg_features_weight', '0', '--output_dir', '/tmp/o'])
    torch.manual_seed(1234)                       # same initial weights on every rank (also broadcast in Trainer)
    tr = Trainer(args, make_vocab(), device=dev, distributed=world > 1)
    tr.model.layout_objects_hint = 9
    tr.share_d_forward = not a.no_share_d_forward
    # two pre-staged batches per rank (different data per rank: weak scaling), resident in HBM
    batches = [batch_to(make_batch(N=B, min_objs=3, max_objs=8, size=S, seed=1000 * rank + i), dev) for i in range(2)]
    hosts = [(b.objs.tolist(), b.obj_to_img.tolist()) for b in batches]
    random.seed(0)                                # the use_gt coin (train.py:195) must agree on all ranks: it decides
    torch.manual_seed(100 + rank)                 # which parameters receive gradients (and hence Adam updates)

    def one_step(i):
        tr.model.objs_host, tr.model.obj_to_img_host = hosts[i % 2]
        tr.step(batches[i % 2], use_gt=random.randint(0, 1) != 0)      # train.py:195

    for i in range(a.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    issue = [0.0]

    def timed(n_steps, first):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            one_step(first + i)
        issue[0] = time.perf_counter() - t0          # host done issuing; the GPU may still be working
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        return d

    # headline pass: exactly K steps, no per-launch instrumentation (the step issues ~2300 kernels and is within a few
    # percent of being launch-bound: two hipEventRecord per kernel cost ~6 % of the step)
    dt = timed(a.steps, a.warmup)
    host_issue = issue[0]
    # roofline pass: the SAME K steps again with a HIP event pair around every kernel launch on the launch stream
    dt_prof = None
    if not a.no_prof:
        ops.prof_reset()
        ops.prof_enable(True)
        dt_prof = timed(a.steps, a.warmup + a.steps)
        ops.prof_enable(False)
    # sanity: the step really trained (finite losses)
    total = dict(tr.generator_losses.items())['total_loss']
    assert total == total and abs(total) < 1e6, 'non-finite generator loss %r' % total

    out = {
        'metric': 'images/sec G+D step, 128x128 <=8-obj scene graphs', 'value': B * world * a.steps / dt,
        'unit': 'images/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: COCO-Stuff-shaped %dx%d, <=8 objects/img (+__image__), batch %d '
                               'per GPU, full G+D train step (G fwd/bwd + 3 D steps + 4 Adam%s), reference default '
                               'widths, VGG loss off' % (S, S, B, ' + RCCL grad all-reduce' if world > 1 else ''),
                   'global_batch': B * world, 'image_size': S, 'parallelism': 'dp%d' % world,
                   'share_d_forward': not a.no_share_d_forward},
        # wall time the host needed to ISSUE the K steps (no sync): close to ms_per_step => launch-bound
        'host_issue_ms_per_step': 1e3 * host_issue / a.steps,
    }
    if rank == 0:
        if not a.no_prof:
            prof = ops.prof_read()
            ig = {k: v for k, v in prof.items() if k.startswith('igemm') and v['launches'] > 0}
            all_ms = sum(v['ms'] for v in prof.values())
            if ig:
                name, v = max(ig.items(), key=lambda kv: kv[1]['ms'])
                ach = v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] > 0 else 0.0
                out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': ach / F32_MFMA_PEAK_TFLOPS, 'traffic': None, 'kernel': name,
                                   'launches': v['launches'], 'avg_us': 1e3 * v['ms'] / v['launches'],
                                   'share_of_step': v['ms'] / (1e3 * dt_prof),
                                   'measured_in': 'second pass of the same %d steps with a HIP event pair around every launch '
                                                  '(%.1f ms/step; the headline pass runs without the events)'
                                                  % (a.steps, 1e3 * dt_prof / a.steps)}
                igms = sum(x['ms'] for x in ig.values())
                igfl = sum(x['flops'] for x in ig.values())
                out['kernels'] = {
                    'all_igemm': {'ms_per_step': igms / a.steps, 'tflops': igfl / (igms * 1e-3) / 1e12 if igms else 0.0},
                    'timed_kernels_ms_per_step': all_ms / a.steps,
                    'top': {k: {'ms_per_step': round(x['ms'] / a.steps, 3), 'launches_per_step': x['launches'] / a.steps,
                                'tflops': round(x['flops'] / (x['ms'] * 1e-3) / 1e12, 2) if x['ms'] and x['flops'] else None,
                                'gbs': round(x['bytes'] / (x['ms'] * 1e-3) / 1e9, 1) if x['ms'] and x['bytes'] else None}
                            for k, x in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:12] if x['launches']}}
        if world == 1 and a.cpu_baseline == 'auto':
            try:
                out['cpu_baseline'] = cpu_baseline(S, a.cpu_images)
            except Exception as e:           # the baseline is a report, never a reason to lose the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'images/s', 'cores': torch.get_num_threads(),
                                       'kind': 'port', 'sample': 'failed: %r' % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
