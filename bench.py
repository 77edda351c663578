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
    p.add_argument('--warmup', type=int, default=4)
    p.add_argument('--batch_per_gpu', type=int, default=32)
    p.add_argument('--image_size', type=int, default=128)
    p.add_argument('--cpu_baseline', default='auto', choices=['auto', 'off'])
    p.add_argument('--cpu_images', type=int, default=8)
    p.add_argument('--cpu_steps', type=int, default=3)
    p.add_argument('--no_prof', action='store_true')
    p.add_argument('--no_secondary', action='store_true', help='skip the secondary passes (default-flags step with the VGG '
                   'loss on; the step with every fast path off)')
    p.add_argument('--vgg', type=float, default=0.0, help='--vgg_features_weight of the HEADLINE pass (SURVEY 8d: 0; the '
                   'reference default 10 is reported as secondary.default_flags_vgg_on)')
    p.add_argument('--no_share_d_forward', action='store_true',
                   help='re-run the mask/image discriminator forwards in the D steps like the reference does')
    return p.parse_args()


def physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or os.cpu_count()
    except OSError:
        return os.cpu_count()


def cpu_baseline(image_size, n_images, n_steps):
    """The oracle (oracle/sg_oracle.py, kind "port") timed on the host cores on a bounded sample of the same workload
    (SURVEY 8d): the full G+D step at the same widths and flags as the headline pass on ``n_images`` images, 1 warm-up +
    ``n_steps`` timed steps."""
    from oracle import sg_oracle as O
    from scene_generation_amd.args import parser
    from scene_generation_amd.synthetic import make_batch, make_vocab
    args = parser.parse_args(['--image_size', '%d,%d' % (image_size, image_size), '--batch_size', str(n_images),
                              '--vgg_features_weight', '0', '--output_dir', '/tmp/o'])
    torch.manual_seed(0)
    tr = O.Trainer(args, make_vocab())
    batches = [make_batch(N=n_images, min_objs=3, max_objs=8, size=image_size, seed=i) for i in range(2)]
    random.seed(0)
    tr.step(batches[0], use_gt=True)
    t0 = time.perf_counter()
    for i in range(n_steps):
        tr.step(batches[(i + 1) % 2], use_gt=random.randint(0, 1) != 0)
    dt = time.perf_counter() - t0
    return {'value': n_images * n_steps / dt, 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'physical_cores': physical_cores(), 'logical_cpus': os.cpu_count(),
            'sample': 'full G+D step (same widths and flags as the headline pass, fp32, torch-CPU oracle), %d images of the '
                      '%dx%d workload per step, 1 warm-up + %d timed steps, %.1f s' % (n_images, image_size, image_size,
                                                                                        n_steps, dt)}


# what the profiler kinds are, as template instantiations (include/sg2im_hip.h, csrc/igemm.hip)
KERNEL_INSTANTIATIONS = {
    'wino_bgemm_t128': 'igemm_kernel<TileCfg<128,128,2,2,1>, LoadKContig<128,true,false>, LoadKContig<128,true,false>, EpRowMajorPlain>: '
                       'the 16 batched dense GEMMs of a Winograd F(2x2,3x3) conv (ResnetBlock / VGG19 convs), batch-major '
                       'tile order, 32-deep k-tiles, software-pipelined fragment reads',
    'wino_bgemm_t64': 'igemm_kernel<TileCfg<64,64,2,1>, LoadKContig<64,true,false>, LoadKContig<64,true,false>, EpRowMajor>: '
                      'Winograd GEMMs of the 192-channel mask_net convs',
}


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
    backend = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('SG_DIST_BACKEND', 'nccl')
        dist.init_process_group(backend, rank=rank, world_size=world)

    from scene_generation_amd import ops, graphs
    from scene_generation_amd.args import parser
    from scene_generation_amd.synthetic import make_batch, make_vocab
    from scene_generation_amd.pipeline import DeviceBatchPrefetcher
    from scene_generation_amd.trainer import Trainer

    S, B = a.image_size, a.batch_per_gpu
    vocab = make_vocab()

    def make_trainer(vgg_weight):
        args = parser.parse_args(['--image_size', '%d,%d' % (S, S), '--batch_size', str(B * world),
                                  '--vgg_features_weight', str(vgg_weight), '--output_dir', '/tmp/o'])
        torch.manual_seed(1234)                   # same initial weights on every rank (also broadcast in Trainer)
        tr = Trainer(args, vocab, device=dev, distributed=world > 1)
        tr.model.layout_objects_hint = 9
        tr.share_d_forward = not a.no_share_d_forward
        tr.dense_layout_outputs = False           # nobody reads the three dense layouts here (TensorBoard-only outputs)
        return tr

    tr = make_trainer(a.vgg)
    # two collated host batches per rank (different data per rank: weak scaling), staged to HBM through the collate ->
    # device adapter (pinned double-buffered H2D, host lists for the VectorPool / factored layout planning)
    host_batches = [make_batch(N=B, min_objs=3, max_objs=8, size=S, seed=1000 * rank + i) for i in range(2)]
    staged = list(DeviceBatchPrefetcher(host_batches, dev))
    random.seed(0)                                # the use_gt coin (train.py:195): drawn on rank 0, broadcast (Trainer)
    torch.manual_seed(100 + rank)

    def one_step(trainer, i):
        db = staged[i % 2]
        trainer.model.objs_host, trainer.model.obj_to_img_host = db.objs_host, db.obj_to_img_host
        trainer.step(db.batch, use_gt=trainer.draw_use_gt())      # train.py:195

    issue = [0.0]

    def timed(trainer, n_steps, first):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            one_step(trainer, first + i)
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

    # untimed warm-up; with fewer than 3 steps the hipGraph capture of the static sub-networks (graphs.py: two eager steps,
    # capture on the third) would land inside the timed region
    for i in range(a.warmup):
        one_step(tr, i)
    # headline pass: exactly K steps, no per-launch instrumentation
    calls0, replays0 = ops.CALLS[0], graphs.REPLAYS[0]
    dt = timed(tr, a.steps, a.warmup)
    host_issue = issue[0]
    calls, replays = (ops.CALLS[0] - calls0) / a.steps, (graphs.REPLAYS[0] - replays0) / a.steps
    # host cost of ISSUING one step, measured with an idle GPU in front of it: once the step is GPU-bound the number above
    # mostly measures back-pressure of the full launch queue, not host work
    iso = []
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_step(tr, a.warmup + a.steps + i)
        iso.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    # roofline pass: the SAME K steps again with a HIP event pair around every kernel launch on the launch stream
    dt_prof = None
    if not a.no_prof:
        ops.prof_reset()
        ops.prof_enable(True)
        dt_prof = timed(tr, a.steps, a.warmup + a.steps + 3)
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
                               'widths, VGG loss %s' % (S, S, B, ' + RCCL grad all-reduce' if world > 1 else '',
                                                        'on (weight %g)' % a.vgg if a.vgg > 0 else 'off (SURVEY 8d)'),
                   'global_batch': B * world, 'image_size': S, 'parallelism': 'dp%d' % world,
                   'share_d_forward': not a.no_share_d_forward, 'vgg_features_weight': a.vgg,
                   'dense_layout_outputs': False, 'hip_graphs': graphs.ENABLED},
        # wall time the host needed to ISSUE the K steps (no sync): close to ms_per_step => launch-bound
        'host_issue_ms_per_step': 1e3 * host_issue / a.steps,
        # the same for a single step issued into an idle GPU (no queue back-pressure): the host-side cost of a step
        'host_issue_isolated_ms_per_step': 1e3 * sum(iso) / len(iso),
        # C-ABI calls (each launches 1..4 kernels) and hipGraph launches the host issues per step; the kernels inside a
        # replayed graph are dispatched by the GPU front-end without host involvement
        'host_calls_per_step': calls, 'graph_replays_per_step': replays,
    }
    if world > 1:
        out['rccl_ranks'] = dist.get_world_size()
        out['dist_backend'] = backend
    if rank == 0 and not a.no_prof:
        prof = ops.prof_read()
        mm = {k: v for k, v in prof.items() if v['launches'] > 0 and v['flops'] > 0 and k != 'linear'}
        all_ms = sum(v['ms'] for v in prof.values())
        out['launches_per_step'] = sum(v['launches'] for v in prof.values()) / a.steps
        if mm:
            name, v = max(mm.items(), key=lambda kv: kv[1]['ms'])
            ach = v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] > 0 else 0.0
            traffic, tsrc = None, None
            tpath = os.path.join(ROOT, 'profiles', 'r02_pmc_traffic.json')
            if os.path.isfile(tpath):
                tab = json.load(open(tpath)).get(name)
                if tab:
                    traffic, tsrc = tab.get('bytes_per_launch'), tab.get('source')
            out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / F32_MFMA_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': tsrc,
                               'kernel': name, 'instantiation': KERNEL_INSTANTIATIONS.get(name, name),
                               'flops_counted': 'MACs of the non-padding tiles the launch computes (useful work)',
                               'launches': v['launches'], 'avg_us': 1e3 * v['ms'] / v['launches'],
                               'flops_per_launch': v['flops'] / v['launches'],
                               'share_of_step': v['ms'] / (1e3 * dt_prof),
                               'measured_in': 'second pass of the same %d steps with a HIP event pair around every launch '
                                              '(%.1f ms/step; the headline pass runs without the events)'
                                              % (a.steps, 1e3 * dt_prof / a.steps)}
            igms = sum(x['ms'] for x in mm.values())
            igfl = sum(x['flops'] for x in mm.values())
            out['kernels'] = {
                'all_mfma_gemms': {'ms_per_step': igms / a.steps, 'tflops': igfl / (igms * 1e-3) / 1e12 if igms else 0.0},
                'timed_kernels_ms_per_step': all_ms / a.steps,
                'top': {k: {'ms_per_step': round(x['ms'] / a.steps, 3), 'launches_per_step': x['launches'] / a.steps,
                            'tflops': round(x['flops'] / (x['ms'] * 1e-3) / 1e12, 2) if x['ms'] and x['flops'] else None,
                            'gbs': round(x['bytes'] / (x['ms'] * 1e-3) / 1e9, 1) if x['ms'] and x['bytes'] else None}
                        for k, x in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:14] if x['launches']}}
    if world == 1 and not a.no_secondary:
        sec = {}
        n2 = max(2, min(a.steps, 6))
        # (1) the reference's DEFAULT flags: VGG19 perceptual loss on (args.py:73), random-init VGG weights
        if a.vgg == 0:
            tr2 = make_trainer(10.0)
            for i in range(4):                    # 2 eager steps + the hipGraph capture + 1 replay before timing
                one_step(tr2, i)
            d2 = timed(tr2, n2, 4)
            sec['default_flags_vgg_on'] = {'images_per_s': B * n2 / d2, 'ms_per_step': 1e3 * d2 / n2, 'steps': n2,
                                           'note': '--vgg_features_weight 10 (args.py:73), He-normal VGG19 weights'}
            del tr2
        # (2) every fast path off: direct convs instead of Winograd, dense 204-channel layout convs (channel-sparse),
        #     discriminator forwards re-run in the D steps like the reference, dense layouts materialised
        saved = (ops.WINOGRAD, ops.FACTORED_LAYOUT)
        try:
            ops.WINOGRAD = ops.FACTORED_LAYOUT = False
            tr3 = make_trainer(a.vgg)
            tr3.share_d_forward = False
            tr3.dense_layout_outputs = True
            for i in range(4):
                one_step(tr3, i)
            d3 = timed(tr3, n2, 4)
            sec['fast_paths_off'] = {'images_per_s': B * n2 / d3, 'ms_per_step': 1e3 * d3 / n2, 'steps': n2,
                                     'note': 'SG_WINOGRAD=0 SG_FACTORED_LAYOUT=0 --no_share_d_forward, dense layouts written'}
            del tr3
        finally:
            ops.WINOGRAD, ops.FACTORED_LAYOUT = saved
        out['secondary'] = sec
    if rank == 0:
        if world == 1 and a.cpu_baseline == 'auto':
            try:
                out['cpu_baseline'] = cpu_baseline(S, a.cpu_images, a.cpu_steps)
            except Exception as e:           # the baseline is a report, never a reason to lose the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'images/s', 'cores': torch.get_num_threads(),
                                       'kind': 'port', 'sample': 'failed: %r' % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
