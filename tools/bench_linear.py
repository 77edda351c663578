"""Micro-benchmark of the two dense-layer kernels (register-streaming skinny.hip vs the LDS-tiled igemm) on the nn.Linear
shapes of the step: graph-convolution MLPs at BASELINE configs[1] (O ~ 208 objects, T ~ 224 triples), configs[4] (O = 1056,
T = 3072) and the per-GPU shape of configs[3].  HIP events on the launch stream, GPU box.  Usage: python tools/bench_linear.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scene_generation_amd import ops, _hip

DEV = 'cuda'
SHAPES = [('c2 net1.0', 224, 454, 512), ('c2 net1.1', 224, 512, 1152), ('c2 net2.0', 208, 512, 512), ('c2 net2.1', 208, 512, 128),
          ('c2 box_net', 208, 128, 512), ('c5 net1.0', 3072, 454, 512), ('c5 net1.1', 3072, 512, 1152), ('c5 net2.0', 1056, 512, 512),
          ('c5 net2.1', 1056, 512, 128), ('c4 net1.1', 140, 512, 1152), ('objD fc', 208, 1024, 172)]


def timeit(fn, n=20, reps=3):
    fn(); fn(); torch.cuda.synchronize()
    best = float('inf')
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def run(name, rows, inf, outf):
    x = torch.randn(rows, inf, device=DEV)
    w = torch.randn(outf, inf, device=DEV) * 0.05
    b = torch.zeros(outf, device=DEV)
    y = torch.empty(rows, outf, device=DEV)
    gy = torch.randn(rows, outf, device=DEV)
    gx = torch.empty_like(x)
    gw = torch.empty_like(w)
    s = ops._stream()
    fl = 2.0 * rows * inf * outf
    res = []
    for kernel, thr in (('tiled', 0), ('skinny', 1 << 30)):
        _hip.set_option('linear_skinny', thr)
        tf = timeit(lambda: ops._call('sg_linear_fwd', x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, inf, outf, 1, 0.0, s))
        td = timeit(lambda: ops._call('sg_linear_bwd_data', gy.data_ptr(), w.data_ptr(), gx.data_ptr(), rows, inf, outf, s))
        tw = timeit(lambda: ops._call('sg_linear_bwd_weight', gy.data_ptr(), x.data_ptr(), gw.data_ptr(), None, rows, inf, outf, s))
        res.append((kernel, tf, td, tw))
    _hip.set_option('linear_skinny', 2048)
    for kernel, tf, td, tw in res:
        print('%-11s %5d x %4d -> %4d  %-6s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF' % (
            name, rows, inf, outf, kernel, 1e3 * tf, fl / tf / 1e9, 1e3 * td, fl / td / 1e9, 1e3 * tw, fl / tw / 1e9), flush=True)


for sh in SHAPES:
    run(*sh)
