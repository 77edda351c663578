"""Micro-benchmark of the InstanceNorm kernels at the shapes of the configs[1] step: scalar register form (instnorm_reg=1) against the
16-byte form (instnorm_reg=2).  GB/s on the algorithmic bytes (x in, y out forward; x, gy in, gx out backward); buffers rotate
through a set larger than the Infinity Cache so the numbers are HBM-side."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scene_generation_amd import ops, _hip
from scene_generation_amd.ops import _core

torch.cuda.set_device(0)
SHAPES = [(32, 64, 128, 128), (32, 128, 64, 64), (32, 256, 32, 32), (32, 512, 16, 16), (64, 128, 32, 32), (64, 256, 16, 16),
          (64, 512, 17, 17), (32, 128, 32, 32), (64, 128, 16, 16), (64, 256, 8, 8), (64, 512, 9, 9), (256, 64, 8, 8), (256, 128, 4, 4)]
R = 30


def bench(fn, nbytes):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(R):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / R
    return us, nbytes / us * 1e-3


print('| shape | pass | scalar us | GB/s | vec us | GB/s | speed-up |\n|---|---|---|---|---|---|---|')
for N, C, H, W in SHAPES:
    n = N * C * H * W
    nbuf = max(2, min(12, int(600e6 / (n * 4)) + 1))
    xs = [torch.randn(N, C, H, W, device='cuda') for _ in range(nbuf)]
    gs = [torch.randn(N, C, H, W, device='cuda') for _ in range(nbuf)]
    ys = [torch.empty(N, C, H, W, device='cuda') for _ in range(nbuf)]
    mean, rstd = torch.empty(N * C, device='cuda'), torch.empty(N * C, device='cuda')
    st = _core._stream()

    def fwd(i):
        k = i % nbuf
        _core._call('sg_instnorm_fwd', _core._p(xs[k]), None, _core._p(ys[k]), _core._p(mean), _core._p(rstd), N * C, H * W, 1e-5, 2, 0.2, st)

    def bwd(i):
        k = i % nbuf
        _core._call('sg_instnorm_bwd', _core._p(xs[k]), _core._p(gs[k]), _core._p(mean), _core._p(rstd), _core._p(ys[k]), N * C, H * W, 2, 0.2, st)
    for name, fn, nb in (('fwd', fwd, 8.0 * n), ('bwd', bwd, 12.0 * n)):
        res = []
        for opt in (1, 2):
            _hip.set_option('instnorm_reg', opt)
            res.append(bench(fn, nb))
        print('| %dx%dx%dx%d | %s | %.1f | %.0f | %.1f | %.0f | %.2f |' % (N, C, H, W, name, res[0][0], res[0][1], res[1][0], res[1][1],
                                                                          res[0][0] / res[1][0]))
    del xs, gs, ys
