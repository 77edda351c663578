"""Micro-benchmark of the implicit-GEMM conv kernels on the shapes of BASELINE config 2 (HIP events, GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scene_generation_amd import ops

DEV = 'cuda'
SHAPES = [
    # name, N, Cin, H, Cout, KS, stride, pad, reflect, ups
    ('res3x3_1024@8', 32, 1024, 8, 1024, 3, 1, 1, True, 1),
    ('first7x7_204@128', 32, 204, 128, 64, 7, 1, 3, True, 1),
    ('down3x3s2_64@128', 32, 64, 128, 128, 3, 2, 1, False, 1),
    ('down3x3s2_512@16', 32, 512, 16, 1024, 3, 2, 1, False, 1),
    ('last7x7_64@128', 32, 64, 128, 3, 7, 1, 3, True, 1),
    ('D0_4x4s2_207@128', 32, 207, 128, 64, 4, 2, 2, False, 1),
    ('D2_4x4s2_128@33', 32, 128, 33, 256, 4, 2, 2, False, 1),
    ('D3_4x4s1_256@17', 32, 256, 17, 512, 4, 1, 2, False, 1),
    ('D4_4x4s1_512@18', 32, 512, 18, 1, 4, 1, 2, False, 1),
    ('mask3x3_192@16up', 288, 192, 16, 192, 3, 1, 1, False, 2),
]


for spec in filter(None, os.environ.get('SG_BENCH_SHAPES', '').split(';')):
    f = spec.split(':')
    SHAPES.append((f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), f[8] == '1', int(f[9])))


def timeit(fn, n=int(os.environ.get('SG_BENCH_ITERS', '10')), reps=3):
    """best of ``reps`` runs of ``n`` back-to-back calls (HIP events on the launch stream)"""
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


for name, N, Cin, H, Cout, KS, st, pad, refl, ups in SHAPES:
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
        continue
    x = torch.randn(N, Cin, H, H, device=DEV, requires_grad=True)
    w = torch.randn(Cout, Cin, KS, KS, device=DEV, requires_grad=True) * 0.05
    b = torch.zeros(Cout, device=DEV)
    y = ops.conv2d(x, w, b, stride=st, pad=pad, reflect=refl, upsample=ups)
    gy = torch.randn_like(y)
    flops = 2.0 * y.numel() * Cin * KS * KS
    t_f = timeit(lambda: ops.conv2d(x, w, b, stride=st, pad=pad, reflect=refl, upsample=ups))
    xw = x.detach().requires_grad_(False)

    def bwd_data():
        xx = x.detach().requires_grad_(True)
        yy = ops.conv2d(xx, w.detach(), b, stride=st, pad=pad, reflect=refl, upsample=ups)
        return xx, yy
    xx, yy = bwd_data()
    t_d = timeit(lambda: torch.autograd.grad(yy, xx, gy, retain_graph=True))
    ww = w.detach().requires_grad_(True)
    y2 = ops.conv2d(x.detach(), ww, b, stride=st, pad=pad, reflect=refl, upsample=ups)
    t_w = timeit(lambda: torch.autograd.grad(y2, ww, gy, retain_graph=True))
    print('%-20s out %-18s GF %7.1f | fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF' % (
        name, tuple(y.shape), flops / 1e9, t_f, flops / t_f / 1e9, t_d, flops / t_d / 1e9, t_w, flops / t_w / 1e9), flush=True)
